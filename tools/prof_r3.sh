# Round-3 profiles (run through gpurun from the repo root): kernel-trace stats of the bench step.
# Summaries are copied into profiles/r3/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r3; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench.log 2>&1
ls $O/*/*
