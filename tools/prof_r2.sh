# Round-2 profiles (run through gpurun from the repo root): kernel-trace stats of the bench step, of the cfg5
# fp16 trunk and of the training step.  Summaries are copied into profiles/r2/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r2; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f16 -o f16 -- python3 tools/bench_local.py cfg5 --crops 1 --reps 3 --precision f16 > $O/f16.log 2>&1
ls $O/*/*
