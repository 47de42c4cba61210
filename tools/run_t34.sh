cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3/trunk16 -o trunk -- python3 tools/prof_layers.py trunk_f16 --reps 5 > gpurun_out/prof_r3/trunk16.log 2>&1
tail -1 gpurun_out/prof_r3/trunk16.log
