cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3/train2 -o train -- python3 bench.py --mode train --steps 10 --warmup 3 > gpurun_out/prof_r3/train2.log 2>&1
grep -E "sheared_bwd|sheared_expand" gpurun_out/prof_r3/train2/train_kernel_stats.csv | cut -c1-60,150-260
