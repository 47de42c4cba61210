# Kernel-trace stats of the cfg4 training step (run through gpurun from the repo root); the summary is copied into profiles/r3/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r3; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 bench.py --mode train --steps 10 --warmup 3 > $O/train.log 2>&1
ls $O/train/*
