cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r3; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/heads -o heads -- python3 tools/prof_heads.py --crops 2 > $O/heads.log 2>&1
tail -2 $O/heads.log
