cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r3; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/pair -o pair -- python3 tools/prof_layers.py pair --reps 10 > $O/pair.log 2>&1
tail -1 $O/pair.log; ls $O/pair
