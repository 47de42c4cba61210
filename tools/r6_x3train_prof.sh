set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in off on; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/x3p_$m -o p -- python3 $R/tools/r6_x3train.py step $m > $R/gpurun_out/x3p_$m.log 2>&1
done
