# usage: bash tools/r6_x3train_prof.sh <tag> <args of tools/r6_x3train.py ...>   -> gpurun_out/x3p_<tag>/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/x3p_$tag -o p -- python3 $R/tools/r6_x3train.py "$@" > $R/gpurun_out/x3p_$tag.log 2>&1
