# Round-6 profiles (run through gpurun from the repo root): kernel-trace stats of the SAME commands the bench lines come from, and
# separate --pmc passes (never combined with a trace) for the round's new kernels.  Every step must succeed before the next one
# starts (a faulting box must not be driven further); summaries are copied to profiles/r6/ by tools/collect_r6.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r6; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
echo "bench ok"
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 bench.py --mode train --steps 20 --warmup 3 > $O/train.log 2>&1
echo "train ok"
for shape in full s2; do
  timeout -k 10 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/wg_${shape}_fetch -o p -- python3 tools/r6_wgrad_prof.py $shape > $O/wg_${shape}_fetch.log 2>&1
  timeout -k 10 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/wg_${shape}_write -o p -- python3 tools/r6_wgrad_prof.py $shape > $O/wg_${shape}_write.log 2>&1
  timeout -k 10 60 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/wg_${shape}_sq -o p -- python3 tools/r6_wgrad_prof.py $shape > $O/wg_${shape}_sq.log 2>&1
  echo "$shape pmc ok"
done
find $O -name "*.csv" | wc -l
