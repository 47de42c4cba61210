# Round-6 counter passes (run through gpurun from the repo root).  One counter group per pass (FETCH_SIZE and WRITE_SIZE cannot share
# one; --pmc is never combined with a trace).  `python tools/make_traffic_json.py r6` turns the CSVs into profiles/r6/traffic.json (tools/collect_r6.py folds it in) and
# records the hash of conv3d_x3q_kernel's source, which bench.py checks before quoting `roofline.traffic`.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_r6; mkdir -p $O
run() {  # layer (output name), prof_layers layer, tag, counters...
  L=$1; P=$2; T=$3; shift 3
  timeout -k 10 120 rocprofv3 --pmc "$@" --output-format csv -d $O/${L}_$T -o p -- python3 tools/prof_layers.py $P --reps 2 > $O/${L}_$T.log 2>&1
  echo "$L $T rc=$?"
}
for pair in ${PAIRS:-x3_conv2:x3_conv2 x3_hg2:x3_hg2 x3_s2:x3_s2 x3_hg5_tail:x3_hg5_tail tail_gather:tail_gather sheared_split:sheared general_split:general}; do
  L=${pair%%:*}; P=${pair##*:}
  run $L $P fetch FETCH_SIZE
  run $L $P write WRITE_SIZE
done
for L in x3_conv2 x3_hg2 x3_s2; do
  run $L $L sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY
done
find $O -name "*counter_collection.csv" | wc -l
