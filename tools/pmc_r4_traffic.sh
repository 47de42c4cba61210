# Round-4 counter passes (run through gpurun from the repo root).  One counter group per pass (FETCH_SIZE and WRITE_SIZE
# cannot share one; --pmc is never combined with a trace).  `python tools/make_traffic_json.py r4` turns the CSVs into
# profiles/r4/traffic.json.  Layer names are tools/prof_layers.py's; sheared_split / general_split run forward_pair in
# split mode (the default), general_f32 / conv2_side on the fp32 kernels.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_r4; mkdir -p $O
run() {  # layer (output name), prof_layers layer, tag, counters...
  L=$1; P=$2; T=$3; shift 3
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/${L}_$T -o p -- python3 tools/prof_layers.py $P --reps 2 > $O/${L}_$T.log 2>&1
  echo "$L $T rc=$?"
}
# PAIRS="x3_conv2:x3_conv2 x3_hg2:x3_hg2" re-collects the split-mode layers only (late r4: their 16x16x32 form)
for pair in ${PAIRS:-x3_conv2:x3_conv2 x3_hg2:x3_hg2 sheared_split:sheared general_split:general general_f32:general_f32 conv2_side:conv2_side}; do
  L=${pair%%:*}; P=${pair##*:}
  run $L $P fetch FETCH_SIZE
  run $L $P write WRITE_SIZE
done
for L in x3_conv2 x3_hg2; do
  run $L $L sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY
  run $L $L sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
done
find $O -name "*counter_collection.csv" | wc -l
