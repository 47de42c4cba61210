# Round-3 counter passes (run through gpurun from the repo root).  One counter group per pass (FETCH_SIZE and
# WRITE_SIZE cannot share one; --pmc is never combined with a trace).  `python tools/make_traffic_json.py r3` turns the
# CSVs into profiles/r3/traffic.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_r3; mkdir -p $O
run() {  # layer, tag, counters...
  L=$1; T=$2; shift 2
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/${L}_$T -o p -- python3 tools/prof_layers.py $L --reps 2 > $O/${L}_$T.log 2>&1
  echo "$L $T rc=$?"
}
for L in ${LAYERS:-conv2_side hg_s2 sheared gather_cfg3 gather_f16 f16_k7_32 cost_volume_right}; do
  run $L fetch FETCH_SIZE
  run $L write WRITE_SIZE
done
for L in ${SQ_LAYERS:-conv2_side hg_s2 f16_k7_32}; do
  run $L sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY
  run $L sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
done
find $O -name "*counter_collection.csv" | wc -l
