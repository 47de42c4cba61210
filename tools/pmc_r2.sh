# Round-2 counter passes (run through gpurun from the repo root).  One counter group per pass (FETCH_SIZE and
# WRITE_SIZE cannot share one; --pmc is never combined with a trace).  tools/make_traffic_json.py turns the CSVs
# into profiles/r2/traffic.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_r2; mkdir -p $O
run() {  # layer, tag, counters...
  L=$1; T=$2; shift 2
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/${L}_$T -o p -- python3 tools/prof_layers.py $L --reps 2 > $O/${L}_$T.log 2>&1
  echo "$L $T rc=$?"
}
for L in ${LAYERS:-conv1_factored cost_volume cost_volume_right cost_volume_bwd gather_proj gather_uniform gather_f16 f16_k7 f16_k5}; do
  run $L fetch FETCH_SIZE
  run $L write WRITE_SIZE
done
for L in ${SQ_LAYERS:-conv1_factored f16_k7 f16_k5}; do
  run $L sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY
  run $L sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
done
find $O -name "*counter_collection.csv" | head -40
