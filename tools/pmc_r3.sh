# SQ counter passes for one layer of tools/prof_layers.py (run through gpurun):
#   LAYER=gather_f16 MATCH=gather bash tools/pmc_r3.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_r3; mkdir -p $O
L=${LAYER:-gather_proj}
run() { T=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/${L}_$T -o p -- python3 tools/prof_layers.py $L --reps 2 > $O/${L}_$T.log 2>&1
  echo "$L $T rc=$?"; }
run a SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
run d SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES
run e GRBM_GUI_ACTIVE
if [ -n "$TRAFFIC" ]; then run f FETCH_SIZE; run g WRITE_SIZE; fi
python3 - <<'PY'
import csv,glob,os
L=os.environ.get("LAYER","gather_proj"); M=os.environ.get("MATCH","gather")
for f in sorted(glob.glob(f"gpurun_out/pmc_r3/{L}_*/**/*counter_collection.csv", recursive=True)):
    agg={}
    for r in csv.DictReader(open(f)):
        if M in r["Kernel_Name"]:
            agg.setdefault((r["Kernel_Name"].replace("snvc::(anonymous namespace)::","")[:60], r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, sum(v)/len(v), len(v))
PY
