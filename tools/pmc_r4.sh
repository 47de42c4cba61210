# SQ counter passes for one layer of tools/prof_layers.py (run through gpurun):
#   LAYER=x3_conv2 MATCH=conv3d_f16 bash tools/pmc_r4.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_r4; mkdir -p $O
L=${LAYER:-x3_conv2}
run() { T=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/${L}_$T -o p -- python3 tools/prof_layers.py $L --reps 2 > $O/${L}_$T.log 2>&1
  echo "$L $T rc=$?"; }
run a SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
run d SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES
run e SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
run h GRBM_GUI_ACTIVE
if [ -n "$TRAFFIC" ]; then run f FETCH_SIZE; run g WRITE_SIZE; fi
python3 - <<'PY'
import csv,glob,os
L=os.environ.get("LAYER","x3_conv2"); M=os.environ.get("MATCH","conv3d_f16")
for f in sorted(glob.glob(f"gpurun_out/pmc_r4/{L}_*/**/*counter_collection.csv", recursive=True)):
    agg={}
    for r in csv.DictReader(open(f)):
        if M in r["Kernel_Name"]:
            agg.setdefault((r["Kernel_Name"].replace("snvc::(anonymous namespace)::","")[:50], r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, sum(v)/len(v), len(v))
PY
