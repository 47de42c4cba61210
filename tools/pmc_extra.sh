# Extra SQ counter passes for one layer of tools/prof_layers.py (run through gpurun): LAYER=conv1_factored bash tools/pmc_extra.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_extra; mkdir -p $O
L=${LAYER:-conv1_factored}
run() { T=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/${L}_$T -o p -- python3 tools/prof_layers.py $L --reps 2 > $O/${L}_$T.log 2>&1
  echo "$L $T rc=$?"; }
run a SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run d SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
python3 - <<'PY'
import csv,glob,os
L=os.environ.get("LAYER","conv1_factored")
for f in sorted(glob.glob(f"gpurun_out/pmc_extra/{L}_*/**/*counter_collection.csv", recursive=True)):
    agg={}
    for r in csv.DictReader(open(f)):
        if "conv3d" in r["Kernel_Name"] or "wgrad" in r["Kernel_Name"] or "deconv" in r["Kernel_Name"]:
            agg.setdefault((r["Kernel_Name"][:70], r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, sum(v)/len(v), len(v))
PY
