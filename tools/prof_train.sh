# Kernel-trace stats of the training step (bench.py --mode train); run through gpurun from the repo root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_train; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 bench.py --mode train --steps 10 --warmup 3 > $O/train.log 2>&1
tail -2 $O/train.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_train/train/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:45]:
    print(f'{r["Name"][:120]:120s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
