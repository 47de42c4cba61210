cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r4; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/local -o local -- python3 tools/prof_layers.py trunk --reps 5 > $O/local.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_r4/local/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{r["Name"][:120]:120s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
print("total ms", tot / 1e6)
PY
