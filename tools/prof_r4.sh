# Round-4 profiles (run through gpurun from the repo root): kernel-trace stats of the bench step.
# Summaries are copied into profiles/r4/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r4; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench.log 2>&1
ls $O/*/*
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_r4/bench/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:40]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
