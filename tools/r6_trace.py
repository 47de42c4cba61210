"""Print the last step's kernels of a rocprofv3 kernel trace (gpurun_out/x3p_<tag>): duration, grid, name."""
import csv, glob, sys
tag, pat, last = sys.argv[1], sys.argv[2].split(","), int(sys.argv[3])
f = glob.glob(f"gpurun_out/x3p_{tag}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(p in r["Kernel_Name"] for p in pat)]
for r in sel[-last:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"].replace("snvc::(anonymous namespace)::", "").replace("void ", "")
    print(f"{d:8.1f} us  grid {r['Grid_Size_X']:>9}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} {n[:100]}")
