#!/usr/bin/env python3
"""profiles/r6/ from gpurun_out/prof_r6 (tools/prof_r6.sh): the kernel-stats summaries of `python bench.py` and of
`python bench.py --mode train`, the bench lines those same runs printed, and the --pmc passes of the round's new kernels
(split-operand weight gradients) as traffic / pipe figures.  HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE KiB (the gfx950 correction of
MI355X_MICROARCH.md's HBM section for 16-byte streaming reads), per launch, last dispatch of the kernel."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import emit
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof_r6"), os.path.join(ROOT, "profiles", "r6")
os.makedirs(DST, exist_ok=True)
for tag in ("bench", "train"):
    shutil.copy(os.path.join(SRC, tag, f"{tag}_kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats.csv"))
    log = open(os.path.join(SRC, f"{tag}.log")).read()
    det = [l for l in log.splitlines() if l.startswith("[bench_detail] ")]
    if det:
        json.dump(json.loads(det[-1][len("[bench_detail] "):]), open(os.path.join(DST, f"{tag}_line_detail.json"), "w"), indent=1)
    last = emit.last_json_line(log)
    assert last is not None and "metric" in last, tag
    json.dump(last, open(os.path.join(DST, f"{tag}_line.json"), "w"), indent=1)

def last_rows(path, needle):
    rows = [r for r in csv.DictReader(open(path)) if needle in r["Kernel_Name"]]
    out = {}
    for r in rows:
        out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: v[-1] for k, v in out.items()}

ALG = {"full": ("conv3d_wgrad_x3_kernel", 2 * 735902208 + 27 * 32 * 32 * 4, 317995352064.0, "32->32 on 192x96x312 (conv2's weight gradient)"),
       "s2": ("conv3d_wgrad_x3s2_kernel", 735902208 + 183975552 * 2 // 2 * 2 + 27 * 64 * 32 * 4, 79498838016.0, "32->64, stride 2, x 192x96x312 (hg conv1's weight gradient)")}
ALG["s2"] = ("conv3d_wgrad_x3s2_kernel", 735902208 + 367951104 + 27 * 64 * 32 * 4, 79498838016.0, ALG["s2"][3])       # x 32 ch full + g 64 ch half
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(DST, "train_kernel_stats.csv")))}
res = {}
for shape, (kern, alg_bytes, flop, what) in ALG.items():
    f = last_rows(glob.glob(os.path.join(SRC, f"wg_{shape}_fetch", "**", "*counter_collection.csv"), recursive=True)[0], kern)
    w = last_rows(glob.glob(os.path.join(SRC, f"wg_{shape}_write", "**", "*counter_collection.csv"), recursive=True)[0], kern)
    s = last_rows(glob.glob(os.path.join(SRC, f"wg_{shape}_sq", "**", "*counter_collection.csv"), recursive=True)[0], kern)
    hbm = (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024
    cyc = s["GRBM_GUI_ACTIVE"] / 8.0                       # per XCD
    res[shape] = {"kernel": kern, "layer": what, "hbm_bytes_corrected": hbm, "fetch_kib": f["FETCH_SIZE"], "write_kib": w["WRITE_SIZE"],
                  "algorithmic_bytes": alg_bytes, "traffic_over_algorithmic": hbm / alg_bytes, "flop_algorithmic": flop,
                  "mfma_insts": s["SQ_INSTS_MFMA"], "mfma_pipe_frac_all_simds": s["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                  "gpu_cycles_per_xcd": cyc, "wait_any_frac": s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"],
                  "wait_inst_any_frac": s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"], "active_inst_frac": s["SQ_ACTIVE_INST_ANY"] / s["SQ_WAVE_CYCLES"]}
    for name, r in stats.items():
        if kern in name and ("<true>" in name or "<" not in name):
            res[shape]["avg_ms_in_the_training_step"] = float(r["AverageNs"]) / 1e6
# layers.*: tools/pmc_r6_traffic.sh + `python tools/make_traffic_json.py r6` (moved to traffic_layers.json) -- re-collected in r6 because
# conv3d_x3q_kernel gained its float32-output instance (bench.py ties `roofline.traffic` to the kernel's source hash)
layers = os.path.join(DST, "traffic_layers.json")
prev = json.load(open(layers if os.path.exists(layers) else os.path.join(ROOT, "profiles", "r5", "traffic.json")))
prev["wgrad_x3"] = res
prev["note_r6"] = ("layers.*: tools/pmc_r6_traffic.sh + tools/make_traffic_json.py r6 (round 6, after conv3d_x3q_kernel's float32-output instance was "
                   "added; bench.py checks the kernel source hash); wgrad_x3: tools/prof_r6.sh + tools/collect_r6.py, the weight-gradient "
                   "launches with the maxima supplied as in the training step")
json.dump(prev, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
