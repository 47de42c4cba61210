#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc CSVs of tools/pmc_r2.sh (gpurun_out/pmc_r2/<layer>_<tag>/.../p_counter_collection.csv)
into profiles/r2/traffic.json and copies the per-layer CSV rows of the profiled kernel into profiles/r2/pmc/.
HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB counters; the factor 2 is the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md's HBM section for 16-byte-per-lane streaming reads)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r2"          # python tools/make_traffic_json.py r3
SRC = os.path.join(ROOT, "gpurun_out", f"pmc_{ROUND}")
DST = os.path.join(ROOT, "profiles", ROUND)
KERNEL = {  # layer -> substring of the kernel whose LAST dispatch is reported
    "conv1_factored": "conv3d_wino_dma_kernel", "cost_volume": "cost_volume_fwd_rows", "cost_volume_right": "cost_volume_fwd_rows",
    "cost_volume_bwd": "cost_volume_bwd_rows_f32", "gather_proj": "voxel_gather_fwd_lds", "gather_uniform": "voxel_gather_fwd_lds",
    "gather_f16": "voxel_gather_fwd_lds", "f16_k7": "conv3d_f16_kernel", "f16_k5": "conv3d_f16_kernel",
    # r3
    "conv2_side": "conv3d_wino_dma_kernel", "hg_s2": "conv3d_winos2_pipe_kernel", "sheared": "sheared_expand_kernel",
    "gather_cfg3": "voxel_gather_fwd_lds", "f16_k7_32": "conv3d_f16_kernel",
    "general": "warped_expand_kernel", "sheared_bwd": "sheared_bwd_kernel",
    # r4: split-mode (f16x3) layers and the split-output expand passes; the fp32 any-shift expand in its register-window form
    "x3_conv2": "conv3d_x3q_kernel", "x3_hg2": "conv3d_x3q_kernel", "sheared_split": "sheared_expand_split_kernel",
    "general_split": "warped_expand_split_kernel", "general_f32": "warped_expand_win_kernel",
}
KERNEL.update({"x3_s2": "conv3d_x3s2q_kernel", "x3_hg5_tail": "conv3d_f16_kernel", "tail_gather": "deconv_tail_gather_kernel"})     # r5
if ROUND == "r4":
    KERNEL = {k: v for k, v in KERNEL.items() if k in ("x3_conv2", "x3_hg2", "sheared_split", "general_split", "general_f32", "conv2_side")}
if ROUND >= "r5":
    KERNEL = {k: v for k, v in KERNEL.items() if k in ("x3_conv2", "x3_hg2", "x3_s2", "x3_hg5_tail", "tail_gather", "sheared_split", "general_split")}
ALGORITHMIC = {  # bytes per launch (SURVEY.md section 8d formulas)
    "conv1_factored": 1472200704, "cost_volume": 1479869184, "cost_volume_right": 739934976, "cost_volume_bwd": 1479869184 + 2 * 3833856,
    "gather_proj": 2 * (786432 * 272 + 2 * 32 * 4096 * 4), "gather_uniform": 2 * (786432 * 272 + 2 * 32 * 4096 * 4),
    "gather_f16": 2048000 * (16 + 4 * 64) + 2 * 64 * 4096 * 4,
    "f16_k7": 2 * 2048000 * (128 + 64), "f16_k5": 2 * 2048000 * (64 + 64),
    # r3: conv2 reads and writes one 32-channel volume (+ the 1-channel projection); hg conv1 reads 32 channels at full and
    # writes 64 at half resolution; the expand pass writes one 32-channel volume; 8 crops of 96^3
    "conv2_side": 2 * 735902208 + 22996944, "hg_s2": 735902208 + 183975552, "sheared": 735902208,
    "gather_cfg3": 8 * (884736 * 272 + 2 * 32 * 4096 * 4), "f16_k7_32": 2 * 786432 * 2 * (64 + 32),
    # the warp-after-convolution expand writes one 32-channel volume; the sheared layer's fused backward reads one (gy)
    "general": 735902208, "sheared_bwd": 735902208,
    # r4: a split pair is two half planes = the fp32 tensor's bytes; conv2 reads one 32-channel pair, writes one and the 1-channel
    # fp32 projection; hg conv2 reads and writes a 64-channel pair at half resolution
    "x3_conv2": 2 * 735902208 + 22996944, "x3_hg2": 2 * 183975552, "sheared_split": 735902208, "general_split": 735902208,
    "general_f32": 735902208,
    # r5: hg conv1 reads the 32-channel pair at full and writes the 64-channel pair at half resolution; conv5 + tail projection reads
    # the 64-channel pair at quarter resolution and `pre` at half, writes 27 x 8 fp32 class planes; the gather reads those and
    # classifier(v2), writes the cost
    "x3_s2": 735902208 + 183975552, "x3_hg5_tail": 22996944 * 2 + 183975552 + 27 * 8 * 359424 * 4,
    "tail_gather": 27 * 8 * 359424 * 4 + 2 * 22996944,
}
F32_MFMA_LAYERS = ("conv1_factored", "conv2_side", "hg_s2")
SOURCE_OF = {"x3_s2": ("snvc_amd/csrc/conv3d_f16.hip", "conv3d_x3s2q_kernel(const F16Args a, const int total_jobs) {"),
             "x3_conv2": ("snvc_amd/csrc/conv3d_f16.hip", "conv3d_x3q_kernel(const F16Args a) {"),
             "x3_hg2": ("snvc_amd/csrc/conv3d_f16.hip", "conv3d_x3q_kernel(const F16Args a) {")}


def last_dispatch(path, needle):
    rows = [r for r in csv.DictReader(open(path)) if needle in r.get("Kernel_Name", "")]
    if not rows:
        return {}, []
    last = max(int(r["Dispatch_Id"]) for r in rows)
    sel = [r for r in rows if int(r["Dispatch_Id"]) == last]
    return {r["Counter_Name"]: float(r["Counter_Value"]) for r in sel}, sel


out = {"note": __doc__.strip().split("\n")[-3:], "layers": {}}
os.makedirs(os.path.join(DST, "pmc"), exist_ok=True)
for layer, needle in KERNEL.items():
    entry = {"kernel": needle}
    for tag in ("fetch", "write", "sq", "sq2"):
        files = glob.glob(os.path.join(SRC, f"{layer}_{tag}", "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        vals, sel = last_dispatch(files[0], needle)
        entry.update(vals)
        if sel:
            with open(os.path.join(DST, "pmc", f"{layer}_{tag}.csv"), "w", newline="") as fh:
                w = csv.DictWriter(fh, fieldnames=list(sel[0].keys()))
                w.writeheader()
                w.writerows(sel)
    if "FETCH_SIZE" in entry and "WRITE_SIZE" in entry:
        entry["hbm_bytes_corrected"] = 2 * entry["FETCH_SIZE"] * 1024 + entry["WRITE_SIZE"] * 1024
        entry["algorithmic_bytes"] = ALGORITHMIC[layer]
        entry["traffic_over_algorithmic"] = entry["hbm_bytes_corrected"] / ALGORITHMIC[layer]
    if "SQ_INSTS_MFMA" in entry and "GRBM_GUI_ACTIVE" in entry:
        # v_mfma_f32_32x32x2_f32: 64 cycles/SIMD; 32x32x16_f16: 32; 16x16x32_f16 (the split-mode 3x3x3 layers since late r4): 16
        cyc = 64 if layer in F32_MFMA_LAYERS else 16 if layer in ("x3_conv2", "x3_hg2", "x3_s2") else 32
        entry["mfma_pipe_frac"] = entry["SQ_INSTS_MFMA"] * cyc / (1024 * entry["GRBM_GUI_ACTIVE"] / 8)
    if layer in SOURCE_OF and len(entry) > 1:
        sys.path.insert(0, ROOT)
        import bench        # noqa: E402  (kernel_source_hash: the counters are tied to the kernel text they were collected on)
        entry["kernel_source_sha256_16"] = bench.kernel_source_hash(*SOURCE_OF[layer])
    if len(entry) > 1:
        out["layers"][layer] = entry
if "conv1_factored" in out["layers"]:
    out["conv1_right_wino43_dma_k3_32to32_cfg2"] = out["layers"]["conv1_factored"]   # the key bench.py reads
json.dump(out, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
