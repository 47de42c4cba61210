#!/usr/bin/env python3
"""Local (V-A) model configs of BASELINE.json, fp32, 1 GPU: feature->voxel gather + 3D trunk.
  cfg3: RoI crops 96x96x96, F=32 (64 crops are sharded 8 per GPU; here `--crops` per call)
  cfg5: high-res 80x160x160, F=64 (3D trunk only; the reference's BEV reshape cannot be built for
        F != 32, vernier.py:290-295)
  rel : released shape 32x128x192, F=32
Prints crops/s and the trunk's TFLOP/s (algorithmic flops from SURVEY.md section 8d)."""
import argparse
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from snvc_amd.models.vernier import VernierScale  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("cfg", choices=["cfg3", "cfg5", "rel"])
ap.add_argument("--crops", type=int, default=2)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--precision", choices=["f32", "f16"], default="f32", help="f16: the fp16-storage mode (C8 half tensors)")
ap.add_argument("--no-heads", action="store_true", help="skip timing the stock-PyTorch 2D neck (keeps MIOpen's find kernels out of a profile)")
args = ap.parse_args()
dev = torch.device("cuda:0")
grid, F, gflop = {"cfg3": ((96, 96, 96), 32, 1907.3), "cfg5": ((80, 160, 160), 64, 17652.9),
                  "rel": ((32, 128, 192), 32, 1695.4)}[args.cfg]
cfg = types.SimpleNamespace(vernier_type="BEV_type3", backbone="hrfeat", gn=False,
                            grid_resolution=[32, grid[1], 192], resolution=(256, 256),
                            x_range=(-1.0, 1.0), z_range=(-1.0, 1.0), num_parts=9)
cfg.hrfeat = types.SimpleNamespace(output_channel=F, name="identity")
cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
m = VernierScale(cfg)
m.load_state_dict(bench.seeded_state(m))
m.eval().to(dev)
n = args.crops
r = np.random.default_rng(5)
v = grid[0] * grid[1] * grid[2]
lf = torch.from_numpy(r.standard_normal((n, F, 64, 64)).astype(np.float32)).to(dev)
rf = torch.from_numpy(r.standard_normal((n, F, 64, 64)).astype(np.float32)).to(dev)
gl = torch.from_numpy(r.uniform(-8, 264, (n, 2, v)).astype(np.float32)).to(dev)
gr = torch.from_numpy(r.uniform(-8, 264, (n, 2, v)).astype(np.float32)).to(dev)


def step():
    with torch.no_grad():
        if args.precision == "f16":
            return m.trunk_3d_f16(m.construct_voxel_f16(lf, rf, gl, gr))
        vox = m.construct_voxel(lf, rf, gl, gr)
        return m.trunk_3d(vox)


step(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(args.reps):
    bev, occ, _ = step()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / args.reps
assert torch.isfinite(bev).all()
if F == 32 and grid[0] == 32 and not args.no_heads:   # the 2D BEV neck + heads that follow the path
    with torch.no_grad():
        m.heads_2d(bev); torch.cuda.synchronize()
        a.record()
        for _ in range(args.reps):
            hm, co = m.heads_2d(bev)
        b.record(); torch.cuda.synchronize()
    print(f"{args.cfg}: 2D neck + heads: {a.elapsed_time(b) / args.reps / n:.2f} ms/crop")
print(f"{args.cfg} [{args.precision}]: grid {grid} F={F} crops/call={n}: {ms / n:.2f} ms/crop = {1e3 * n / ms:.1f} crops/s/GPU, "
      f"{gflop * n / ms:.1f} TFLOP/s ({100 * gflop * n / ms / 157.3:.0f}% of fp32 MFMA), "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
