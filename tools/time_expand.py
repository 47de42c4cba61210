#!/usr/bin/env python3
"""Times the first layer's expand passes at cfg2 size (0.74 GB written per launch): snvc_warped_expand in both kernel forms
on several shift patterns, beside snvc_sheared_expand.   python tools/time_expand.py [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from snvc_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--d", type=int, default=bench.D)
args = ap.parse_args()
dev = torch.device("cuda:0")
C, H, W, D = bench.C, bench.H, bench.W, args.d
p = torch.randn(1, 3 * C, H, W, device=dev)
q = torch.randn(1, 3 * C, H, W, device=dev)
e = torch.randn(1, 9 * C, H, 4, device=dev)
planes = torch.randn(1, C, 3, H, W, device=dev)
scale, bias = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
out = torch.empty(1, C, D, H, W, device=dev)
d_ = np.arange(D, dtype=np.float64)
r = np.random.default_rng(0)
patterns = {"cfg2 d/2": d_ / 2, "cfg1 d+0.5odd": d_ + 0.5 * (d_ % 2), "quarter": d_ / 4, "whole": d_.copy(), "0.7d": 0.7 * d_,
            "down 100-d/2": 100.0 - d_ / 2, "random": r.uniform(0, 100, D), "constant 3.5": np.full(D, 3.5)}
nbytes = out.numel() * 4
from snvc_amd.models.submodule import sheared_geometry  # noqa: E402
for q_, m0 in ((2, 0), (1, 0)):
    off, wu, off_col, wu_col = sheared_geometry(q_, m0, D, W)
    g = torch.randn(1, 3 * C, H, wu, device=dev)
    gcol = torch.randn(1, 3 * C, H, wu_col, device=dev)
    ms, _ = bench.timed_ms(lambda: ops.sheared_expand(g, gcol, planes, scale, bias, out, q_, m0, off, off_col, ops.EPI_RELU), args.reps, 5)
    print(f"sheared q={q_}        {ms * 1e3:7.1f} us {nbytes / ms / 1e9:6.2f} TB/s ({nbytes / ms / 1e9 / 8 * 100:4.1f} %)", flush=True)
    outs = torch.empty(1, 2, C // 8, D, H, W, 8, device=dev, dtype=torch.float16)
    ms, _ = bench.timed_ms(lambda: ops.sheared_expand_split(g, gcol, planes, scale, bias, outs, q_, m0, off, off_col, ops.EPI_RELU), args.reps, 5)
    print(f"sheared q={q_} split  {ms * 1e3:7.1f} us {nbytes / ms / 1e9:6.2f} TB/s ({nbytes / ms / 1e9 / 8 * 100:4.1f} %)", flush=True)
    del outs
ms, _ = bench.timed_ms(lambda: out.fill_(1.0), args.reps, 5)
print(f"torch fill_         {ms * 1e3:7.1f} us {nbytes / ms / 1e9:6.2f} TB/s ({nbytes / ms / 1e9 / 8 * 100:4.1f} %)", flush=True)
for name, s in patterns.items():
    sh = torch.from_numpy(s.astype(np.float32)[None].copy()).to(dev)
    row = []
    for form in (0, ops.WARPED_EXPAND_R3):
        ops.WARPED_EXPAND_FORM[0] = form
        ms, _ = bench.timed_ms(lambda: ops.warped_expand(p, q, e, planes, sh, scale, bias, out, ops.EPI_RELU), args.reps, 5)
        row.append(f"{'r3 ' if form else 'win'} {ms * 1e3:7.1f} us {nbytes / ms / 1e9:6.2f} TB/s ({nbytes / ms / 1e9 / 8 * 100:4.1f} %)")
    ops.WARPED_EXPAND_FORM[0] = 0
    print(f"{name:16s} " + "   ".join(row), flush=True)
