#!/usr/bin/env python3
"""The 16x16x32 form of the split-mode 3x3x3 layers at cfg2's sizes: launch times on random / all-zero operands (the power probe),
and a float64 check on ragged shapes.   python tools/time_x3q.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from snvc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
torch.manual_seed(1)
for (cin, cout, shp) in ((16, 32, (9, 11, 45)), (32, 64, (12, 8, 70)), (8, 96, (5, 6, 33))):
    x = torch.randn(2, cin, *shp, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.1
    lay = ops.Conv3dLayerX3(w, algo=ops._lib.ALGO_X3_Q16)
    y = ops.from_split(lay(ops.to_split(x, 2), 2, out_exp=2), 2)
    ref = torch.nn.functional.conv3d(x.double().cpu(), w.double().cpu(), padding=1)
    print(cin, cout, shp, "max|err| / max|ref| vs float64:", (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item(), flush=True)
full, half = (bench.D, bench.H, bench.W), (bench.D // 2, bench.H // 2, bench.W // 2)
for name, cin, cout, shp in (("conv2 32->32", 32, 32, full), ("32->64 full", 32, 64, full), ("hg conv2 64->64", 64, 64, half)):
    for kind in ("random", "zeros"):
        xin = torch.relu(torch.randn(1, cin, *shp, device=dev)) if kind == "random" else torch.zeros(1, cin, *shp, device=dev)
        wt = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05) if kind == "random" else torch.zeros(cout, cin, 3, 3, 3, device=dev)
        lay = ops.Conv3dLayerX3(wt, algo=ops._lib.ALGO_X3_Q16)
        xs_ = ops.to_split(xin, 4)
        del xin
        ys_ = torch.empty(1, 2, cout // 8, *shp, 8, dtype=torch.float16, device=dev)
        flag_ = torch.zeros(1, dtype=torch.int32, device=dev)
        ms, _ = bench.timed_ms(lambda: lay(xs_, 4, flags=ops.EPI_RELU, out=ys_, out_exp=4, overflow=flag_), 50, 5)
        print(f"{name:18s} {kind:7s} {ms:7.3f} ms", flush=True)
        del xs_, ys_
