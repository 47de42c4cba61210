#!/usr/bin/env python3
"""The folded one-channel transposed tail of the cfg2 step (ConvTranspose3d(64, 1, k3, s2, p1, op1) + residual plane):
launch time and a float64 check on a crop.   python tools/time_tail.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import bench  # noqa: E402
from snvc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
torch.manual_seed(0)
h = (bench.D // 2, bench.H // 2, bench.W // 2)
x = torch.randn(1, 64, *h, device=dev)
w = torch.randn(64, 1, 3, 3, 3, device=dev) * 0.05
res = torch.randn(1, 1, bench.D, bench.H, bench.W, device=dev)
lay = ops.Conv3dLayer(w, 3, 2, 1, 1, True)
y = lay(x, None, None, res, ops.EPI_ADD_PRE)
ref = F.conv_transpose3d(x[:, :, :6, :6, :12].double().cpu(), w.double().cpu(), None, 2, 1, 1)[:, :, :8, :8, :16] + res[:, :, :8, :8, :16].double().cpu()
print("max|err| vs float64 on a crop:", (y[:, :, :8, :8, :16].double().cpu() - ref).abs().max().item())
out = torch.empty_like(y)
ms, _ = bench.timed_ms(lambda: lay(x, None, None, res, ops.EPI_ADD_PRE, out=out), 100, 10)
print(f"deconv 64->1 on {h}: {ms * 1e3:.1f} us  ({(x.numel() + 2 * y.numel()) * 4 / ms / 1e6:.0f} GB/s)", flush=True)
