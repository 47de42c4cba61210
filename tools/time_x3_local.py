#!/usr/bin/env python3
"""Split-mode (f16x3) forms of the local trunk's layer kinds against the fp32 Winograd kernels at the released shape (2 crops of
32x128x192): accuracy against float64 on a crop, launch times.   python tools/time_x3_local.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import bench  # noqa: E402
from snvc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
shape = (32, 128, 192)
for name, cin, cout, k, dil in (("conv1 k7 64->32", 64, 32, 7, 1), ("conv2 k5 32->32", 32, 32, 5, 1), ("conv3 k5 dil2 32->32", 32, 32, 5, 2),
                                ("vimg_feat k1 64->32", 64, 32, 1, 1), ("conv4 k3 64->32", 64, 32, 3, 1)):
    pad = dil * (k - 1) // 2
    x = torch.relu(torch.randn(2, cin, *shape, device=dev)) * 1.5
    w = torch.randn(cout, cin, k, k, k, device=dev) * np.sqrt(2.0 / (cin * k ** 3))
    scale, bias = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2
    ref_layer = ops.Conv3dLayer(w, k, 1, pad, dil, False)
    x3 = ops.Conv3dLayerX3(w, k, 1, pad, dil, False)
    xs = ops.to_split(x, 4)
    y_ref = ref_layer(x, scale, bias, None, ops.EPI_RELU)
    ys = x3(xs, 4, scale, bias, flags=ops.EPI_RELU, out_exp=4)
    y_rt = ops.from_split(ys, 4)
    r = pad
    d0, h0, w0 = 12, 40, 100
    crop = x[:1, :, d0 - r:d0 + 4 + r, h0 - r:h0 + 4 + r, w0 - r:w0 + 40 + r].double().cpu()
    y64 = torch.relu(F.conv3d(crop, w.double().cpu(), None, 1, 0, dil) * scale.double().cpu().view(1, -1, 1, 1, 1) + bias.double().cpu().view(1, -1, 1, 1, 1))
    sl = (slice(0, 1), slice(None), slice(d0, d0 + 4), slice(h0, h0 + 4), slice(w0, w0 + 40))
    rng = y64.abs().max().item()
    e = lambda t: (t[sl].double().cpu() - y64).abs().max().item() / rng       # noqa: E731
    flop = 2.0 * 2 * np.prod(shape) * cin * cout * k ** 3
    ms_r, _ = bench.timed_ms(lambda: ref_layer(x, scale, bias, None, ops.EPI_RELU, y_ref), 10, 3)
    ms_x, _ = bench.timed_ms(lambda: x3(xs, 4, scale, bias, flags=ops.EPI_RELU, out=ys, out_exp=4), 10, 3)
    print(f"{name:24s} fp32 default {ms_r / 2:7.3f} ms/crop (err {e(y_ref):.1e})   split {ms_x / 2:7.3f} ms/crop (err {e(y_rt):.1e}, "
          f"{flop / ms_x / 1e9:6.0f} TFLOP/s algorithmic)   whole tensor split vs fp32: {(y_rt - y_ref).abs().max().item() / y_ref.abs().max().item():.1e}", flush=True)
    del x, xs, ys, y_ref, y_rt
