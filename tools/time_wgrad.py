#!/usr/bin/env python3
"""Times the weight-gradient kernels on cfg2's k3 layers: Winograd-domain (stride 1) / 12-wave (stride 2) forms vs the
direct tap-split forms (SNVC_ALGO_DIRECT)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from snvc_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
for name, cin, cout, shape in (("conv1 right 32->32", 32, 32, (192, 96, 312)), ("conv1 full 64->32", 64, 32, (192, 96, 312)),
                               ("hg conv2 64->64", 64, 64, (96, 48, 156))):
    x = torch.randn((1, cin) + shape, device=dev)
    g = torch.randn((1, cout) + shape, device=dev)
    gf = 2.0 * x[0, 0].numel() * cin * cout * 27 / 1e9
    for label, bits in (("winograd", 0), ("direct", _lib.ALGO_DIRECT)):
        with ops.conv_variant(bits):
            ms, _ = bench.timed_ms(lambda: ops.conv3d_wgrad(x, g, 3, 1, 1, 1), 5)
        print(f"wgrad {name:20s} {label:9s}: {ms:7.3f} ms  {gf / ms:7.1f} TFLOP/s algorithmic", flush=True)
    del x, g
for name, cin, cout, shape in (("hg conv1 s2 32->64", 32, 64, (192, 96, 312)), ("hg conv3 s2 64->64", 64, 64, (96, 48, 156))):
    x = torch.randn((1, cin) + shape, device=dev)
    g = torch.randn((1, cout) + tuple(v // 2 for v in shape), device=dev)
    gf = 2.0 * g[0, 0].numel() * cin * cout * 27 / 1e9
    for label, bits in (("12-wave", 0), ("direct", _lib.ALGO_DIRECT)):
        with ops.conv_variant(bits):
            ms, _ = bench.timed_ms(lambda: ops.conv3d_wgrad(x, g, 3, 2, 1, 1), 5)
        print(f"wgrad {name:20s} {label:9s}: {ms:7.3f} ms  {gf / ms:7.1f} TFLOP/s", flush=True)
    del x, g
