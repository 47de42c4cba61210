#!/usr/bin/env python3
"""Times the fp16-storage conv family on cfg5's layers (grid 80x160x160, F = 64) and prints TFLOP/s against the
dense fp16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).  Events on the launch stream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from snvc_amd import ops  # noqa: E402
from snvc_amd.models import submodule as S  # noqa: E402

dev = torch.device("cuda:0")
PEAK = 2500.0
only = sys.argv[1:]


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def case(name, cin, cout, k, s, dil, shape, transposed=False, res=False):
    if only and not any(o in name for o in only):
        return
    pad = dil * (k - 1) // 2
    m = (S._deconvbn_3d(cin, cout, False) if transposed else S.convbn_3d(cin, cout, k, s, pad, dilation=dil)).to(dev).eval()
    x = torch.randn((1, cin // 8) + shape + (8,), device=dev).half()
    with torch.no_grad():
        y = m.fused_f16(x, relu=True)
        r = torch.randn_like(y) if res else None
        ms = timed(lambda: m.fused_f16(x, relu=True, residual=r, residual_after_act=True))
    vox = (x.shape[2] * x.shape[3] * x.shape[4]) if transposed else (y.shape[2] * y.shape[3] * y.shape[4])
    gf = 2.0 * vox * cin * cout * (27 if transposed else k ** 3) / 1e9
    print(f"{name:46s} {ms:8.3f} ms  {gf / ms:8.1f} TFLOP/s  {100 * gf / ms / PEAK:5.1f}% of fp16 dense peak", flush=True)


G = (80, 160, 160)
case("conv1 k7 128->64 80x160x160", 128, 64, 7, 1, 1, G)
case("conv2 k5 64->64 (+res)", 64, 64, 5, 1, 1, G, res=True)
case("conv3 k5 dil2 64->64 (+res)", 64, 64, 5, 1, 2, G, res=True)
case("conv4 k3 128->64", 128, 64, 3, 1, 1, G)
case("fg_cls_head[0] k3 64->64", 64, 64, 3, 1, 1, G)
case("vimg_feat k1 128->64", 128, 64, 1, 1, 1, G)
case("hg conv1 k3s2 64->128", 64, 128, 3, 2, 1, G)
case("hg conv2 k3 128->128 40x80x80", 128, 128, 3, 1, 1, (40, 80, 80))
case("hg conv12 deconv 128->64 40x80x80", 128, 64, 3, 2, 1, (40, 80, 80), transposed=True, res=True)
