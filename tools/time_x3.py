#!/usr/bin/env python3
"""Split-mode ("f16x3") 3x3x3 layers against the fp32 Winograd kernels at cfg2's sizes: accuracy against a float64 convolution
on a crop, and launch times.   python tools/time_x3.py [--reps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import bench  # noqa: E402
from snvc_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, cin, cout, shape in (("conv2 32->32 192x96x312", 32, 32, (192, 96, 312)), ("hg conv2 64->64 96x48x156", 64, 64, (96, 48, 156))):
    x = torch.relu(torch.randn(1, cin, *shape, device=dev)) * 1.5
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * np.sqrt(2.0 / (cin * 27))
    scale, bias = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2
    ref_layer = ops.Conv3dLayer(w, 3, 1, 1, 1, False)
    x3 = ops.Conv3dLayerX3(w, algo=0)
    x_exp = 4
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    head = torch.randn(cout, device=dev)
    xs = ops.to_split(x, x_exp)
    y_ref = ref_layer(x, scale, bias, None, ops.EPI_RELU)
    y_x3 = x3(xs, x_exp, scale, bias, flags=ops.EPI_RELU, to_f32=True)
    ys = x3(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out_exp=4)
    y_rt = ops.from_split(ys, 4)
    # float64 reference on a crop (with its halo)
    d0, h0, w0 = 40, 20, 100
    crop = x[:, :, d0 - 1:d0 + 9, h0 - 1:h0 + 9, w0 - 1:w0 + 41].double().cpu()
    y64 = torch.relu(F.conv3d(crop, w.double().cpu()) * scale.double().cpu().view(1, -1, 1, 1, 1) + bias.double().cpu().view(1, -1, 1, 1, 1))
    sl = (slice(None), slice(None), slice(d0, d0 + 8), slice(h0, h0 + 8), slice(w0, w0 + 40))
    rng = y64.abs().max().item()
    e = lambda t: (t[sl].double().cpu() - y64).abs().max().item() / rng       # noqa: E731
    print(f"{name}: max|err| / max|ref| vs float64 on a crop: fp32 Winograd {e(y_ref):.2e}   split -> f32 {e(y_x3):.2e}   split -> split {e(y_rt):.2e}")
    print(f"   whole tensor, split vs fp32 path: {(y_x3 - y_ref).abs().max().item() / y_ref.abs().max().item():.2e}")
    flop = 2.0 * np.prod(shape) * cin * cout * 27
    x3s = ops.Conv3dLayerX3(w, algo=_lib.ALGO_X3_SERIAL)
    ysr = x3s(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out_exp=4)
    print(f"   serial form vs resident form: {(ops.from_split(ysr, 4) - y_rt).abs().max().item() / y_rt.abs().max().item():.2e}")
    extra = []
    if cout == 64:
        for bit, nm in ((_lib.ALGO_X3_NARROW, "32-channel blocks"), (_lib.ALGO_X3_SMALL, "32-channel blocks, 2x4x32 tiles")):
            lx = ops.Conv3dLayerX3(w, algo=bit)
            yx = lx(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out_exp=4)
            print(f"   {nm} vs default form: {(ops.from_split(yx, 4) - y_rt).abs().max().item() / y_rt.abs().max().item():.2e}")
            extra.append((nm, (lambda l_: (lambda: l_(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out=ys, out_exp=4, overflow=flag)))(lx)))
    for label, fn in tuple(extra) + (("serial: split -> split", lambda: x3s(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out=ys, out_exp=4, overflow=flag)),
                      ("fp32 Winograd F(4,3)", lambda: ref_layer(x, scale, bias, None, ops.EPI_RELU, y_ref)),
                      ("split -> f32 NCDHW", lambda: x3(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out_f32=y_x3)),
                      ("split -> split C8", lambda: x3(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out=ys, out_exp=4)),
                      ("split -> split C8 + flag", lambda: x3(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out=ys, out_exp=4, overflow=flag)),
                      ("split -> split + head", (lambda: x3(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out=ys, out_exp=4, head=head, overflow=flag))
                       if cout == 32 else None),
                      ("to_split (layout pass)", lambda: ops.to_split(x, x_exp, xs))):
        if fn is None:
            continue
        ms, _ = bench.timed_ms(fn, args.reps, 3)
        print(f"   {label:24s} {ms:7.3f} ms   {flop / ms / 1e9:7.1f} TFLOP/s algorithmic", flush=True)
