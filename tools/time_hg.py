#!/usr/bin/env python3
"""The split-mode hourglass layers of cfg2 one by one, back to back, in their selectable kernel forms (r5).
   python tools/time_hg.py [--reps 20]
Also: a read-after-write probe of the Infinity Cache (does a tensor a kernel has just WRITTEN come back from the 256 MB cache?)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from snvc_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--no-probe", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)


def rnd(c, shape, e=3):
    return ops.to_split(torch.relu(torch.randn(1, c, *shape, device=dev)) * 1.5, e)


def layer(cin, cout, stride=1, transposed=False, algo=None):
    shape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = torch.randn(*shape, device=dev) * np.sqrt(2.0 / (cin * 27))
    return ops.Conv3dLayerX3(w, 3, stride, 1, 1, transposed, algo=algo)


full, half, quarter = (192, 96, 312), (96, 48, 156), (48, 24, 78)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
sc64, bi64 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.2
cases = [
    ("conv2  s1 32->32  192x96x312", 32, 32, 1, False, full, [("q16", _lib.ALGO_X3_Q16)]),
    ("hg conv1  s2 32->64  192x96x312 -> 96x48x156", 32, 64, 2, False, full, [("2x4x32", 0), ("q16 3 slots", _lib.ALGO_X3_Q16)]),
    ("hg conv3  s2 64->64  96x48x156 -> 48x24x78", 64, 64, 2, False, half, [("2x4x32", 0), ("q16 3 slots", _lib.ALGO_X3_Q16)]),
    ("hg conv2  s1 64->64  96x48x156", 64, 64, 1, False, half, [("auto", None)]),
    ("hg conv4  s1 64->64  48x24x78", 64, 64, 1, False, quarter, [("auto", None), ("q16", _lib.ALGO_X3_Q16), ("serial 64", _lib.ALGO_X3_SERIAL),
                                                               ("narrow", _lib.ALGO_X3_NARROW)]),
    ("hg conv5  transposed 64->64  48x24x78 -> 96x48x156 (+pre, ReLU, fp32 out)", 64, 64, 2, True, quarter,
     [("4x4x32", 0), ("2x4x32 (SMALL)", _lib.ALGO_X3_SMALL)]),
]
for name, cin, cout, stride, tr, shape, forms in cases:
    x = rnd(cin, shape)
    print(name)
    for label, algo in forms:
        lay = layer(cin, cout, stride, tr, algo)
        out_sp = lay.out_spatial(shape)
        if tr:
            res = rnd(cout, out_sp, 2)
            tail = ops.TailWeightsX3(torch.randn(cout, 27, device=dev) * 0.1)
            fn32 = lambda: lay(x, 3, sc64, bi64, residual=res, flags=ops.EPI_RELU | ops.EPI_ADD_PRE, out_exp=2, to_f32=True)  # noqa: E731
            fnt = lambda: lay.forward_tail(x, 3, sc64, bi64, tail, residual=res, flags=ops.EPI_RELU | ops.EPI_ADD_PRE, out_exp=2, overflow=flag)  # noqa: E731
            ms32, _ = bench.timed_ms(fn32, args.reps, 3)
            mst, t = bench.timed_ms(fnt, args.reps, 3)
            msg, _ = bench.timed_ms(lambda: ops.deconv_tail_gather(t), args.reps, 3)
            print(f"   {label:18s} fp32 NCDHW out {ms32 * 1e3:7.1f} us    tail projection {mst * 1e3:7.1f} us   (+ gather {msg * 1e3:5.1f} us)", flush=True)
            del res
        else:
            y = torch.empty((1, 2, cout // 8) + out_sp + (8,), dtype=torch.float16, device=dev)
            ms, _ = bench.timed_ms(lambda: lay(x, 3, sc64[:cout], bi64[:cout], flags=ops.EPI_RELU, out=y, out_exp=2, overflow=flag), args.reps, 3)
            flop = 2.0 * np.prod(out_sp) * cin * cout * 27
            print(f"   {label:18s} {ms * 1e3:7.1f} us   {flop / ms / 1e9:7.1f} TFLOP/s algorithmic", flush=True)
            del y
    del x
    torch.cuda.empty_cache()

if not args.no_probe:
    # Infinity Cache, read after write: a tensor of S MB is written by one kernel (fill) and read by the next (sum); the read's rate
    # tells whether the just-written lines are served by the 256 MB cache (S well below it) or by HBM (S far above it)
    print("read-after-write probe (fill_ then sum), GB/s of the read:")
    for mb in (64, 128, 192, 256, 384, 768, 1536):
        t = torch.empty(mb * (1 << 20) // 4, dtype=torch.float32, device=dev)
        best = 1e9
        for _ in range(5):
            t.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            s_ = t.sum()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        t.fill_(1.0)
        msr, _ = bench.timed_ms(lambda: t.sum(), 10, 2)      # read after read (back to back)
        print(f"   {mb:5d} MB   after its own write: {mb * 1.048576 / best:8.1f} GB/s     re-read back to back: {mb * 1.048576 / msr:8.1f} GB/s", flush=True)
        del t
