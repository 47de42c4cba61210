#!/usr/bin/env python3
"""Do the MFMA-bound conv2 and the HBM-bound hourglass conv1 (stride 2) / expand pass overlap when launched on two streams?
Independent buffers, no dependency: sequential on one stream against concurrent on two (r5 feasibility probe for slab pipelining).
   python tools/time_overlap.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from snvc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
full = (192, 96, 312)


def rnd(c, shape, e=3):
    return ops.to_split(torch.relu(torch.randn(1, c, *shape, device=dev)) * 1.5, e)


w2 = torch.randn(32, 32, 3, 3, 3, device=dev) * np.sqrt(2.0 / (32 * 27))
w1 = torch.randn(64, 32, 3, 3, 3, device=dev) * np.sqrt(2.0 / (32 * 27))
conv2, hg1 = ops.Conv3dLayerX3(w2), ops.Conv3dLayerX3(w1, 3, 2, 1, 1)
xa, xb = rnd(32, full), rnd(32, full)
ya = torch.empty_like(xa)
yb = torch.empty((1, 2, 8, 96, 48, 156, 8), dtype=torch.float16, device=dev)
sc32, bi32 = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.2
sc64, bi64 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.2
head = torch.randn(32, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
f_conv2 = lambda: conv2(xa, 3, sc32, bi32, flags=ops.EPI_RELU, out=ya, out_exp=2, head=head, overflow=flag)      # noqa: E731
f_hg1 = lambda: hg1(xb, 3, sc64, bi64, flags=ops.EPI_RELU, out=yb, out_exp=2, overflow=flag)                    # noqa: E731
big = torch.empty(736 * (1 << 20) // 4, dtype=torch.float32, device=dev)
f_fill = lambda: big.fill_(1.0)      # noqa: E731  a 0.74 GB write stream (stands in for the expand pass)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def concurrent(f, g):
    def run():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            f()
        with torch.cuda.stream(s2):
            g()
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    return run


t2, t1, tf = timed(f_conv2), timed(f_hg1), timed(f_fill)
print(f"alone: conv2 {t2:.1f} us, hg conv1 {t1:.1f} us, 0.74 GB fill {tf:.1f} us")
print(f"conv2 then hg conv1 on one stream: {timed(lambda: (f_conv2(), f_hg1())):.1f} us;  on two streams: {timed(concurrent(f_conv2, f_hg1)):.1f} us")
print(f"conv2 then fill on one stream:     {timed(lambda: (f_conv2(), f_fill())):.1f} us;  on two streams: {timed(concurrent(f_conv2, f_fill)):.1f} us")
print(f"hg conv1 then fill on one stream:  {timed(lambda: (f_hg1(), f_fill())):.1f} us;  on two streams: {timed(concurrent(f_hg1, f_fill)):.1f} us")
s3 = torch.cuda.Stream()


def three():
    cur = torch.cuda.current_stream()
    for s in (s1, s2, s3):
        s.wait_stream(cur)
    with torch.cuda.stream(s1):
        f_conv2()
    with torch.cuda.stream(s2):
        f_hg1()
    with torch.cuda.stream(s3):
        f_fill()
    for s in (s1, s2, s3):
        cur.wait_stream(s)


print(f"all three on one stream: {timed(lambda: (f_conv2(), f_hg1(), f_fill())):.1f} us;  on three streams: {timed(three):.1f} us")
