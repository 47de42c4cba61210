#!/usr/bin/env python3
"""Is a layer bound by its schedule or by the chip's power budget?  The same launch -- same instruction stream, same addresses,
same bytes moved -- on random operands and on all-zero operands (no switching in the matrix pipe: the clock stays up).
   python tools/power_probe.py          (results: profiles/r4/kernel_experiments_r4.txt item 10)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from snvc_amd.models import submodule as S
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
for name, cin, cout, k, shape in (("cfg5 k7 128->64", 128, 64, 7, (1, 16, 80, 160, 160, 8)), ("released k7 64->32", 64, 32, 7, (2, 8, 32, 128, 192, 8)),
                                  ("cfg5 k5 64->64", 64, 64, 5, (1, 8, 80, 160, 160, 8)), ("cfg5 k3 64->64", 64, 64, 3, (1, 8, 80, 160, 160, 8))):
    for kind in ("random", "zeros"):
        m = S.convbn_3d(cin, cout, k, 1, (k - 1) // 2).to(dev).eval()
        if kind == "zeros":
            with torch.no_grad():
                m[0].weight.zero_()
        xh = (torch.randn(shape, device=dev) if kind == "random" else torch.zeros(shape, device=dev)).half()
        fn = lambda: m.fused_f16(xh, relu=True)
        ms, _ = bench.timed_ms(fn, 10, 3)
        flop = 2.0 * shape[0] * shape[2] * shape[3] * shape[4] * cin * cout * k ** 3
        print(f"{name:22s} {kind:7s} {ms:7.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s", flush=True)

# split mode (f16x3): conv2 of the cfg2 step
from snvc_amd import ops  # noqa: E402
cin, shp = 32, (bench.D, bench.H, bench.W)
for kind in ("random", "zeros"):
    xin = torch.relu(torch.randn(1, cin, *shp, device=dev)) if kind == "random" else torch.zeros(1, cin, *shp, device=dev)
    wt = (torch.randn(cin, cin, 3, 3, 3, device=dev) * 0.05) if kind == "random" else torch.zeros(cin, cin, 3, 3, 3, device=dev)
    lay = ops.Conv3dLayerX3(wt)
    xs_ = ops.to_split(xin, 4)
    del xin
    ys_ = torch.empty_like(xs_)
    flag_ = torch.zeros(1, dtype=torch.int32, device=dev)
    ms, _ = bench.timed_ms(lambda: lay(xs_, 4, flags=ops.EPI_RELU, out=ys_, out_exp=4, overflow=flag_), 50, 3)
    print(f"{'split conv2 32->32':22s} {kind:7s} {ms:7.3f} ms  {3 * 2.0 * 27 * cin * cin * shp[0] * shp[1] * shp[2] / ms / 1e9:7.1f} TFLOP/s executed", flush=True)
