"""r6: the split-operand weight gradient (conv3d_wgrad_x3_kernel) against float64 and against the fp32 forms; timing at cfg2 sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from snvc_amd import _lib, ops
import bench
dev = torch.device("cuda:0")

def ref64(x, g, k=3):
    w = torch.zeros(g.shape[1], x.shape[1], k, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv3d(x.double(), w, padding=k // 2)
    (y * g.double()).sum().backward()
    return w.grad

cases = {"many tiles": (1, 32, 32, (10, 40, 96)), "ragged": (2, 40, 24, (5, 10, 44)), "one tile": (1, 32, 64, (2, 4, 32)),
         "64x64": (1, 64, 64, (9, 12, 64)), "deep": (1, 32, 32, (40, 8, 32)), "tiny grads": (1, 32, 32, (6, 8, 64)), "w78": (1, 64, 64, (6, 10, 78)), "w38 ragged": (2, 40, 24, (5, 7, 38))}
for name, (N, ci, co, shp) in cases.items():
    r = np.random.default_rng(5)
    x = torch.from_numpy(r.standard_normal((N, ci) + shp).astype(np.float32))
    g = torch.from_numpy(r.standard_normal((N, co) + shp).astype(np.float32))
    if name == "tiny grads":
        g = g * 1e-9
        x = x * 3e4
    exp = ref64(x, g)
    with ops.conv_variant(0):
        a = ops.conv3d_wgrad(x.to(dev), g.to(dev), 3, 1, 1, 1)
        a2 = ops.conv3d_wgrad(x.to(dev), g.to(dev), 3, 1, 1, 1)
    with ops.conv_variant(_lib.ALGO_WGRAD_FP32):
        b = ops.conv3d_wgrad(x.to(dev), g.to(dev), 3, 1, 1, 1)
    ea = ((a.cpu().double() - exp).abs().max() / exp.abs().max()).item()
    eb = ((b.cpu().double() - exp).abs().max() / exp.abs().max()).item()
    print(f"{name:12s} x3 err {ea:.2e}  fp32 err {eb:.2e}  deterministic {torch.equal(a, a2)}  differs {not torch.equal(a, b)}", flush=True)

if "time" in sys.argv:
    for (ci, co, shp) in ((32, 32, (192, 96, 312)), (64, 64, (96, 48, 156)), (64, 64, (48, 24, 78))):
        x = torch.relu(torch.randn(1, ci, *shp, device=dev)); g = torch.randn(1, co, *shp, device=dev) * 1e-4
        for tag, bits in (("x3", 0), ("fp32", _lib.ALGO_WGRAD_FP32)):
            with ops.conv_variant(bits):
                ms, dw = bench.timed_ms(lambda: ops.conv3d_wgrad(x, g, 3, 1, 1, 1), 10, 3)
            print(ci, co, shp, tag, round(ms, 3), "ms", flush=True)
        with ops.conv_variant(0):
            a = ops.conv3d_wgrad(x, g, 3, 1, 1, 1)
        with ops.conv_variant(_lib.ALGO_WGRAD_FP32):
            b = ops.conv3d_wgrad(x, g, 3, 1, 1, 1)
        print("   x3 vs fp32 rel diff", ((a - b).abs().max() / b.abs().max()).item())

# ---- stride 2
def ref64_s2(x, g):
    w = torch.zeros(g.shape[1], x.shape[1], 3, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv3d(x.double(), w, stride=2, padding=1)
    assert y.shape == g.shape, (y.shape, g.shape)
    (y * g.double()).sum().backward()
    return w.grad
cases2 = {"s2": (2, 32, 64, (6, 12, 40)), "s2 many": (1, 32, 32, (12, 40, 160)), "s2 64x64": (1, 64, 64, (8, 8, 72)), "s2 deep": (1, 32, 64, (40, 8, 64)), "s2 w76": (1, 32, 64, (4, 12, 76)), "s2 w156": (1, 64, 64, (8, 6, 156))}
for name, (N, ci, co, shp) in cases2.items():
    r = np.random.default_rng(6)
    x = torch.from_numpy(r.standard_normal((N, ci) + shp).astype(np.float32))
    g = torch.from_numpy(r.standard_normal((N, co) + tuple(s // 2 for s in shp)).astype(np.float32))
    exp = ref64_s2(x, g)
    with ops.conv_variant(0):
        a = ops.conv3d_wgrad(x.to(dev), g.to(dev), 3, 2, 1, 1)
        a2 = ops.conv3d_wgrad(x.to(dev), g.to(dev), 3, 2, 1, 1)
    with ops.conv_variant(_lib.ALGO_WGRAD_FP32):
        b = ops.conv3d_wgrad(x.to(dev), g.to(dev), 3, 2, 1, 1)
    ea = ((a.cpu().double() - exp).abs().max() / exp.abs().max()).item()
    eb = ((b.cpu().double() - exp).abs().max() / exp.abs().max()).item()
    print(f"{name:12s} x3 err {ea:.2e}  fp32 err {eb:.2e}  deterministic {torch.equal(a, a2)}  differs {not torch.equal(a, b)}", flush=True)
if "time" in sys.argv:
    for (ci, co, shp) in ((32, 64, (192, 96, 312)), (64, 64, (96, 48, 156))):
        x = torch.relu(torch.randn(1, ci, *shp, device=dev)); g = torch.randn(1, co, *(s // 2 for s in shp), device=dev) * 1e-4
        for tag, bits in (("x3", 0), ("fp32", _lib.ALGO_WGRAD_FP32)):
            with ops.conv_variant(bits):
                ms, dw = bench.timed_ms(lambda: ops.conv3d_wgrad(x, g, 3, 2, 1, 1), 10, 3)
            print("s2", ci, co, shp, tag, round(ms, 3), "ms", flush=True)
