import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from snvc_amd.models import submodule as S
dev = torch.device("cuda:0")
r = np.random.default_rng(0)
for (cin, cout, k, shape, n) in [(64, 32, 7, (16, 16, 24), 1), (32, 64, 7, (16, 16, 24), 2), (32, 32, 5, (16, 16, 24), 2),
                                  (32, 32, 5, (3, 5, 8), 1), (6, 32, 7, (4, 4, 36), 1), (32, 32, 7, (9, 6, 44), 1)]:
    x = torch.from_numpy(r.standard_normal((n, cin) + shape).astype(np.float32))
    conv = S.HipConv3d(cin, cout, k, 1, k // 2, bias=False)
    w = torch.from_numpy((r.standard_normal(tuple(conv.weight.shape)) * 0.05).astype(np.float32))
    conv.weight.data.copy_(w)
    conv = conv.to(dev)
    ref = F.conv3d(x, w, None, 1, k // 2)
    with torch.no_grad():
        y = conv(x.to(dev)).cpu()
    e = (y - ref).abs().max().item() / ref.abs().max().item()
    print(cin, cout, k, shape, n, "rel err %.2e" % e)
    if e > 1e-3:
        d = (y - ref).abs()
        idx = (d > 1e-3 * ref.abs().max()).nonzero()
        print("  bad count", len(idx), "first", idx[:5].tolist(), "last", idx[-3:].tolist())

print("---- k5 dilation 2")
for (cin, cout, shape, n) in [(32, 32, (16, 16, 24), 2), (32, 64, (9, 9, 36), 1), (6, 32, (4, 5, 40), 1)]:
    x = torch.from_numpy(r.standard_normal((n, cin) + shape).astype(np.float32))
    conv = S.HipConv3d(cin, cout, 5, 1, 4, dilation=2, bias=False)
    w = torch.from_numpy((r.standard_normal(tuple(conv.weight.shape)) * 0.05).astype(np.float32))
    conv.weight.data.copy_(w)
    conv = conv.to(dev)
    ref = F.conv3d(x, w, None, 1, 4, 2)
    with torch.no_grad():
        y = conv(x.to(dev)).cpu()
    print(cin, cout, shape, n, "rel err %.2e" % ((y - ref).abs().max().item() / ref.abs().max().item()))
print("---- backward")
for (cin, cout, k, shape, n) in [(64, 32, 7, (16, 16, 24), 2), (32, 32, 5, (16, 16, 24), 2), (32, 32, 3, (16, 16, 24), 2)]:
    x = torch.from_numpy(r.standard_normal((n, cin) + shape).astype(np.float32))
    conv = S.HipConv3d(cin, cout, k, 1, k // 2, bias=False)
    w = torch.from_numpy((r.standard_normal(tuple(conv.weight.shape)) * 0.05).astype(np.float32))
    conv.weight.data.copy_(w)
    gy = torch.from_numpy(r.standard_normal((n, cout) + shape).astype(np.float32))
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    F.conv3d(xr, wr, None, 1, k // 2).backward(gy)
    conv = conv.to(dev)
    xd = x.to(dev).requires_grad_(True)
    y = conv(xd)
    y.backward(gy.to(dev))
    ex = (xd.grad.cpu() - xr.grad).abs().max().item() / xr.grad.abs().max().item()
    ew = (conv.weight.grad.cpu() - wr.grad).abs().max().item() / wr.grad.abs().max().item()
    print(cin, cout, k, shape, "dx rel err %.2e  dw rel err %.2e" % (ex, ew))
