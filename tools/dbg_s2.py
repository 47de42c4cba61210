import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from snvc_amd.models import submodule as S
dev = torch.device("cuda:0")
r = np.random.default_rng(0)
for (cin, cout, shape, n) in [(32, 64, (8, 8, 40), 1), (32, 64, (8, 8, 40), 2), (64, 64, (16, 16, 48), 1), (64, 32, (6, 10, 72), 2), (7, 32, (4, 4, 8), 1)]:
    x = torch.from_numpy(r.standard_normal((n, cin) + shape).astype(np.float32))
    conv = S.HipConv3d(cin, cout, 3, 2, 1, bias=False)
    w = torch.from_numpy((r.standard_normal(tuple(conv.weight.shape)) * 0.05).astype(np.float32))
    conv.weight.data.copy_(w)
    conv = conv.to(dev)
    ref = F.conv3d(x, w, None, 2, 1)
    with torch.no_grad():
        y = conv(x.to(dev)).cpu()
    e = (y - ref).abs().max().item() / ref.abs().max().item()
    print(cin, cout, shape, n, "->", tuple(ref.shape), "rel err %.2e" % e)
    if e > 1e-3:
        d = (y - ref).abs() > 1e-3 * ref.abs().max()
        idx = d.nonzero()
        print("  bad", len(idx), "of", d.numel(), "first", idx[:4].tolist(), "last", idx[-2:].tolist())

print("---- backward: s2 conv and deconv layers")
for kind in ("s2", "deconv"):
    for (cin, cout, shape, n) in [(32, 64, (8, 8, 40), 2), (64, 32, (4, 4, 20), 2)]:
        x = torch.from_numpy(r.standard_normal((n, cin) + shape).astype(np.float32))
        if kind == "s2":
            m = S.convbn_3d(cin, cout, 3, 2, 1)
            ref_fn = lambda xx, ww: F.conv3d(xx, ww, None, 2, 1)
        else:
            m = S._deconvbn_3d(cin, cout, False)
            ref_fn = lambda xx, ww: F.conv_transpose3d(xx, ww, None, 2, 1, 1)
        conv = m[0]
        w = torch.from_numpy((r.standard_normal(tuple(conv.weight.shape)) * 0.05).astype(np.float32))
        conv.weight.data.copy_(w)
        xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
        yr = ref_fn(xr, wr)
        gy = torch.from_numpy(r.standard_normal(tuple(yr.shape)).astype(np.float32))
        yr.backward(gy)
        conv = conv.to(dev)
        xd = x.to(dev).requires_grad_(True)
        y = S.fused_conv3d(conv, None, xd)
        y.backward(gy.to(dev))
        ey = (y.detach().cpu() - yr.detach()).abs().max().item() / yr.abs().max().item()
        ex = (xd.grad.cpu() - xr.grad).abs().max().item() / xr.grad.abs().max().item()
        ew = (conv.weight.grad.cpu() - wr.grad).abs().max().item() / wr.grad.abs().max().item()
        print(kind, cin, cout, shape, "y %.1e dx %.1e dw %.1e" % (ey, ex, ew))
