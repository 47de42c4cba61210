import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from snvc_amd import _lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
torch.set_grad_enabled(False)
for name, cin, cout, shape in (("small 32->32", 32, 32, (10, 9, 70)), ("conv2 32->32", 32, 32, (192, 96, 312)), ("hg2 64->64", 64, 64, (96, 48, 156))):
    x = torch.relu(torch.randn(1, cin, *shape, device=dev)) * 1.5
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * np.sqrt(2.0 / (cin * 27))
    scale, bias = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    head = torch.randn(cout, device=dev) if cout == 32 else None
    xs = ops.to_split(x, 4)
    ref = ops.Conv3dLayerX3(w, algo=0 if cout == 32 else _lib.ALGO_X3_NARROW)
    q16 = ops.Conv3dLayerX3(w, algo=_lib.ALGO_X3_Q16)
    a = ref(xs, 4, scale, bias, flags=ops.EPI_RELU, out_exp=4, head=head, overflow=flag)
    b = q16(xs, 4, scale, bias, flags=ops.EPI_RELU, out_exp=4, head=head, overflow=flag)
    if head is not None:
        (ya, ha), (yb, hb) = a, b
        print(name, "head max diff / range", float((ha - hb).abs().max() / ha.abs().max()))
    else:
        ya, yb = a, b
    fa, fb = ops.from_split(ya, 4), ops.from_split(yb, 4)
    print(name, "max|diff| / range:", float((fa - fb).abs().max() / fa.abs().max()), "flag", int(flag.item()))
    ys = torch.empty_like(ya)
    for label, lay in (("32x32x16", ref), ("16x16x32", q16)):
        ms, _ = bench.timed_ms(lambda: lay(xs, 4, scale, bias, flags=ops.EPI_RELU, out=ys, out_exp=4, head=head, overflow=flag), 20, 5)
        print(f"   {label}: {ms:.3f} ms", flush=True)
