import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snvc_amd import _lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
torch.set_grad_enabled(False)
for shape in ((8, 8, 64), (8, 8, 70), (10, 8, 64), (8, 9, 64), (4, 4, 32), (8, 4, 32), (4, 8, 32), (4, 4, 64), (12, 12, 96), (16, 8, 32)):
    cin = cout = 32
    x = torch.relu(torch.randn(1, cin, *shape, device=dev)) * 1.5
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * np.sqrt(2.0 / (cin * 27))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    xs = ops.to_split(x, 4)
    ref = ops.Conv3dLayerX3(w, algo=0)
    q16 = ops.Conv3dLayerX3(w, algo=_lib.ALGO_X3_Q16)
    fa = ops.from_split(ref(xs, 4, flags=ops.EPI_RELU, out_exp=4, overflow=flag), 4)
    fb = ops.from_split(q16(xs, 4, flags=ops.EPI_RELU, out_exp=4, overflow=flag), 4)
    d = (fa - fb).abs()
    bad = (d > 1e-4 * fa.abs().max()).nonzero()
    print(shape, "max diff/range", float(d.max() / fa.abs().max()), "bad", len(bad), bad[:3].tolist() if len(bad) else "")
