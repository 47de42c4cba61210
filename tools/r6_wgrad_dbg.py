import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from snvc_amd import ops
dev = torch.device("cuda:0")
x = torch.relu(torch.randn(1, 32, 192, 96, 312, device=dev)); g = torch.randn(1, 32, 192, 96, 312, device=dev) * 1e-4
for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
    os.environ["SNVC_X3WG_DBG"] = str(dbg)
    ax, ag = ops.amax_word(dev), ops.amax_word(dev)
    ax[0:1] = x.abs().max().reshape(1).view(torch.int32); ag[0:1] = g.abs().max().reshape(1).view(torch.int32)
    ms, _ = bench.timed_ms(lambda: ops.conv3d_wgrad(x, g, 3, 1, 1, 1, amax_x=ax, amax_g=ag), 10, 3)
    print("dbg", dbg, "(no mfma)" if dbg & 1 else "", "(no store)" if dbg & 2 else "", "(no loads)" if dbg & 4 else "", round(ms, 3), "ms incl. reduce 0.04", flush=True)
