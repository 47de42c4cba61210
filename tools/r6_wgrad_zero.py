import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from snvc_amd import ops
dev = torch.device("cuda:0")
for kind in ("random", "zeros", "random"):
    if kind == "random":
        x = torch.relu(torch.randn(1, 32, 192, 96, 312, device=dev)); g = torch.randn(1, 32, 192, 96, 312, device=dev) * 1e-4
    else:
        x = torch.zeros(1, 32, 192, 96, 312, device=dev); g = torch.zeros(1, 32, 192, 96, 312, device=dev)
    ax, ag = ops.amax_word(dev), ops.amax_word(dev)
    ax[0:1] = torch.tensor([1.0], device=dev).view(torch.int32); ag[0:1] = torch.tensor([1e-4], device=dev).view(torch.int32)
    ms, _ = bench.timed_ms(lambda: ops.conv3d_wgrad(x, g, 3, 1, 1, 1, amax_x=ax, amax_g=ag), 20, 5)
    print(kind, round(ms, 3), "ms", flush=True)
    del x, g
