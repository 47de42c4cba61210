import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snvc_amd import ops
dev = torch.device("cuda:0")
shapes = {"full": (32, 32, (192, 96, 312), 1), "half": (64, 64, (96, 48, 156), 1), "s2": (32, 64, (192, 96, 312), 2)}
ci, co, shp, st = shapes[sys.argv[1] if len(sys.argv) > 1 else "full"]
x = torch.relu(torch.randn(1, ci, *shp, device=dev)); g = torch.randn(1, co, *(s // st for s in shp), device=dev) * 1e-4
ax, ag = ops.amax_word(dev), ops.amax_word(dev)        # the maxima as the training step supplies them
ax[0:1] = x.abs().max().reshape(1).view(torch.int32); ag[0:1] = g.abs().max().reshape(1).view(torch.int32)
for _ in range(6):
    ops.conv3d_wgrad(x, g, 3, st, 1, 1, amax_x=ax, amax_g=ag)
torch.cuda.synchronize()
