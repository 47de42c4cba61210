import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snvc_amd import ops
dev = torch.device("cuda:0")
shapes = {"full": (32, 32, (192, 96, 312)), "half": (64, 64, (96, 48, 156))}
ci, co, shp = shapes[sys.argv[1] if len(sys.argv) > 1 else "full"]
x = torch.relu(torch.randn(1, ci, *shp, device=dev)); g = torch.randn(1, co, *shp, device=dev) * 1e-4
for _ in range(6):
    ops.conv3d_wgrad(x, g, 3, 1, 1, 1)
torch.cuda.synchronize()
