import sys, os
sys.path.insert(0, os.getcwd())
import torch
from snvc_amd import ops
dev = torch.device("cuda:0")
x = torch.randn((1, 32, 192, 96, 312), device=dev); g = torch.randn((1, 32, 192, 96, 312), device=dev)
for _ in range(3): ops.conv3d_wgrad(x, g, 3, 1, 1, 1)
torch.cuda.synchronize()
