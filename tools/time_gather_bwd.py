import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from snvc_amd import ops
dev = torch.device("cuda:0")
grid, n, F = (32, 128, 192), 2, 32
v = grid[0]*grid[1]*grid[2]
r = np.random.default_rng(0)
g = torch.randn(n, 2*F, v, device=dev)
pl, pr = bench.projected_coordinates(n, grid, dev)
ul = torch.from_numpy(r.uniform(-8, 264, (n, 2, v)).astype(np.float32)).to(dev)
t = np.linspace(0, 1, v, dtype=np.float32)
cl = torch.from_numpy(np.stack([np.stack([100 + 8*t, 120 + 3*t])]*n)).to(dev)
for name, a, b in (("projected", pl, pr), ("uniform", ul, ul), ("coherent line", cl, cl)):
    for det in (True, False):
        ms, _ = bench.timed_ms(lambda: ops.voxel_gather_backward(g, a, b, (n, F, 64, 64), (256, 256), deterministic=det), 3)
        print(f"gather backward {name:14s} deterministic={det}: {ms:8.3f} ms / {n} crops", flush=True)
