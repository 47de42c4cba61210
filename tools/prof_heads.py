#!/usr/bin/env python3
"""2D BEV neck + heads (VernierScale.heads_2d) on the released shape, a few calls: for a rocprofv3 --kernel-trace pass.
    rocprofv3 --kernel-trace --stats -d out -o heads -- python3 tools/prof_heads.py --crops 2"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--crops", type=int, default=2)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
grid, F = (32, 128, 192), 32
m = bench.local_model(grid, F, dev)
bev = torch.randn(args.crops, F * grid[0] // 4, grid[1], grid[2], device=dev)
with torch.no_grad():
    for _ in range(3):
        m.heads_2d(bev)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        m.heads_2d(bev)
    b.record()
    torch.cuda.synchronize()
print(f"heads_2d: {a.elapsed_time(b) / args.reps / args.crops:.3f} ms/crop at {args.crops} crops per call")
