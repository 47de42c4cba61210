#!/usr/bin/env python3
"""Times the cost-volume backward kernels at cfg2 size (full volume and right half only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from snvc_amd import ops
dev = torch.device("cuda:0")
left, right, shift = bench.make_inputs(0, dev)
C, H, W, D = left.shape[1], left.shape[2], left.shape[3], shift.shape[1]
g = torch.randn(1, 2 * C, D, H, W, device=dev)
ms, _ = bench.timed_ms(lambda: ops.cost_volume_backward(g, shift, 1), 10)
gb = g.numel() * 4 / 1e9
print(f"cost_volume_backward full : {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s read ({gb / ms / 8 * 100 * 1e3 / 1e3:.1f} % of 8 TB/s)")
gr = g[:, C:].contiguous()
ms, _ = bench.timed_ms(lambda: ops.cost_volume_backward_right(gr, shift), 10)
gb = gr.numel() * 4 / 1e9
print(f"cost_volume_backward right: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s read ({gb / ms / 8 * 100:.1f} % of 8 TB/s)")
