import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from snvc_amd import ops
dev = torch.device("cuda:0")
left, right, shift = bench.make_inputs(0, dev)
ld, rd, sd = left.double(), right.double(), shift.double()
ms, vol = bench.timed_ms(lambda: ops.cost_volume_forward(ld, rd, sd, 1), 5, 2)
print("fp64 fwd", round(ms, 3), "ms", round(2 * bench.CV_BYTES / ms / 1e6, 1), "GB/s")
from oracle import native as O
ref = O.cost_volume_forward(ld[:, :4].cpu().numpy(), rd[:, :4].cpu().numpy(), sd.cpu().numpy(), 1)
got = ops.cost_volume_forward(ld[:, :4].contiguous(), rd[:, :4].contiguous(), sd, 1).cpu().numpy()
print("fp64 bit-exact vs oracle:", np.array_equal(got, ref))
g = torch.randn(1, 64, 192, 96, 312, device=dev)
ms, _ = bench.timed_ms(lambda: ops.cost_volume_backward(g, shift, 1), 10, 3)
print("fp32 bwd", round(ms, 3), "ms", round(bench.CV_BYTES / ms / 1e6, 1), "GB/s")
