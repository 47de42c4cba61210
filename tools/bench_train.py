#!/usr/bin/env python3
"""cfg4 (BASELINE.json configs[3]) per-GPU measurement: one training step of the global stack on a
cfg2-sized pair -- build_cost_volume + GlobalStack forward with train-mode BatchNorm, loss =
mean(cost^2), backward through the HIP kernels (dgrad, wgrad, BN/ReLU backward, cost-volume
backward), then the flat-bucket gradient all-reduce (RCCL when launched with torchrun; a no-op on
one GPU).  Prints ms per step and a per-phase breakdown.  Not the driver's bench (bench.py is)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from snvc_amd import parallel as P  # noqa: E402
from snvc_amd.extension.build_cost_volume import build_cost_volume  # noqa: E402
from snvc_amd.models.stereo_volume import GlobalStack  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--disp", type=int, default=bench.D)
args = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    torch.distributed.init_process_group("nccl", device_id=dev)
model = GlobalStack(bench.C)
model.load_state_dict(bench.seeded_state(model))
model.train().to(dev)
left, right, shift = bench.make_inputs(local, dev, args.disp)
left.requires_grad_(); right.requires_grad_()
nparam = sum(p.numel() for p in model.parameters())


def step():
    for p in model.parameters():
        p.grad = None
    left.grad = right.grad = None
    t0 = time.perf_counter()
    out = model.forward_pair(left, right, shift, 1)
    loss = out.pow(2).mean()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    moved = P.all_reduce_gradients(model.parameters())
    torch.cuda.synchronize(); t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2, moved


for _ in range(args.warmup):
    step()
acc = [0.0, 0.0, 0.0]
for _ in range(args.steps):
    f, b, r, moved = step()
    acc[0] += f; acc[1] += b; acc[2] += r
f, b, r = (1e3 * a / args.steps for a in acc)
if int(os.environ.get("RANK", "0")) == 0:
    print(f"train step (D={args.disp}, 1 pair/GPU, {world} GPU): fwd {f:.2f} ms  bwd {b:.2f} ms  "
          f"all-reduce {r:.3f} ms ({moved / 1e6:.2f} MB, {nparam} params)  total {f + b + r:.2f} ms  "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
if world > 1:
    torch.distributed.destroy_process_group()
