#!/usr/bin/env python3
"""A/B timing of the cfg2 step under switches, INTERLEAVED (A B C A B C ...) in one process: the power-limited kernels drift by 2-4 % with
the part's temperature, so only alternating legs on the same box separate effects of that size.   python tools/ab_step.py [--rounds 4]"""
import argparse
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from snvc_amd import ops  # noqa: E402
from snvc_amd.models.stereo_volume import GlobalStack  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--only", default="", help="comma-separated substrings: only legs matching one (default is always run)")
args = ap.parse_args()
dev = torch.device("cuda:0")
model = GlobalStack(bench.C)
model.load_state_dict(bench.seeded_state(model))
model.eval().to(dev)
left, right, shift = bench.make_inputs(0, dev)


def setup(prep_streams=False, check="call", s2q=True, fused_tail=True, q16_min=256, split_prep=True, stream_out=False):
    model.prep_streams = prep_streams
    model.overflow_check = check
    model.fused_tail = fused_tail
    ops.X3_Q16_S2[0] = s2q
    ops.X3_Q16_MIN_JOBS[0] = q16_min
    model.split_prep = split_prep
    model.stream_out = stream_out


LEGS = {
    "default": {},
    "prep chains on side streams": {"prep_streams": True},
    "overflow check deferred": {"check": "deferred"},
    "stride-2 layers: 32x32x16 serial-plane form": {"s2q": False},
    "two-launch tail (r4)": {"fused_tail": False},
    "hg conv4 on the 2x4x32 32x32x16 form (r4 rule: 1024 jobs)": {"q16_min": 1024},
    "sheared prep (G, G') on the fp32 matrix pipe (r4)": {"split_prep": False},
    "first layer written with non-temporal stores": {"stream_out": True},
}
if args.only:
    LEGS = {k: v for k, v in LEGS.items() if k == "default" or any(t in k for t in args.only.split(","))}
res = {k: [] for k in LEGS}
with torch.no_grad():
    for _ in range(30):
        model.forward_pair(left, right, shift, 1)
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    for r in range(args.rounds):
        for name, kw in LEGS.items():
            setup(**kw)
            for _ in range(8):
                model.forward_pair(left, right, shift, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model.forward_pair(left, right, shift, 1)
            torch.cuda.synchronize()
            res[name].append(1e3 * (time.perf_counter() - t0) / args.steps)
            model.check_overflow()
    gc.enable()
base = float(np.median(res["default"]))
for name, v in res.items():
    m = float(np.median(v))
    print(f"{name:52s} median {m:.4f} ms/step  ({', '.join(f'{x:.3f}' for x in v)})   {m - base:+.4f} vs default")
