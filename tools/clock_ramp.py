#!/usr/bin/env python3
"""How long the GPU takes to reach its sustained clocks after an idle stretch: ms/step of the cfg2 step in consecutive
20-step windows, from a cold start and again after a 3-second sleep.  (Result on MI355X: the first window after idle reads
2.54-2.56 ms/step, every later one 2.42-2.45; bench.py therefore runs 0.15 s of the step, untimed, before each leg's W warm-up
steps and says so in `config.prewarm`.)        python tools/clock_ramp.py"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from snvc_amd.models.stereo_volume import GlobalStack  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
m = GlobalStack(bench.C)
m.load_state_dict(bench.seeded_state(m))
m.eval().to(dev)
left, right, shift = bench.make_inputs(0, dev)


def timed(k=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        m.forward_pair(left, right, shift, 1)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k


for _ in range(5):
    m.forward_pair(left, right, shift, 1)
gc.collect()
gc.disable()
tt = time.perf_counter()
for i in range(24):
    print(f"t = {time.perf_counter() - tt:5.2f} s   {timed(20):.3f} ms/step", flush=True)
    if i == 11:
        time.sleep(3.0)
        print("-- slept 3 s --")
