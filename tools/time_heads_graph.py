#!/usr/bin/env python3
"""2D neck + heads at the released shape: eager launches against one captured hipGraph (torch.cuda.CUDAGraph), 1 / 2 / 8 crops."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
m = bench.local_model((32, 128, 192), 32, dev)
with torch.no_grad():
    for crops in (1, 2, 8):
        bev = torch.randn(crops, 256, 128, 192, device=dev)
        ms_e, out_e = bench.timed_ms(lambda: m.heads_2d(bev), 20, 5)
        static = bev.clone()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                m.heads_2d(static)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out_g = m.heads_2d(static)

        def run():
            static.copy_(bev)
            g.replay()
            return out_g
        ms_g, _ = bench.timed_ms(run, 20, 5)
        same = all(torch.equal(a, b) for a, b in zip(out_e, out_g))
        print(f"{crops} crops: eager {ms_e:.3f} ms ({ms_e / crops:.3f} / crop)   graph {ms_g:.3f} ms ({ms_g / crops:.3f} / crop)   identical: {same}", flush=True)
