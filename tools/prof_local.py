#!/usr/bin/env python3
"""One local-model trunk leg under rocprofv3 --kernel-trace: python3 tools/prof_local.py <cfg5|released|cfg3> <f16|f32> [reps]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

name, precision = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
grid, F, crops = {"cfg5": ((80, 160, 160), 64, 1), "released": ((32, 128, 192), 32, 2), "cfg3": ((96, 96, 96), 32, 8)}[name]
dev = torch.device("cuda:0")
m = bench.local_model(grid, F, dev)
f16 = precision == "f16"
m.precision = "f16" if f16 else "auto"
r = np.random.default_rng(5)
lf = torch.from_numpy(r.standard_normal((crops, F, 64, 64)).astype(np.float32)).to(dev)
rf = torch.from_numpy(r.standard_normal((crops, F, 64, 64)).astype(np.float32)).to(dev)
pl, pr = bench.projected_coordinates(crops, grid, dev)
with torch.no_grad():
    def step():
        if f16:
            return m.trunk_3d_f16(m.construct_voxel_f16(lf, rf, pl, pr))
        vs = m.construct_voxel_x3(lf, rf, pl, pr)
        return m.trunk_3d(vs if vs is not None else m.construct_voxel(lf, rf, pl, pr))
    ms, _ = bench.timed_ms(step, reps, 3)
print(f"{name} {precision}: {ms / crops:.3f} ms/crop", flush=True)
