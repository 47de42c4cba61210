"""r6 scratch diagnostics: (1) the cfg3 gather's time standalone / after a training leg; (2) spin bound of ops.spin_wait vs the step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from snvc_amd import ops

dev = torch.device("cuda:0")
def gather_ms(tag):
    m = bench.local_model((96, 96, 96), 32, dev)
    import numpy as np
    r = np.random.default_rng(5)
    v = 96 ** 3
    lf = torch.from_numpy(r.standard_normal((8, 32, 64, 64)).astype(np.float32)).to(dev)
    rf = torch.from_numpy(r.standard_normal((8, 32, 64, 64)).astype(np.float32)).to(dev)
    gl = torch.from_numpy(r.uniform(-8, 264, (8, 2, v)).astype(np.float32)).to(dev)
    gr = torch.from_numpy(r.uniform(-8, 264, (8, 2, v)).astype(np.float32)).to(dev)
    with torch.no_grad():
        ms, _ = bench.timed_ms(lambda: m.construct_voxel(lf, rf, gl, gr), 20)
        out = torch.empty(8, 64, 96, 96, 96, device=dev)
        ms2, _ = bench.timed_ms(lambda: ops.voxel_gather_forward(lf, rf, gl, gr, (256, 256)), 20)
    print(tag, "construct_voxel", round(ms, 3), "ops.voxel_gather_forward", round(ms2, 3), "reserved GB", torch.cuda.memory_reserved() / 1e9, flush=True)
    del m, out

gather_ms("fresh")
def barrier():
    torch.cuda.synchronize()
if "train" in sys.argv:
    tr = bench.run_train(0, 1, dev, None, 5, 2, barrier)
    print("train ms", tr["ms_per_step"], "reserved GB", torch.cuda.memory_reserved() / 1e9)
    gather_ms("after train")
    torch.cuda.empty_cache()
    gather_ms("after train + empty_cache")

from snvc_amd.models.stereo_volume import GlobalStack
from snvc_amd.extension.build_cost_volume import build_cost_volume
model = GlobalStack(32); model.load_state_dict(bench.seeded_state(model)); model.eval().to(dev)
left, right, shift = bench.make_inputs(0, dev)
orig = ops.spin_wait
for us in (1000.0, 5000.0, 20000.0, 100.0):
    def sw(ev, spin_us=us):
        return orig(ev, spin_us)
    ops.spin_wait = sw
    with torch.no_grad():
        for _ in range(300):
            model.forward_pair(left, right, shift, 1)
        torch.cuda.synchronize()
        res = []
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(200):
                model.forward_pair(left, right, shift, 1)
            torch.cuda.synchronize()
            res.append(round((time.perf_counter() - t0) / 200 * 1e3, 4))
        t0 = time.perf_counter()
        for _ in range(200):
            model(build_cost_volume(left, right, shift, 1))
        torch.cuda.synchronize()
        api = round((time.perf_counter() - t0) / 200 * 1e3, 4)
    print("spin_us", us, "forward_pair ms", res, "reference_api ms", api, flush=True)
