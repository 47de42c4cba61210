"""The `cpu_baseline` legs of bench.py: the ONLY place outside tests/ and __graft_entry__.smoke() that runs anything under
``oracle/`` -- as the checker and the reported CPU rate, never as the thing measured or shipped."""
import os
import time
import types

import numpy as np
import torch

from .common import C, D, make_inputs, seeded_state

def cpu_baseline(d_sample=D, outputs=None):
    """CPU oracle on ONE FULL cfg2 pair (all 192 planes, measured, not scaled): C/OpenMP cost volume +
    torch-CPU 3D stack, every host core.  The inputs and weights are rank 0's (make_inputs(0), seeded_state), so the
    oracle's result (left in ``outputs["cost"]``) is what every timed leg of the GPU line has to reproduce."""
    from oracle import native as O
    from oracle import torch_ref as T
    # torch's own default thread count: on the GPU box (256 logical CPUs, a 16-CPU share per GPU) forcing os.cpu_count() threads made
    # the torch-CPU stack 2-4x SLOWER than the default (r5: the 96^3 trunk 15.9 s at 256 threads, 4.3 s at the default) -- the
    # baseline is the faster setting, and `cores` says how many threads that was
    cores = torch.get_num_threads()
    threads_before = cores
    left, right, shift = make_inputs(0, "cpu", d_sample)
    ref = T.GlobalStack(C)
    ref.load_state_dict(seeded_state(ref))
    ref.eval()
    ln, rn, sn = left.numpy(), right.numpy(), shift.numpy()
    t0 = time.perf_counter()
    vol = O.cost_volume_forward(ln, rn, sn, 1)
    t1 = time.perf_counter()
    with torch.no_grad():
        cost = ref(torch.from_numpy(vol))
    t2 = time.perf_counter()
    torch.set_num_threads(threads_before)
    scale = D / float(d_sample)
    if outputs is not None:
        outputs["cost"] = cost.numpy()
    return {
        "value": 1.0 / ((t2 - t0) * scale), "unit": "stereo-pairs/s", "cores": cores, "kind": "port",
        "sample": f"1 pair, {d_sample} of {D} disparity planes{'' if d_sample == D else ' (scaled)'}: {t2 - t0:.2f}s = "
                  f"cost volume (C oracle, OpenMP, {os.cpu_count()} logical CPUs visible) {t1 - t0:.2f}s + 3D stack (torch-CPU {torch.__version__}, "
                  f"{cores} threads = torch's default here) {t2 - t1:.2f}s",
    }


def local_inputs(grid, F, crops=1, seed=7):
    """Seeded host inputs of one local-model call: feature maps ~N(0,1) [crops,F,64,64] and grid projections
    uniform in [-8, 264) px (~6 % outside the 256 x 256 crop: zero padding)."""
    r = np.random.default_rng(seed)
    v = grid[0] * grid[1] * grid[2]
    return (r.standard_normal((crops, F, 64, 64)).astype(np.float32), r.standard_normal((crops, F, 64, 64)).astype(np.float32),
            r.uniform(-8, 264, (crops, 2, v)).astype(np.float32), r.uniform(-8, 264, (crops, 2, v)).astype(np.float32))


def local_oracle(grid, F, crops=1, seed=7, keep_layers=False, heads=False, gn=False):
    """The CPU oracle of the local (V-A) model's path on `crops` crops: numpy restatement of _sample_2d_feat (vernier.py:323-349) +
    the torch-CPU restatement of the BEV_type3 3D trunk (vernier.py:414-438) that tests/golden pins bit-equal to the imported
    reference, with bench.seeded_state's weights of the product model.  Returns a dict: the inputs, "voxel", "bev", "occupancy" (host
    tensors; with keep_layers every intermediate of trunk_3d), and the two timings."""
    from oracle import numpy_ref as NR
    from oracle import torch_ref as T
    from snvc_amd.models.vernier import VernierScale
    cores = torch.get_num_threads()                    # torch's own default: the caller's process setting is left alone (a test
    #                                                    process that is switched to os.cpu_count() threads on a 16-CPU share crawls)
    cfg = types.SimpleNamespace(vernier_type="BEV_type3", backbone="hrfeat", gn=gn, grid_resolution=[32, grid[1], 192],
                                resolution=(256, 256), x_range=(-1.0, 1.0), z_range=(-1.0, 1.0), num_parts=9)
    cfg.hrfeat = types.SimpleNamespace(output_channel=F, name="identity")
    cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
    sd = seeded_state(VernierScale(cfg))               # the product model's parameters (CPU construction: nothing runs)
    ref = T.VernierTrunk(F, grid, gn=gn, heads=heads)      # heads: the 2D BEV neck + heat-map / coordinate heads too (grids with nh in {16, 32})
    ref.load_state_dict({k: sd[k] for k in ref.state_dict()})
    ref.eval()
    lf, rf, gl, gr = local_inputs(grid, F, crops, seed)
    o = {"lf": lf, "rf": rf, "gl": gl, "gr": gr, "cores": cores}
    t0 = time.perf_counter()
    vox = NR.sample_2d_feat(lf, rf, gl, gr, (256, 256)).reshape((crops, 2 * F) + tuple(grid))
    t1 = time.perf_counter()
    with torch.no_grad():
        vt = torch.from_numpy(vox)
        if not keep_layers:
            bev, occ, _ = ref.trunk_3d(vt)
        else:       # trunk_3d (oracle/torch_ref.py, reference vernier.py:415-438) statement by statement, everything kept
            o["img"] = ref.vimg_feat(vt)
            o["v1"] = ref.conv1(vt)
            o["v2"] = ref.conv2(o["v1"]) + o["v1"]
            o["v3"] = ref.conv3(o["v2"]) + o["v2"]
            o["vh"] = (ref.hg_conv3d(o["v3"], None, None)[0] if ref.small else ref.hg_conv3d(o["v3"])) + o["v3"]
            o["t"] = ref.fg_cls_head[1](ref.fg_cls_head[0](o["vh"]))
            occ = ref.fg_cls_head[3](ref.fg_cls_head[2](o["t"]))
            o["cat"] = torch.cat([o["vh"], o["img"] * occ], dim=1)
            v4 = ref.pool_3d(ref.conv4(o["cat"]))
            bev = v4.reshape(crops, -1, v4.shape[3], v4.shape[4])
        if heads:                                    # vernier.py:440-450 (predict_3d_heatmaps' 2D half)
            o["heat"], o["coords"] = ref.heads_2d(bev)
    t2 = time.perf_counter()
    o.update(voxel=vt, bev=bev, occupancy=occ, ref=ref, gather_s=t1 - t0, trunk_s=t2 - t1)
    return o
