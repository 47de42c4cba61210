#!/usr/bin/env python3
"""Headline benchmark: stereo-pairs/sec, cost-volume build + 3D CNN forward (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (starts N ranks itself when N > 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N --mode train                    (cfg4: fwd + bwd + RCCL gradient all-reduce)
    python bench.py --gpus N --config cfg3                   (BASELINE configs[2]: 64 RoI crops 96^3 sharded over N ranks)

Workload (N=1): BASELINE.json configs[1], "Global scene model: 1242x375, 192 disparities, full 3D
hourglass fwd, batch=1 on 1 MI355X", synthesised as SURVEY.md section 8(d) cfg2:
left/right features [1,32,96,312] (1242x375 padded to 1248x384, stride 4), shift =
linspace(0, 95.5, 192), downsample 1 -> concat volume [1,64,192,96,312] -> GlobalStack(32)
(conv 64->32, conv 32->32, hourglass(32) + residual, 1x1x1 classifier), eval-mode BatchNorm,
random-init weights, fp32.  A step = one pair through build_cost_volume + the 3D stack, inputs
resident in HBM.  Multi-GPU = one process per GPU, each with its own pair (batch sharding, no
data-path collective): weak scaling.

One JSON line on stdout (rank 0):
  value / ms_per_step  the step through the fused entry point GlobalStack.forward_pair (factored first
                       convolution); `materialized` repeats it through the reference's own operator API
                       (build_cost_volume(...) then the modules) -- config.entry_points says which is which
  roofline             dominant kernel (the second 3x3x3 convolution + side head on the headline path), events on the
                       launch stream inside the timed loop.  `frac` = EXECUTED MFMA flops / launch time / fp32-MFMA peak (<= 1: the
                       Winograd F(4,3) kernel issues 6 of the direct form's 12 multiplies);
                       `algorithmic_over_peak` prices the convolution's algorithmic flops instead
  roofline_hbm         the headline path's HBM-bound kernel (sheared expand: the first layer's 0.74 GB write stream);
                       sub-entries: the any-shift expand, the cost-volume builders, the gather.  Same event timing
  parity_vs_cpu_baseline  every timed leg's output against the CPU oracle's on the same inputs and weights
  configs              the other BASELINE configs on this GPU (N=1 only): cfg3 96^3 crops, the released
                       local shape, cfg5 high-res, cfg4 training step; each with its dominant kernel's
                       executed pipe fraction; gather bandwidth on projected and on uniform coordinates
  train                (every N) cfg4 step incl. the RCCL flat-bucket gradient all-reduce
  cpu_baseline         CPU oracle (C/OpenMP cost volume + torch-CPU stack) on one full cfg2 pair, rank 0, N=1
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

C, H, W, D = 32, 96, 312, 192
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: BF16/F16 dense (no sparsity)
PEAK_HBM_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E
CONV1_FLOP = 2.0 * D * H * W * (2 * C) * C * 27           # algorithmic FLOP of the materialised first conv
STEP_FLOP = 1332.0e9                                      # SURVEY.md section 8(d), cfg2 3D stack
CV_BYTES = 4.0 * (2 * C * D * H * W + 2 * C * H * W + D)  # a1 algorithmic bytes per pair = 1479.9 MB
CV_RIGHT_BYTES = 4.0 * (C * D * H * W + C * H * W + D)    # right half only


def wino_executed_share(ksize, w, tile_w=32):
    """Share of a layer's algorithmic multiply-adds the Winograd F(4,k)-along-W kernels put on the
    matrix pipe: (k+3)/(4k) x the padding of W to whole tiles."""
    pad = (-(-w // tile_w) * tile_w) / float(w)
    return (ksize + 3.0) / (4.0 * ksize) * pad


def make_inputs(rank, device, d=D):
    r = np.random.default_rng(1234 + rank)
    left = torch.from_numpy(r.standard_normal((1, C, H, W)).astype(np.float32)).to(device)
    right = torch.from_numpy(r.standard_normal((1, C, H, W)).astype(np.float32)).to(device)
    shift = torch.from_numpy(np.linspace(0.0, (d - 1) / 2.0, d, dtype=np.float32)[None].copy()).to(device)
    return left, right, shift


def seeded_state(model, seed=2024):
    """Random-init weights (kaiming, as the reference) + non-trivial BatchNorm statistics."""
    g = np.random.default_rng(seed)
    sd = model.state_dict()
    for k, v in sd.items():
        if k.endswith("running_mean"):
            sd[k] = torch.from_numpy(g.uniform(-0.2, 0.2, tuple(v.shape)).astype(np.float32))
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(g.uniform(0.5, 1.5, tuple(v.shape)).astype(np.float32))
        elif v.dim() == 1 and k.endswith("weight"):
            sd[k] = torch.from_numpy(g.uniform(0.5, 1.5, tuple(v.shape)).astype(np.float32))
        elif v.dim() == 1 and k.endswith("bias"):
            sd[k] = torch.from_numpy(g.uniform(-0.2, 0.2, tuple(v.shape)).astype(np.float32))
        elif v.dim() >= 4:
            fan_in = int(np.prod(v.shape[1:]))
            sd[k] = torch.from_numpy((g.standard_normal(tuple(v.shape)) * np.sqrt(2.0 / fan_in)).astype(np.float32))
    return sd


# ------------------------------------------------------------------------------------------ CPU baseline
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


def kernel_source_hash(rel_path, marker):
    """sha256 of one kernel's source text: from the line containing `marker` to the first line that is just "}".  profiles/*/traffic.json
    records it when the PMC passes are turned into a file; bench.py quotes those counters only while the kernel's text is unchanged."""
    import hashlib
    try:
        with open(os.path.join(ROOT, rel_path)) as fh:
            lines = fh.read().split("\n")
    except OSError:
        return None
    for i, ln in enumerate(lines):
        if marker in ln:
            for j in range(i, len(lines)):
                if lines[j] == "}":
                    return hashlib.sha256("\n".join(lines[i:j + 1]).encode()).hexdigest()[:16]
    return None


X3Q_SOURCE = ("snvc_amd/csrc/conv3d_f16.hip", "conv3d_x3q_kernel(const F16Args a) {")


def parity_vs(got, exp, rel=1e-3):
    """The timed path's output against the CPU oracle's on the same inputs and weights: max|err| / max|ref| and north_star's
    1e-3 criterion element by element (|err| <= rel*|ref| + rel*rms(ref); the same rule as tests/test_gpu_parity.py::check)."""
    a = np.asarray(got, dtype=np.float64).ravel()
    b = np.asarray(exp, dtype=np.float64).ravel()
    if a.shape != b.shape:
        return {"error": f"shape {a.shape} vs {b.shape}"}
    err = np.abs(a - b)
    bound = rel * np.abs(b) + rel * max(float(np.sqrt(np.mean(b * b))), 1e-30)
    return {"rel_err": float(err.max() / max(np.abs(b).max(), 1e-30)), "elementwise_fail_frac": float((err > bound).mean()),
            "elementwise_worst_over_bound": float((err / bound).max()), "elements": int(b.size)}


# ------------------------------------------------------------------------------------------ helpers
def x3_power_probe(device):
    """The dominant kernel's launch (split-mode conv2, 32 -> 32 on 192 x 96 x 312, same instruction stream, addresses and bytes)
    on dense random operands and on all-zero operands: the difference is clock the chip gives up to operand switching in the
    matrix pipe under its power limit -- the part of `roofline.frac`'s distance from 1 that no schedule removes."""
    from snvc_amd import ops
    out = {}
    for kind in ("random", "zeros"):
        xin = torch.relu(torch.randn(1, C, D, H, W, device=device)) if kind == "random" else torch.zeros(1, C, D, H, W, device=device)
        wt = (torch.randn(C, C, 3, 3, 3, device=device) * 0.05) if kind == "random" else torch.zeros(C, C, 3, 3, 3, device=device)
        lay = ops.Conv3dLayerX3(wt)
        xs = ops.to_split(xin, 4)
        del xin
        ys = torch.empty_like(xs)
        flag = torch.zeros(1, dtype=torch.int32, device=device)
        ms, _ = timed_ms(lambda: lay(xs, 4, flags=ops.EPI_RELU, out=ys, out_exp=4, overflow=flag), 30, 5)
        out[kind + "_operands_ms"] = ms
        del xs, ys
    torch.cuda.empty_cache()
    out["note"] = ("conv2's launch without the side head, back to back: all-zero activations and weights (no switching in the matrix "
                   "pipe) against dense random ones -- the layer is limited by the chip's power budget, not by a stall")
    return out


_SYSFS_DEV = {}


def _sysfs_device_dir(index=0):
    """/sys/bus/pci/devices/<address> of THIS process's GPU `index` (the host may expose the other GPUs of the node in sysfs too:
    matched by PCI address, never by card number)."""
    if index in _SYSFS_DEV:
        return _SYSFS_DEV[index]
    path = None
    try:
        p = torch.cuda.get_device_properties(index)
        addr = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        cand = os.path.join("/sys/bus/pci/devices", addr)
        if os.path.isdir(cand):
            path = cand
    except Exception:
        path = None
    _SYSFS_DEV[index] = path
    return path


def read_gpu_clock_mhz(index=0):
    """The shader clock the driver reports right now for this process's GPU (sysfs, no subprocess: a 20-us file read between
    blocks of steps), or None when the node does not expose it."""
    import glob
    d = _sysfs_device_dir(index)
    if d is None:
        return None
    try:
        with open(os.path.join(d, "pp_dpm_sclk")) as fh:
            for ln in fh:
                if "*" in ln:
                    return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
    except (OSError, ValueError, IndexError):
        pass
    for path in sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*", "freq1_input"))):
        try:
            with open(path) as fh:
                hz = float(fh.read().strip())
            if hz > 0:
                return hz / 1e6
        except (OSError, ValueError):
            pass
    return None


def sustained_leg(step, seconds=5.0, min_steps=2000, block=100):
    """The headline step back to back for >= `seconds` AND >= `min_steps` steps, timed in blocks of `block` steps (one sync per
    block): what a deployment that runs the step continuously sees, on a part that is already warm (this leg runs after the extras)."""
    torch.cuda.synchronize()
    blocks, clocks = [], []
    t_start = time.perf_counter()
    gc.collect()
    gc.disable()
    try:
        while True:
            t0 = time.perf_counter()
            for _ in range(block):
                step()
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / block)
            c = read_gpu_clock_mhz()
            if c is not None:
                clocks.append(c)
            if len(blocks) * block >= min_steps and time.perf_counter() - t_start >= seconds:
                break
    finally:
        gc.enable()
    total_s = time.perf_counter() - t_start
    n = len(blocks) * block
    ms = [1e3 * b for b in blocks]                 # per-step time of each block
    busy_s = block * sum(blocks)                   # seconds inside the blocks (the clock readings between them excluded)
    return {"steps": n, "seconds": total_s, "pairs_per_s": n / busy_s, "ms_per_step": 1e3 * busy_s / n,
            "first_100_ms_per_step": ms[0], "last_100_ms_per_step": ms[-1], "first_100_vs_last_100": ms[0] / ms[-1],
            "slowest_block_ms_per_step": max(ms), "fastest_block_ms_per_step": min(ms),
            "sclk_mhz": ({"mean": float(np.mean(clocks)), "min": float(np.min(clocks)), "max": float(np.max(clocks)),
                          "source": "sysfs pp_dpm_sclk (current level) / hwmon freq1_input of this GPU's PCI device, one reading per 100-step block"} if clocks else None)}


PREWARM_S = 1.5


def prewarm(fn, seconds=PREWARM_S, fixed=None):
    """Runs `fn` for `seconds` before a leg's W warm-up steps.  After any idle stretch (model set-up, the host work between legs)
    the GPU needs ~50 ms of load to reach its sustained clocks: measured on the cfg2 step, the first 20-step window after an idle
    second reads 2.54-2.56 ms/step, every later one 2.42-2.45 (tools/clock_ramp.py).  W = 5 steps are 13 ms, so without this a
    20-step measurement sits inside that transient; what is reported is the sustained rate.  Untimed, disclosed in the line
    (`config.prewarm`).  r5: 0.15 -> 0.5 s -- the sustained leg's blocks of 100 steps show the first 0.22 s of load still 2 % slower
    than the steady state (2.168 against 2.114-2.13 ms/step), and the first leg of the process (`value`) read 3 % under the legs
    behind it (2.189 against 2.116-2.16); later r5: 0.5 -> 1.5 s -- box to box the ramp differs (one box: value 2.082 ms against a sustained
    2.033 after 0.5 s; another: 1.976 against 1.972); the `sustained` entry (>= 5 s) is the number to hold `value` against."""
    torch.cuda.synchronize()
    if fixed is not None:       # a step with a collective in it: every rank must run the SAME number of steps
        for _ in range(fixed):
            fn()
        torch.cuda.synchronize()
        return
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()


def timed_ms(fn, reps=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


def projected_coordinates(n, grid, device, res=256.0):
    """grid_proj_left / grid_proj_right as the data loader would produce them: GridProjector (the HIP
    restatement of refinementDataset._generate_grid_proj) on KITTI-like calibration, car-sized boxes, and a
    crop affine that maps each box's projected bounding rectangle (+20 %) onto the res x res RoI crop."""
    from snvc_amd.geometry import GridProjector
    P2 = np.array([7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01, 0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01,
                   0.0, 0.0, 1.0, 2.745884e-03]).reshape(3, 4)
    P3 = P2.copy()
    P3[0, 3], P3[1, 3] = -3.395242e+02, 2.199936e+00
    r = np.random.default_rng(99)
    samples = np.stack([np.array([1.5 + 0.1 * r.random(), 1.6 + 0.1 * r.random(), 3.9 + 0.4 * r.random(),
                                  r.uniform(-8, 8), 1.65, r.uniform(8, 40), r.uniform(-np.pi, np.pi)]) for _ in range(n)])
    xr, yr, zr = (-1.6, 1.6), (-0.8, 0.8), (-2.4, 2.4)
    tl, tr = np.zeros((n, 2, 3)), np.zeros((n, 2, 3))
    for i, s in enumerate(samples):
        ry = s[6] + 0.5 * np.pi
        rot = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
        corners = np.array([[x, y, z] for x in xr for y in yr for z in zr]).T
        cam = rot @ corners + np.array([[s[3]], [s[4] - 0.5 * s[0]], [s[5]]])
        for P, t in ((P2, tl), (P3, tr)):
            uvw = P @ np.vstack([cam, np.ones((1, 8))])
            uv = uvw[:2] / uvw[2:]
            lo, hi = uv.min(1), uv.max(1)
            ctr, ext = 0.5 * (lo + hi), 1.2 * (hi - lo)
            t[i, 0, 0], t[i, 1, 1] = res / ext[0], res / ext[1]
            t[i, 0, 2], t[i, 1, 2] = 0.5 * res - ctr[0] * t[i, 0, 0], 0.5 * res - ctr[1] * t[i, 1, 1]
    cfg = types.SimpleNamespace(x_range=xr, y_range=yr, z_range=zr, grid_resolution=list(grid))
    return GridProjector(cfg).generate(samples, P2, P3, tl, tr, device)


_ORACLES = {}


def local_parity(grid, F, device, precision, sample_grid=None):
    """``parity_vs_cpu_baseline`` + ``cpu_baseline`` of a local-model config: ONE crop through the CPU oracle (bench.local_oracle: numpy
    gather + torch-CPU trunk, every host core) and through the HIP path (the config's own arithmetic) on the same inputs and
    weights; bev and occupancy compared on all elements.  ``sample_grid``: a smaller grid of the same model when the full one would
    take the oracle more than ~30 s (cfg5: 17.7 TFLOP per RoI); the crops/s figure is then scaled by the voxel ratio and says so."""
    g = tuple(sample_grid or grid)
    key = (g, F)
    if key not in _ORACLES:
        _ORACLES.clear()                                   # one oracle's tensors at a time
        o = local_oracle(g, F, 1)                           # torch's default thread count (see cpu_baseline)
        _ORACLES[key] = {k: o[k] for k in ("lf", "rf", "gl", "gr", "bev", "occupancy", "gather_s", "trunk_s", "cores")}
    o = _ORACLES[key]
    m = local_model(g, F, device)
    m.precision = "f16" if precision == "f16" else "auto"
    lf, rf, gl, gr = (torch.from_numpy(o[k]).to(device) for k in ("lf", "rf", "gl", "gr"))
    with torch.no_grad():
        if precision == "f16":
            bev, occ, _ = m.trunk_3d_f16(m.construct_voxel_f16(lf, rf, gl, gr))
        else:
            vs = m.construct_voxel_x3(lf, rf, gl, gr)
            bev, occ, _ = m.trunk_3d(vs if vs is not None else m.construct_voxel(lf, rf, gl, gr))
    sec = o["gather_s"] + o["trunk_s"]
    scale = float(np.prod(grid)) / float(np.prod(g))
    res = {"parity_vs_cpu_baseline": {"bev": parity_vs(bev.cpu().numpy(), o["bev"].numpy()),
                                      "occupancy": parity_vs(occ.cpu().numpy(), o["occupancy"].numpy()),
                                      "sample": f"1 crop {g[0]}x{g[1]}x{g[2]}, F={F}, uniform coordinates in [-8, 264) px, all elements",
                                      "tolerance": ("fp16 STORAGE: bev max|err| <= 2e-2 rms, occupancy <= 5e-3 (tests/test_gpu_f16.py)" if precision == "f16"
                                                    else "north_star 1e-3 relative fp32; tests/test_gpu_fullsize_oracle_local.py asserts rel_err <= 1e-4")},
           "cpu_baseline": {"value": 1.0 / (sec * scale), "unit": "RoI-crops/s", "cores": o["cores"], "kind": "port",
                            "sample": f"1 crop {g[0]}x{g[1]}x{g[2]}" + (f" (x{scale:.0f} voxels to the config's grid)" if scale != 1.0 else "") +
                                      f": gather (numpy, 1 thread) {o['gather_s']:.2f}s + trunk (torch-CPU {torch.__version__}, {o['cores']} threads) {o['trunk_s']:.2f}s"}}
    if precision == "f16":
        ref = o["bev"].numpy().astype(np.float64)
        res["parity_vs_cpu_baseline"]["bev"]["max_err_over_rms"] = float(np.abs(bev.cpu().numpy() - ref).max() / np.sqrt((ref * ref).mean()))
    del m
    torch.cuda.empty_cache()
    return res


def local_config(name, grid, F, crops, device, reps=20, heads=False, precision="f32"):
    """gather + 3D trunk of the local (V-A) model on `crops` RoI crops; returns the `configs` entry.
    precision "f16": the fp16-storage mode (C8 half activations / weights, fp32 accumulate; BASELINE configs[4])."""
    from snvc_amd import ops as ops_
    m = local_model(grid, F, device)
    r = np.random.default_rng(5)
    v = grid[0] * grid[1] * grid[2]
    lf = torch.from_numpy(r.standard_normal((crops, F, 64, 64)).astype(np.float32)).to(device)
    rf = torch.from_numpy(r.standard_normal((crops, F, 64, 64)).astype(np.float32)).to(device)
    # uniform coordinates in [-8, 264) px (~6 % outside the crop)
    gl = torch.from_numpy(r.uniform(-8, 264, (crops, 2, v)).astype(np.float32)).to(device)
    gr = torch.from_numpy(r.uniform(-8, 264, (crops, 2, v)).astype(np.float32)).to(device)
    pl, pr = projected_coordinates(crops, grid, device)
    f16 = precision == "f16"
    m.precision = "f16" if f16 else "auto"      # auto: the fp32 trunk in split mode (f16x3) when it qualifies; f16: fp16 STORAGE
    gather_bytes = crops * (v * (16 + (4 if f16 else 8) * F) + 2 * F * 64 * 64 * 4)
    conv1_flop = 2.0 * crops * v * (2 * F) * F * 343
    out = {"grid": list(grid), "F": F, "crops_per_call": crops, "dtype": precision}
    gather = m.construct_voxel_f16 if f16 else m.construct_voxel
    trunk = m.trunk_3d_f16 if f16 else m.trunk_3d
    conv1 = m.conv1.fused_f16 if f16 else m.conv1
    from snvc_amd.models import submodule as S_
    with torch.no_grad():
        ms_u, vox = timed_ms(lambda: gather(lf, rf, gl, gr), reps)
        ms_p, _ = timed_ms(lambda: gather(lf, rf, pl, pr), reps)
        x3_before = S_._ROUTES["x3_local_trunk"]

        def gather_for_trunk(l_, r_, a_, b_):       # what VernierScale.forward does: in split mode the gather writes the (hi, lo) pair
            vs = None if f16 else m.construct_voxel_x3(l_, r_, a_, b_)
            return vs if vs is not None else gather(l_, r_, a_, b_)
        ms, res = timed_ms(lambda: trunk(gather_for_trunk(lf, rf, pl, pr)), reps)
        x3 = S_._ROUTES["x3_local_trunk"] > x3_before         # the trunk ran in split mode
        if x3:
            ms_ps, vsp = timed_ms(lambda: m.construct_voxel_x3(lf, rf, pl, pr), reps)
            if vsp is not None:
                out["gather_projected_split"] = {"ms": ms_ps, "GBps": gather_bytes / (ms_ps * 1e-3) / 1e9,
                                                 "frac_hbm": gather_bytes / (ms_ps * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                 "note": "snvc_voxel_gather_forward_split: the same samples written as the split C8 pair the "
                                                         "trunk starts from (same bytes as the fp32 tensor; incl. the scale from the features' maximum)"}
            del vsp
        if x3:      # its dominant layer: conv1 (k7) in split mode, on the split pair of the same voxel tensor
            mul_ = ops_.split_scale_for(vox)
            vs_ = S_.SplitT(ops_.to_split(vox, mul_dev=mul_), 0, None, mul_)
            ms_c1, _ = timed_ms(lambda: m.conv1.fused_x3(vs_), reps)
            ms_c1_f32, _ = timed_ms(lambda: m.conv1(vox), 5, 2)
            del vs_
        else:
            ms_c1, _ = timed_ms(lambda: conv1(vox), reps)
        del vox
        assert torch.isfinite(res[0]).all()
        if heads:
            ms_h, _ = timed_ms(lambda: m.heads_2d(res[0]), reps)
            out["heads_2d_ms_per_crop"] = ms_h / crops
            # the neck is ~36 small launches whose time barely depends on the crops in the call (0.95 ms for 1 or 2 crops):
            # the same call on 8 crops (BASELINE configs[2]'s crops per GPU) beside it
            bev8 = res[0].repeat((8 + crops - 1) // crops, 1, 1, 1)[:8].contiguous()
            ms_h8, _ = timed_ms(lambda: m.heads_2d(bev8), reps)
            out["heads_2d_ms_per_crop_at_8_crops"] = ms_h8 / 8
            del bev8
            # everything after the backbone (gather + trunk + 2D neck + heads) through VernierScale.forward
            del res
            ms_e, _ = timed_ms(lambda: m(lf, rf, pl, pr), reps)
            out["forward_ms_per_crop"] = ms_e / crops
    out["arithmetic"] = ("fp16 storage (C8 half activations / weights, fp32 accumulate)" if f16 else
                         "fp32 tensors; 3D trunk in split mode (f16x3: three half-precision MFMAs per fp32 product, fp32 accuracy)" if x3 else
                         "fp32 (Winograd F(4,k) on fp32 MFMA)")
    if f16:     # direct form: every algorithmic multiply-add is executed (+ one padding slot: 2 x 43 quads of taps for 343)
        kernel = f"conv3d_q16s_kernel<k7> {2 * F}->{F} (direct, v_mfma_f32_16x16x32_f16: four taps per MFMA over the flat tap list, C8 half storage)"
        frac = conv1_flop / (ms_c1 * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS
    elif x3:    # 3 MFMAs per product, 344 tap slots for 343 taps
        kernel = (f"conv3d_q16s_kernel<k7, split mode, planes serial> {2 * F}->{F} (three v_mfma_f32_16x16x32_f16 per fp32 product, "
                  "four taps per MFMA over the flat tap list)")
        frac = 3.0 * (344.0 / 343.0) * conv1_flop / (ms_c1 * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS
        out["dominant_fp32_winograd_ms"] = ms_c1_f32
    else:
        kernel = f"conv3d_winok_kernel<k7> {2 * F}->{F} (Winograd F(4,7) along W, fp32 MFMA)"
        frac = conv1_flop * wino_executed_share(7, grid[2]) / (ms_c1 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS
    out.update({
        "ms_per_crop": ms / crops, "crops_per_s": 1e3 * crops / ms,
        "dominant_kernel": kernel,
        "dominant_ms": ms_c1, "dominant_gflop_algorithmic": conv1_flop / 1e9,
        "dominant_pipe_frac": frac, "dominant_peak_tflops": PEAK_F16_MFMA_TFLOPS if (f16 or x3) else PEAK_F32_MFMA_TFLOPS,
        "gather_projected": {"ms": ms_p, "GBps": gather_bytes / (ms_p * 1e-3) / 1e9,
                             "frac_hbm": gather_bytes / (ms_p * 1e-3) / 1e9 / PEAK_HBM_GBS,
                             "coords": "GridProjector on KITTI-like calibration, car-sized boxes"},
        "gather_uniform": {"ms": ms_u, "GBps": gather_bytes / (ms_u * 1e-3) / 1e9,
                           "frac_hbm": gather_bytes / (ms_u * 1e-3) / 1e9 / PEAK_HBM_GBS,
                           "coords": "uniform in [-8, 264) px (SURVEY 8d)"},
        "gather_bytes_algorithmic": gather_bytes,
    })
    del m
    torch.cuda.empty_cache()
    return out


def off_fast_path(device, reps=10):
    """What a caller pays OFF the default inference path (one-line entries): GroupNorm models
    (``convbn_3d(..., gn=True)``, reference submodule.py:49) and ``downsample != 1`` run on the fp32-MFMA kernels / the
    materialised volume, fp64 exists for the cost-volume op only (as in the reference: BuildCostVolume_cuda.cu dispatches float/double)."""
    from snvc_amd.extension.build_cost_volume import build_cost_volume
    from snvc_amd.models import submodule as S_
    from snvc_amd.models.stereo_volume import GlobalStack
    out = {}
    with torch.no_grad():
        # 1. the local trunk with GroupNorm (statistics of each conv result: no folded affine, no a-priori range -> fp32-MFMA kernels)
        grid, F, crops = (32, 128, 192), 32, 2
        pl, pr = projected_coordinates(crops, grid, device)
        lf, rf = (torch.from_numpy(a).to(device) for a in local_inputs(grid, F, crops, 5)[:2])
        # (r5: a GroupNorm trunk runs in split mode too -- convolution in split mode with an fp32 result, statistics, one affine pass
        # that writes the split pair; `released_trunk_groupnorm_fp32_mfma` is the same model with that switched off = r4's behaviour)
        for tag, gn, prec, x3gn in (("released_trunk_groupnorm", True, "auto", True), ("released_trunk_groupnorm_fp32_mfma", True, "auto", False),
                                    ("released_trunk_batchnorm_fp32_mfma", False, "f32", True)):
            m = local_model(grid, F, device, gn=gn)
            m.precision = prec
            S_.X3_GROUP_NORM[0] = x3gn
            try:
                b = S_._ROUTES["x3_local_trunk"]
                ms, _ = timed_ms(lambda: m.trunk_3d(m.construct_voxel(lf, rf, pl, pr)), reps, 3)
            finally:
                S_.X3_GROUP_NORM[0] = True
            out[tag] = {"ms_per_crop": ms / crops, "crops_per_s": 1e3 * crops / ms, "split_mode": S_._ROUTES["x3_local_trunk"] > b}
            del m
        # 2. the global stack with GroupNorm
        left, right, shift = make_inputs(0, device)
        g = GlobalStack(C, gn=True)
        g.load_state_dict(seeded_state(g))
        g.eval().to(device)
        for tag, x3gn in (("cfg2_groupnorm", True), ("cfg2_groupnorm_fp32_mfma", False)):
            S_.X3_GROUP_NORM[0] = x3gn
            try:
                b = S_._ROUTES["x3_gn_tail"]
                ms, _ = timed_ms(lambda: g.forward_pair(left, right, shift, 1), reps, 3)
            finally:
                S_.X3_GROUP_NORM[0] = True
            out[tag] = {"ms_per_step": ms, "pairs_per_s": 1e3 / ms, "split_mode": S_._ROUTES["x3_gn_tail"] > b,
                        "note": "GlobalStack(gn=True): every norm needs its conv result's statistics -- the volume is built, each layer is "
                                "convolution -> statistics -> affine pass (r5: the convolutions in split mode, nothing fused around them)"}
        del g
        torch.cuda.empty_cache()
        # 3. downsample = 2: features at twice the resolution, the volume sampled at every second pixel (materialised volume)
        g = GlobalStack(C)
        g.load_state_dict(seeded_state(g))
        g.eval().to(device)
        r = np.random.default_rng(3)
        l2 = torch.from_numpy(r.standard_normal((1, C, 2 * H, 2 * W)).astype(np.float32)).to(device)
        r2 = torch.from_numpy(r.standard_normal((1, C, 2 * H, 2 * W)).astype(np.float32)).to(device)
        ms, _ = timed_ms(lambda: g(build_cost_volume(l2, r2, shift, 2)), reps, 3)
        out["cfg2_downsample_2"] = {"ms_per_step": ms, "pairs_per_s": 1e3 / ms,
                                    "note": "model(build_cost_volume(left [1,32,192,624], right, shift, 2)): same volume shape as cfg2, "
                                            "the eager op + conv1 over all 64 channels (the fused first layer is built for downsample 1)"}
        del g, l2, r2
        torch.cuda.empty_cache()
        # 4. fp64: the cost-volume op (the 3D stack has no fp64 kernels; neither does a user of the reference get one from cuDNN at speed)
        ld, rd = left.double(), right.double()
        ms, vol = timed_ms(lambda: ops_cost_volume(ld, rd, shift.double()), 5, 2)
        out["fp64_cost_volume"] = {"ms": ms, "GBps": 2 * CV_BYTES / (ms * 1e-3) / 1e9, "frac_hbm": 2 * CV_BYTES / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                   "note": "build_cost_volume in float64 (2.96 GB written); 3D convolutions are float32-only: a float64 volume "
                                           "raises in the stack"}
        del vol, ld, rd
        torch.cuda.empty_cache()
    return out


def ops_cost_volume(left, right, shift):
    from snvc_amd import ops
    return ops.cost_volume_forward(left, right, shift, 1)


def local_model(grid, F, device, gn=False):
    from snvc_amd.models.vernier import VernierScale
    cfg = types.SimpleNamespace(vernier_type="BEV_type3", backbone="hrfeat", gn=gn,
                                grid_resolution=[32, grid[1], 192], resolution=(256, 256),
                                x_range=(-1.0, 1.0), z_range=(-1.0, 1.0), num_parts=9)
    cfg.hrfeat = types.SimpleNamespace(output_channel=F, name="identity")
    cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
    m = VernierScale(cfg)
    m.load_state_dict(seeded_state(m))
    return m.eval().to(device)


def cfg3_crop_inputs(i, grid, F, fh=64, fw=64):
    """Crop i of the cfg3 job, seeded per crop: every rank could draw any crop, each draws only its own."""
    v = grid[0] * grid[1] * grid[2]
    r = np.random.default_rng(4321 + i)
    return (r.standard_normal((F, fh, fw)).astype(np.float32), r.standard_normal((F, fh, fw)).astype(np.float32),
            r.uniform(-8, 264, (2, v)).astype(np.float32), r.uniform(-8, 264, (2, v)).astype(np.float32))


def cfg3_shard_inputs(lo, hi, grid, F, device, fh=64, fw=64):
    """This rank's crops [lo, hi) as four stacked tensors (empty tensors of the right trailing shape for an empty shard)."""
    v = grid[0] * grid[1] * grid[2]
    mine = [cfg3_crop_inputs(i, grid, F, fh, fw) for i in range(lo, hi)]
    return tuple(torch.from_numpy(np.stack([c[k] for c in mine])).to(device) if mine else torch.empty((0,) + s_, device=device)
                 for k, s_ in enumerate(((F, fh, fw), (F, fh, fw), (2, v), (2, v))))


def cfg3_shard_step(m, lf, rf, gl, gr, per_call, total, grid, gather=True):
    """One step of a rank's shard: gather + trunk on its crops, `per_call` at a time, NO data-path collective; the per-crop occupancy
    volumes are optionally all-gathered into dim-0 order at the end (DataParallel's gather).  `m`: anything with construct_voxel_x3 /
    construct_voxel / trunk_3d (the model; a stub in tests/test_parallel_gloo.py)."""
    from snvc_amd import parallel as P
    occ = []
    n = lf.shape[0]
    for a in range(0, n, per_call):
        b = min(a + per_call, n)
        vox = m.construct_voxel_x3(lf[a:b], rf[a:b], gl[a:b], gr[a:b])      # split mode: the gather writes the (hi, lo) pair
        if vox is None:
            vox = m.construct_voxel(lf[a:b], rf[a:b], gl[a:b], gr[a:b])
        occ.append(m.trunk_3d(vox)[1])
    occ = torch.cat(occ) if occ else torch.empty((0, 1) + tuple(grid), device=lf.device)
    return P.gather_outputs(occ, total) if gather else occ


def run_cfg3(rank, world, device, dist, steps, warmup, barrier, total=64, per_call=8, gather=True):
    """BASELINE configs[2]: `total` object-centric RoI crops (96^3 voxels, F = 32) sharded over the ranks on dim 0
    (snvc_amd.parallel.shard: what replaces DataParallel's scatter, tools/inference_agnostic.py:472); every rank runs
    feature->voxel gather + the 3D trunk on ITS crops, `per_call` at a time, with no data-path collective; the per-crop
    occupancy volumes are optionally all-gathered at the end of a step (DataParallel's gather).  A step = all `total`
    crops; crops/s is barrier to barrier, max over ranks."""
    from snvc_amd import parallel as P
    grid, F = (96, 96, 96), 32
    m = local_model(grid, F, device)
    lo, hi = P.shard_range(total, rank, world)
    lf, rf, gl, gr = cfg3_shard_inputs(lo, hi, grid, F, device)

    def step():
        return cfg3_shard_step(m, lf, rf, gl, gr, per_call, total, grid, gather)

    with torch.no_grad():
        gc.collect()
        gc.disable()
        prewarm(step, fixed=1)   # one step (>= 0.04 s per rank); a fixed count: the step ends in an all-gather
        for _ in range(warmup):
            out = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        barrier()
        elapsed = time.perf_counter() - t0
        gc.enable()
    if dist is not None and world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all() and (not gather or out.shape[0] == total)
    res = {"workload": f"cfg3: {total} RoI crops {grid[0]}x{grid[1]}x{grid[2]}, F={F} (voxel volume [n,64,96,96,96]), "
                       f"feature->voxel gather + 3D trunk (BEV_type3), sharded on dim 0 over {world} rank(s), "
                       f"{per_call} crops per call",
           "crops_total": total, "crops_this_rank": hi - lo, "crops_per_call": per_call,
           "crops_per_s": total * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps,
           "ms_per_crop_per_gpu": 1e3 * elapsed / steps / max(hi - lo, 1),
           "step_tflops_algorithmic": total * 1907.3e9 / (elapsed / steps) / 1e12,
           "outputs_gathered": "occupancy [64,96,96,96] all-gathered per step" if (gather and world > 1) else "none (one rank)",
           "steps": steps}
    print(json.dumps({"rank_record_cfg3": {"rank": rank, "device": str(device), "world": world, "crops_this_rank": hi - lo,
                                           "steps": steps, "ms_per_step_max_over_ranks": res["ms_per_step"]}}), file=sys.stderr, flush=True)
    del m, lf, rf, gl, gr, out
    torch.cuda.empty_cache()
    return res


class TrainStep:
    """cfg4: build_cost_volume + GlobalStack forward (train-mode BatchNorm), loss = mean(cost^2), backward through
    the HIP kernels, then the flat-bucket gradient all-reduce (RCCL when world > 1)."""

    def __init__(self, rank, device, sheared=True):
        from snvc_amd.models.stereo_volume import GlobalStack
        self.sheared = sheared
        self.model = GlobalStack(C)
        self.model.load_state_dict(seeded_state(self.model))
        self.model.train().to(device)
        self.left, self.right, self.shift = make_inputs(rank, device)
        self.left.requires_grad_()
        self.right.requires_grad_()
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        self.nparam = sum(p.numel() for p in self.model.parameters())

    def __call__(self):
        from snvc_amd import parallel as P
        for p in self.model.parameters():
            p.grad = None
        self.left.grad = self.right.grad = None
        self.ev[0].record()
        out = self.model.forward_pair(self.left, self.right, self.shift, 1, sheared=self.sheared)
        loss = out.pow(2).mean()
        self.ev[1].record()
        loss.backward()
        self.ev[2].record()
        self.moved = P.all_reduce_gradients(self.model.parameters(), force=True)   # one rank too: RCCL really runs
        self.ev[3].record()
        return loss

    def phases_ms(self):
        return [self.ev[i].elapsed_time(self.ev[i + 1]) for i in range(3)]


def run_train(rank, world, device, dist, steps, warmup, barrier):
    ts = TrainStep(rank, device)
    gc.collect()
    gc.disable()
    prewarm(ts, fixed=8)         # ~0.17 s; a fixed count: the step ends in a collective
    for _ in range(warmup):
        ts()
    barrier()
    acc = np.zeros(3)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = ts()
        torch.cuda.synchronize()
        acc += np.array(ts.phases_ms())
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(loss)
    f, b, r = acc / steps
    res = {
        "workload": "cfg4: 1 pair/GPU at cfg2 size, train-mode BatchNorm, loss = mean(cost^2), fwd + bwd on the HIP "
                    "kernels + flat-bucket gradient all-reduce",
        "ms_per_step": 1e3 * elapsed / steps, "pairs_per_s": world * steps / elapsed,
        "fwd_ms": f, "bwd_ms": b, "allreduce_us": 1e3 * r, "allreduce_bytes": ts.moved, "params": ts.nparam,
        "step_tflops_algorithmic": 3 * STEP_FLOP / (elapsed / steps) / 1e12, "steps": steps,
    }
    if rank == 0 and world == 1:
        # the same step as ANY shift array takes it (sheared=False: warp after convolution, forward and -- r4 -- backward), and with
        # rounds 1-3's backward of that layer (right half built; 3D data and weight gradients over it)
        from snvc_amd.models import submodule as S
        del ts
        torch.cuda.empty_cache()
        gen = {}
        for tag, flag in (("ms_per_step", True), ("ms_per_step_built_volume_backward", False)):
            S.COMMUTED_BACKWARD[0] = flag
            try:
                tg = TrainStep(rank, device, sheared=False)
                for _ in range(3):
                    tg()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                k = max(5, steps // 2)
                for _ in range(k):
                    tg()
                    torch.cuda.synchronize()
                gen[tag] = 1e3 * (time.perf_counter() - t1) / k
                del tg
                torch.cuda.empty_cache()
            finally:
                S.COMMUTED_BACKWARD[0] = True
        gen["note"] = ("forward_pair(..., sheared=False): the first layer warps after the convolution in both directions "
                       "(snvc_warped_expand / snvc_warped_expand_backward); second figure: its backward through the built right half")
        res["general_shift"] = gen
        ts = TrainStep(rank, device)
    if dist is not None:
        # the collective alone on synthetic buckets: the 3D stack's gradients and SURVEY.md 8(d)'s 133.5 MB full model,
        # as one all-reduce and as reduce-scatter + all-gather (what `algorithm="auto"` picks from 8 MB on)
        from snvc_amd import parallel as P
        sizes = {"stack": ts.nparam * 4, "full_model_133p5MB": 133_500_000}
        res["collective_us"] = {f"{k}_{algo}": P.all_reduce_bucket(nb, device, algorithm=algo, reps=5)[0]
                                for k, nb in sizes.items() for algo in ("all_reduce", "rs_ag")}
        res["collective_backend"] = f"{dist.get_backend()} x{dist.get_world_size()}"
    del ts
    torch.cuda.empty_cache()
    return res


# ------------------------------------------------------------------------------------------ launcher
def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) BEFORE anything in
    this process touches the GPU (torch.cuda.device_count() does not initialise it), wait for them and exit
    with the worst exit code.  Rank 0 inherits stdout, so the JSON line comes out as usual."""
    ndev = torch.cuda.device_count()
    if ndev < n:
        raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible on this node")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer: the headline metric (+ extras); train: cfg4 step as the headline value")
    ap.add_argument("--config", choices=["cfg2", "cfg3"], default="cfg2",
                    help="cfg2: the headline (BASELINE configs[1]); cfg3: 64 RoI crops sharded over the ranks (configs[2])")
    ap.add_argument("--crops", type=int, default=64, help="cfg3: crops per step over all ranks")
    ap.add_argument("--crops-per-call", type=int, default=8)
    ap.add_argument("--no-gather", action="store_true", help="cfg3: skip the all-gather of the occupancy volumes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the `configs` / `train` legs")
    ap.add_argument("--breakdown", action="store_true", help="per-layer timing on stderr (extra untimed pass)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 5 s sustained-rate leg")
    ap.add_argument("--sustained-seconds", type=float, default=5.0)
    ap.add_argument("--sustained-steps", type=int, default=2000)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    rccl_note = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        joined = dist.get_world_size()      # ranks that actually joined the RCCL group
    else:
        joined = 1
        # one rank: still a real RCCL group, so that the gradient collective of the `train` leg runs through
        # RCCL on this GPU (allreduce_bytes != 0) and the init path the N > 1 runs take is exercised
        try:
            import torch.distributed as dist_mod
            if "MASTER_PORT" not in os.environ:
                s_ = socket.socket()
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
                s_.close()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist_mod.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            dist = dist_mod
            rccl_note = "one-rank nccl (RCCL) group"
        except Exception as e:      # never take the headline down
            rccl_note = f"one-rank RCCL group failed: {type(e).__name__}: {e}"

    from snvc_amd.extension.build_cost_volume import build_cost_volume
    from snvc_amd.models.stereo_volume import GlobalStack

    def barrier():
        torch.cuda.synchronize()
        if dist is not None and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode == "train":
        tr = run_train(rank, world, device, dist, args.steps, args.warmup, barrier)
        if rank == 0:
            print(json.dumps({
                "metric": "stereo-pairs/sec (training step: cost-volume + 3D CNN fwd+bwd + gradient all-reduce)",
                "value": tr["pairs_per_s"], "unit": "stereo-pairs/s", "n_gpus": joined, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": tr["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": tr["workload"], "sharding": f"batch x{world}; RCCL all-reduce of "
                           f"{tr['allreduce_bytes']} gradient bytes per step"},
                "train": tr}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    if args.config == "cfg3":
        c3 = run_cfg3(rank, world, device, dist, args.steps, args.warmup, barrier, args.crops, args.crops_per_call,
                      not args.no_gather)
        if rank == 0:
            print(json.dumps({
                "metric": "RoI-crops/sec (feature->voxel gather + 3D trunk fwd, 96^3 crops)",
                "value": c3["crops_per_s"], "unit": "RoI-crops/s", "n_gpus": joined, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": c3["ms_per_step"], "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": c3["workload"], "sharding": f"{args.crops} crops on dim 0 over {world} rank(s), "
                           "no data-path collective; " + c3["outputs_gathered"]},
                "cfg3": c3}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    model = GlobalStack(C)
    model.load_state_dict(seeded_state(model))
    model.eval().to(device)
    left, right, shift = make_inputs(rank, device)

    outs, local_elapsed = {}, {}

    def run(factored, sheared=True, commuted=True, tag="value", arithmetic=None, brackets=("volume", "conv1", "conv2")):
        """W warm-up + K timed steps; returns (seconds for the K steps, mean ms of the "conv1" bracket, of the "volume"
        bracket and of the "conv2" bracket).  sheared path: volume = Rq + the 2D convolution G + the 4-plane edge slab,
        conv1 = the expand pass (0.74 GB write) + edge-plane copies; general path: volume = the right-half cost-volume
        launch, conv1 = the first 3D convolution; conv2 = the second 3D convolution (+ side head) either way."""
        names = brackets      # the headline leg records the dominant kernel's bracket only: three brackets (six event records per step)
        #                       cost 0.66 % of the step (2.051 against 2.038 ms, interleaved), one costs nothing (2.040)
        ev = [{k: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for k in names}
              for _ in range(args.steps)]
        with torch.no_grad():
            # the cyclic garbage collector stays out of the timed region (as timeit does): a generation-2 pass over the
            # ~1e6 objects torch keeps alive costs ~35 ms, i.e. six steps, whenever its counter happens to trip.  It is run
            # BEFORE the warm-up: a 50 ms host pause between the warm-up and the timed steps lets the GPU fall out of its
            # sustained clocks again (see prewarm)
            gc.collect()
            gc.disable()
            prewarm(lambda: model.forward_pair(left, right, shift, 1, factored=factored, sheared=sheared, commuted=commuted,
                                               arithmetic=arithmetic))
            for _ in range(args.warmup):
                model.forward_pair(left, right, shift, 1, factored=factored, sheared=sheared, commuted=commuted, arithmetic=arithmetic)
            barrier()
            t0 = time.perf_counter()
            for i in range(args.steps):
                # events go to torch's current stream == the stream the kernels are launched on
                out = model.forward_pair(left, right, shift, 1, factored=factored, timing=ev[i], sheared=sheared, commuted=commuted,
                                         arithmetic=arithmetic)
            torch.cuda.synchronize()
            local_elapsed[tag] = time.perf_counter() - t0     # this rank's own K steps (before it waits for the others)
            barrier()
            elapsed = time.perf_counter() - t0
            gc.enable()
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert torch.isfinite(out).all()
        def mean(k):
            try:
                return float(np.mean([e[k][0].elapsed_time(e[k][1]) for e in ev]))
            except (RuntimeError, ValueError, KeyError):      # a bracket this path / this leg does not record
                return float("nan")
        outs[tag] = out.cpu().numpy() if rank == 0 and world == 1 else None    # 23 MB, compared with the CPU oracle below
        return elapsed, mean("conv1"), mean("volume"), mean("conv2")

    # Headline: GlobalStack.forward_pair.  cfg2's disparity planes are uniformly spaced (linspace(0, 95.5, 192): half-pixel
    # steps), so the first convolution over the warped half runs as a 2D convolution along the shear (csrc/sheared_conv.hip).
    # The same step on the general path (any shift array: factored first convolution over the built right half) and
    # through the reference's operator API (build_cost_volume + conv1 over all 64 channels) is timed in the same process.
    from snvc_amd.models import submodule as S_
    routes0, routes_x3 = S_._ROUTES["sheared_first_conv"], S_._ROUTES["x3_tail"]
    elapsed, _, _, conv2_ms = run(True, brackets=("conv2",))
    sheared_taken = S_._ROUTES["sheared_first_conv"] > routes0
    x3_taken = S_._ROUTES["x3_tail"] > routes_x3          # conv2 + hourglass on the split-mode (f16x3) kernels
    # what reading the split-mode overflow flag INSIDE the call costs (r5: the default): the same leg with the flag only posted
    model.overflow_check = "deferred"
    elapsed_deferred = run(True, tag="deferred_overflow_check", brackets=("conv2",))[0]
    model.check_overflow()
    model.overflow_check = "call"
    # the first layer's own brackets (prep chains, expand pass) for `roofline_hbm`: the headline step once more with all three brackets
    _, expand_ms, shear_prep_ms, _ = run(True, tag="first_layer_brackets")
    # ... and r4's tail (conv5 -> fp32 `post` -> the one-channel transposed layer as its own VALU kernel) for comparison
    model.fused_tail = False
    elapsed_tail2 = run(True, tag="two_launch_tail")[0]
    model.fused_tail = True
    # the same step with conv2 and the hourglass on the fp32-MFMA kernels (r1-r3's arithmetic: Winograd F(4,3), v_mfma_f32_32x32x2_f32)
    elapsed_f32, expand_ms_f32, _, conv2_ms_f32 = run(True, tag="fp32_mfma", arithmetic="fp32")
    elapsed_gen, warp_expand_ms, warp_prep_ms, _ = run(True, sheared=False, tag="general_shift")      # any shift array: warp after convolution
    elapsed_built, conv_ms, cvr_ms, _ = run(True, sheared=False, commuted=False, tag="built_right_half")   # right half built + 3D convolution over it
    elapsed_mat, conv_ms_mat, cv_ms, _ = run(False, tag="materialized")

    def run_reference_api():
        """the reference's call sequence, verbatim: volume = build_cost_volume(l, r, s, 1); cost = model(volume)"""
        with torch.no_grad():
            gc.collect()
            gc.disable()
            prewarm(lambda: model(build_cost_volume(left, right, shift, 1)))
            for _ in range(args.warmup):
                model(build_cost_volume(left, right, shift, 1))
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = model(build_cost_volume(left, right, shift, 1))
            barrier()
            dt = time.perf_counter() - t0
            gc.enable()
        assert torch.isfinite(out).all()
        outs["reference_api"] = out.cpu().numpy() if rank == 0 and world == 1 else None
        return dt
    elapsed_api = run_reference_api()
    dom_flop = CONV1_FLOP / 2                       # a 32 -> 32 channel 3x3x3 layer on the full grid (conv2; conv1's right half)
    share = wino_executed_share(3, W)               # F(4,3): 6 of 12 multiplies x padding of W=312 to 320
    # split mode: three half-precision MFMAs per product, 28 tap slots for 27 taps (two taps per MFMA), W = 312 on 32-wide tiles
    share_x3 = 3.0 * (28.0 / 27.0) * ((-(-W // 32) * 32) / float(W))
    dom_ms = conv2_ms if sheared_taken else conv_ms
    exec_tflops = dom_flop * (share_x3 if x3_taken else share) / (dom_ms * 1e-3) / 1e12
    # what `frac` prices: the flops the arithmetic NEEDS on the pipe it runs on -- split mode: three half-precision MFMA
    # flops per fp32 product (no padding); fp32 Winograd form: the algorithm's 6 of 12 multiplies -- not what the tiling pads on top
    need_tflops = dom_flop * (3.0 if x3_taken else 0.5) / (dom_ms * 1e-3) / 1e12
    alg_tflops = dom_flop / (dom_ms * 1e-3) / 1e12
    dom_peak = PEAK_F16_MFMA_TFLOPS if x3_taken else PEAK_F32_MFMA_TFLOPS
    exec_tflops_f32 = dom_flop * share / (conv2_ms_f32 * 1e-3) / 1e12
    V1_BYTES = 4.0 * C * D * H * W                  # the first layer's output, written once by the expand pass
    alg_tflops_mat = CONV1_FLOP / (conv_ms_mat * 1e-3) / 1e12
    # HBM bytes per launch: PMC counters cannot be read from inside this process; separate rocprofv3 --pmc
    # passes (FETCH_SIZE, WRITE_SIZE, gfx950 correction) are committed under profiles/
    traffic, traffic_src, traffic_conv2 = None, None, None
    traffic_x3, traffic_x3_rel, traffic_stale = None, None, None
    for rel in ("profiles/r5/traffic.json", "profiles/r4/traffic.json", "profiles/r3/traffic.json", "profiles/r2/traffic.json", "profiles/r1/traffic.json"):
        try:
            with open(os.path.join(ROOT, rel)) as fh:
                tj = json.load(fh)
            if traffic_x3 is None:
                ent = tj.get("layers", {}).get("x3_conv2", {})
                traffic_x3 = ent.get("hbm_bytes_corrected")
                if traffic_x3 is not None:
                    traffic_x3_rel = rel
                    # the counters belong to the kernel text they were collected on: a changed kernel voids them
                    then, now = ent.get("kernel_source_sha256_16"), kernel_source_hash(*X3Q_SOURCE)
                    if then is None or then != now:
                        traffic_stale = (f"{rel}: collected on kernel source {then}, the kernel is now {now}: re-run tools/pmc_r5_traffic.sh "
                                         "+ tools/make_traffic_json.py r5")
            if traffic_conv2 is None:
                traffic_conv2 = tj.get("layers", {}).get("conv2_side", {}).get("hbm_bytes_corrected")
            if traffic is None:
                traffic = tj.get("conv1_right_wino43_dma_k3_32to32_cfg2", {}).get("hbm_bytes_corrected")
            if traffic is not None and traffic_src is None:
                traffic_src = rel + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        except Exception:
            pass
    if sheared_taken and traffic_conv2 is not None:
        traffic_src = "profiles/r3/traffic.json, layer conv2_side (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    if x3_taken:
        traffic_conv2 = traffic_x3 if traffic_stale is None else None
        traffic_src = (f"{traffic_x3_rel}, layer x3_conv2 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; kernel source hash checked)"
                       if traffic_conv2 is not None else traffic_stale)

    # every rank (not only rank 0) leaves one line on stderr: a multi-GPU run can be audited rank by rank
    print(json.dumps({"rank_record": {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(device),
                                      "world_joined": joined, "steps": args.steps, "ms_per_step_this_rank": 1e3 * local_elapsed["value"] / args.steps,
                                      "ms_per_step_max_over_ranks": 1e3 * elapsed / args.steps,
                                      "split_mode": bool(x3_taken), "sclk_mhz": read_gpu_clock_mhz()}}), file=sys.stderr, flush=True)
    x3_state = model.__dict__.get("_snvc_x3")
    x3_overflow = int(x3_state["flag"].item()) if x3_state is not None else None       # 0: no value was clamped to half's range
    x3_exponents = dict(x3_state["exp"]) if x3_state is not None else None
    if args.breakdown and rank == 0:
        _breakdown(model, left, right, shift, build_cost_volume)
    del model
    torch.cuda.empty_cache()
    power_probe = None
    if x3_taken and rank == 0 and not args.no_extras:
        power_probe = x3_power_probe(device)

    line = None
    if rank == 0:
        line = {
            "metric": "stereo-pairs/sec (cost-volume build + 3D CNN fwd)",
            "value": world * args.steps / elapsed,
            "unit": "stereo-pairs/s",
            "n_gpus": joined,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "cfg2 global scene model: 1 pair/GPU, features [1,32,96,312] (1242x375 /4), "
                            "192 disparities -> concat volume [1,64,192,96,312] -> conv3d x2 + hourglass(32) + classifier",
                "arithmetic": ("fp32 tensors in and out; conv2 + hourglass in SPLIT MODE (" + ("taken" if x3_taken else "NOT taken") + "): activations / weights "
                               "travel as (hi, lo) pairs of halves (22 significant bits), each fp32 product = three half-precision MFMAs with fp32 "
                               "accumulation -- the fp32 layers at fp32 accuracy (5e-7 of the range vs float64 per layer; the fp32 Winograd "
                               "kernels: 2e-6), held to the SAME per-layer 2e-5 / stack 1e-4 tolerances as the fp32 kernels "
                               "(tests/test_gpu_fullsize_oracle.py, parity_vs_cpu_baseline below); `fp32_mfma` repeats the step on the fp32-MFMA kernels"),
                "prewarm": (f"{PREWARM_S} s of the same step, untimed, in front of every leg's W warm-up steps: after an idle stretch the GPU "
                            "needs a few hundred ms of load to reach its sustained clocks (the `sustained` leg's first 100-step block "
                            "reads 1-4 % slower than its last; r4 used 0.15 s and its first leg read 3 % under the later ones); W = 5 "
                            "steps are 11 ms.  `value` is meant to be the sustained rate: hold it against `sustained` (>= 5 s, >= 2000 steps)"),
                "split_mode": {"taken": bool(x3_taken), "overflow_flag": x3_overflow, "tensor_exponents": x3_exponents,
                               "rule": "2^e * (|beta| + 64 |gamma|) <= 2^15 per tensor (folded eval BatchNorm); a value beyond it is clamped "
                                       "and flagged, the model then falls back to the fp32-MFMA kernels"},
                "entry_points": {
                    "value": "GlobalStack.forward_pair(left, right, shift): fused entry point.  Left half of the concat volume: "
                             "d-invariant -> 3 depth-class planes.  Warped right half: the disparity planes are uniformly "
                             "spaced (shift = d/2), so it is a shear of one 2D image and conv1 over it is a 2D convolution "
                             "evaluated along the shear (" + ("taken" if sheared_taken else "NOT taken") + "; csrc/sheared_conv.hip; "
                             "tests/test_gpu_parity.py::test_sheared_first_conv_vs_oracle_and_general_path, tests/test_gpu_fullsize.py)",
                    "general_shift": "the same entry point for ANY shift array (sheared=False): interpolation along w commutes with "
                                     "the convolution -- three 2D convolutions of the right feature + three interpolations per "
                                     "output voxel (snvc_warped_expand); the warped volume is not built either",
                    "built_right_half": "forward_pair(..., sheared=False, commuted=False): right half of the volume built, factored "
                                        "first 3D convolution over it (r2's path)",
                    "reference_api": "model(build_cost_volume(left, right, shift, 1)): same kernels as `value` (lazy volume)",
                    "materialized": "the full concat volume built in HBM, then the modules (conv1 over all 64 channels)"},
                "pairs_per_gpu_per_step": 1,
                "sharding": f"batch x{world}, no collective",
                "step_gflop_algorithmic": STEP_FLOP / 1e9,
                "step_cost_volume_mb_algorithmic": CV_BYTES / 1e6,
            },
            "roofline": {
                "kernel": ("conv3d_x3q_kernel<side head>: second 3D convolution, 32->32 on 192x96x312 + the classifier's projection of its own "
                           "result, split mode (f16x3), 4x4x32 tile, 2 WG/CU; every fp32 product = three v_mfma_f32_16x16x32_f16 on (hi, lo) "
                           "half pairs, fp32 accumulate (csrc/conv3d_f16.hip; the 16x16x32 shape sustains ~20 % more than 32x32x16 under the "
                           "chip's power limit: tools/micro/mfma_power.hip)" if x3_taken else
                           "conv3d_wino_dma_kernel<4x4x32 tile, KC2, 3 WG/CU, side head>: second 3D convolution, 32->32 on 192x96x312 "
                           "+ the classifier's projection of its own result (Winograd F(4,3) along W, fp32 MFMA, LDS-DMA staged)"
                           if sheared_taken else
                           "conv3d_wino_dma_kernel<4x4x32 tile, KC2, 3 WG/CU, planes>: first conv over the right half of the volume, "
                           "32->32 on 192x96x312, + depth-class planes (Winograd F(4,3) along W, fp32 MFMA, LDS-DMA staged)"),
                "bound": "mfma",
                # `achieved` = the matrix-pipe flops the layer's arithmetic NEEDS per second: split mode = 3 half-precision MFMA flops per
                # algorithmic fp32 multiply-add, priced against the dense f16 MFMA peak (fp32 Winograd form: 6 of 12, against the fp32
                # peak).  `executed_tflops` adds what the tiling pads on top (28 tap slots for 27 taps, 320 columns for 312: +6.3 %,
                # = SQ_INSTS_MFMA x 16384 flop) and is NOT what frac counts.  `algorithmic_tflops` = 2*voxels*Cin*Cout*27 / time.
                "achieved": need_tflops,
                "peak": dom_peak,
                "unit": "TFLOP/s",
                "frac": need_tflops / dom_peak,
                "executed_tflops": exec_tflops,
                "executed_frac": exec_tflops / dom_peak,
                "algorithmic_tflops": alg_tflops,
                "algorithmic_over_fp32_mfma_peak": alg_tflops / PEAK_F32_MFMA_TFLOPS,
                "flop_per_launch_algorithmic": dom_flop,
                "flop_per_launch_executed": dom_flop * (share_x3 if x3_taken else share),
                "avg_launch_ms": dom_ms,
                "traffic": traffic if not sheared_taken else traffic_conv2,
                "traffic_source": traffic_src,
                "power_probe": power_probe,
                "fp32_mfma_form": {"kernel": "conv3d_wino_dma_kernel<4x4x32, side head> (the fp32_mfma leg's conv2: Winograd F(4,3), v_mfma_f32_32x32x2_f32)",
                                   "avg_launch_ms": conv2_ms_f32, "achieved": exec_tflops_f32, "peak": PEAK_F32_MFMA_TFLOPS,
                                   "frac": exec_tflops_f32 / PEAK_F32_MFMA_TFLOPS},
            },
            "roofline_hbm": {
                # the headline path's own HBM-bound kernel: conv1's result written along the shear (one 0.74 GB write stream)
                "kernel": "sheared_expand_kernel + 2 edge-plane copies: the first layer's output of the headline path, "
                          "written along the shear",
                "bound": "hbm", "achieved": V1_BYTES / (expand_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": V1_BYTES / (expand_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes_per_launch": V1_BYTES,
                "avg_launch_ms": expand_ms, "prep_ms": shear_prep_ms,
                "measured_in": "a repeat of the headline leg with the first layer's event brackets on (`value` itself records the conv2 bracket only: "
                               "three brackets cost 0.66 % of the step)",
                "prep": "Rq on two grids + the depth-1 3x7 convolutions G (all columns) and G' (last column), 3 depth classes each",
                "warped_expand": {"kernel": "warped_expand_kernel: the same layer for ANY shift array (general_shift leg): three "
                                            "interpolations of three 2D convolutions per voxel, same 0.74 GB write stream",
                                  "achieved": V1_BYTES / (warp_expand_ms * 1e-3) / 1e9,
                                  "frac": V1_BYTES / (warp_expand_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                  "bytes_per_launch": V1_BYTES, "avg_launch_ms": warp_expand_ms, "prep_ms": warp_prep_ms},
                "right_half_builder": {"kernel": "cost_volume_fwd_rows: right (warped) half only (built_right_half leg; on no default path)",
                                       "achieved": CV_RIGHT_BYTES / (cvr_ms * 1e-3) / 1e9,
                                       "frac": CV_RIGHT_BYTES / (cvr_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                       "bytes_per_launch": CV_RIGHT_BYTES, "avg_launch_ms": cvr_ms},
                "full_volume": {"kernel": "cost_volume_fwd_rows: build_cost_volume, both halves (materialized leg)",
                                "achieved": CV_BYTES / (cv_ms * 1e-3) / 1e9, "frac": CV_BYTES / (cv_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                "bytes_per_launch": CV_BYTES, "avg_launch_ms": cv_ms},
            },
            "overflow_check": {
                "note": "split mode clamps a value beyond the range its BatchNorm parameters promise and raises a device flag.  `value` reads "
                        "that flag INSIDE the call (4-byte copy queued behind the last layer that can clamp, waited for after the rest "
                        "of the call is queued) and redoes a flagged call on the fp32-MFMA kernels: no clamped result is ever returned "
                        "(tests/test_gpu_overflow.py).  `deferred` = the same leg with the flag only posted (r4's behaviour)",
                "checked_ms_per_step": 1e3 * elapsed / args.steps, "deferred_ms_per_step": 1e3 * elapsed_deferred / args.steps,
                "cost_ms_per_step": 1e3 * (elapsed - elapsed_deferred) / args.steps,
                "redone_calls": int(S_._ROUTES["x3_overflow_redo"]),
            },
            "two_launch_tail": {
                "note": "same step with r4's tail: conv5 writes `post` (64 channels, fp32), deconv3d_cout1_kernel reads it; `value` contracts "
                        "conv5's result with the folded tail's 27 taps in conv5's epilogue (snvc_f16x3_deconv3d_tail_forward) + snvc_deconv_tail_gather",
                "value": world * args.steps / elapsed_tail2, "ms_per_step": 1e3 * elapsed_tail2 / args.steps,
            },
            "fp32_mfma": {
                "note": "same step, same entry point, with conv2 and the hourglass on the fp32-MFMA kernels (forward_pair(..., arithmetic='fp32'): "
                        "Winograd F(4,3) / polyphase kernels on v_mfma_f32_32x32x2_f32 -- rounds 1-3's arithmetic)",
                "value": world * args.steps / elapsed_f32, "ms_per_step": 1e3 * elapsed_f32 / args.steps,
                "conv2_ms": conv2_ms_f32, "expand_ms": expand_ms_f32,
            },
            "reference_api": {
                "note": "the reference's call sequence verbatim -- volume = build_cost_volume(left, right, shift, 1); "
                        "cost = model(volume) -- under torch.no_grad(): build_cost_volume returns a LazyCostVolume "
                        "(snvc_amd/lazy.py) that GlobalStack.forward consumes on the fused path; any other use of it "
                        "builds the real volume (tests/test_gpu_parity.py::test_lazy_cost_volume_reference_call_sequence)",
                "value": world * args.steps / elapsed_api,
                "ms_per_step": 1e3 * elapsed_api / args.steps,
            },
            "general_shift": {
                "note": "same step on the path any shift array takes (forward_pair(..., sheared=False)): warp after convolution "
                        "(csrc/sheared_conv.hip: three depth-1 convolutions of the right feature + snvc_warped_expand)",
                "value": world * args.steps / elapsed_gen,
                "ms_per_step": 1e3 * elapsed_gen / args.steps,
                "expand_ms": warp_expand_ms, "prep_ms": warp_prep_ms,
            },
            "built_right_half": {
                "note": "same step with the right half of the volume built (cost_volume_fwd_rows) and the factored first 3D "
                        "convolution over it (forward_pair(..., sheared=False, commuted=False))",
                "value": world * args.steps / elapsed_built,
                "ms_per_step": 1e3 * elapsed_built / args.steps,
                "conv1_ms": conv_ms, "conv1_pipe_frac": dom_flop * share / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "volume_ms": cvr_ms,
            },
            "materialized": {
                "note": "same step with the full concat volume built in HBM (what the reference does, and what this library "
                        "does whenever the volume is written to): build_cost_volume_cuda.build_cost_volume_forward, "
                        "conv1 runs over all 64 channels",
                "value": world * args.steps / elapsed_mat,
                "ms_per_step": 1e3 * elapsed_mat / args.steps,
                "conv1_ms": conv_ms_mat,
                "conv1_tflops_algorithmic": alg_tflops_mat,
                # split mode: three half-precision MFMAs per product against the f16 peak; fp32 form: Winograd's 6/12 against the fp32 peak
                "conv1_pipe_frac": (alg_tflops_mat * share_x3 / PEAK_F16_MFMA_TFLOPS if x3_taken else alg_tflops_mat * share / PEAK_F32_MFMA_TFLOPS),
                "conv1_kernel": ("conv3d_f16_kernel<k3, split mode> 64->32 after a layout pass of the 1.47 GB volume (snvc_f16x3_from_ncdhw)"
                                 if x3_taken else "conv3d_wino_dma_kernel 64->32 (fp32 Winograd F(4,3))"),
                "conv1_flop_per_launch": CONV1_FLOP,
            },
            "step_tflops_algorithmic": STEP_FLOP / (elapsed / args.steps) / 1e12,
        }

    if not args.no_extras:
        # cfg4 on every N: the one leg with a collective (RCCL all-reduce of the 3D stack's gradients)
        tr = run_train(rank, world, device, dist, 10, 3, barrier)
        if rank == 0:
            line["train"] = tr
        if world == 1:
            cfgs = {}
            for name, grid, F, crops, heads, prec in (("cfg3_crops_96", (96, 96, 96), 32, 8, False, "f32"),
                                                      ("released_32x128x192", (32, 128, 192), 32, 2, True, "f32"),
                                                      ("released_32x128x192_f16", (32, 128, 192), 32, 2, True, "f16"),
                                                      ("cfg5_highres_80x160x160", (80, 160, 160), 64, 1, False, "f16"),
                                                      ("cfg5_highres_80x160x160_f32", (80, 160, 160), 64, 1, False, "f32")):
                try:
                    cfgs[name] = local_config(name, grid, F, crops, device, heads=heads, precision=prec)
                except Exception as e:  # an extra must never take the headline down with it
                    cfgs[name] = {"error": f"{type(e).__name__}: {e}"}
                if rank == 0 and not args.no_cpu_baseline and "error" not in cfgs[name]:
                    try:        # the same model on the oracle's inputs: whole-tensor parity + the CPU rate beside the GPU rate
                        cfgs[name].update(local_parity(grid, F, device, prec, sample_grid=(48, 80, 80) if grid == (80, 160, 160) else None))
                    except Exception as e:
                        cfgs[name]["parity_vs_cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
            _ORACLES.clear()
            try:        # BASELINE configs[2] as the N > 1 runs shard it (`--config cfg3`), here all 64 crops on one rank
                cfgs["cfg3_64crops_sharded"] = run_cfg3(rank, world, device, dist, 2, 1, barrier)
            except Exception as e:
                cfgs["cfg3_64crops_sharded"] = {"error": f"{type(e).__name__}: {e}"}
            cfgs["cfg4_train_step"] = {k: tr[k] for k in ("ms_per_step", "fwd_ms", "bwd_ms", "step_tflops_algorithmic")}
            try:        # its dominant launch: the Winograd-domain weight gradient of a 32->32 layer on the full grid
                from snvc_amd import ops                     # (conv1's right half, conv2, the classifier: 3 per step)
                xg = torch.randn(1, C, D, H, W, device=device)
                gg = torch.randn(1, C, D, H, W, device=device)
                ms_w, _ = timed_ms(lambda: ops.conv3d_wgrad(xg, gg, 3, 1, 1, 1), 3)
                flop = CONV1_FLOP / 2                          # 32 of conv1's 64 input channels
                cfgs["cfg4_train_step"].update({
                    "dominant_kernel": "conv3d_wgrad_wino_kernel 32->32 on 192x96x312 (Winograd-domain weight gradient, fp32 MFMA, "
                                       "deterministic) + wgrad_wino_reduce_kernel",
                    "dominant_ms": ms_w, "dominant_gflop_algorithmic": flop / 1e9,
                    "dominant_tflops_algorithmic": flop / (ms_w * 1e-3) / 1e12,
                    # executed on the matrix pipe: 6 of 12 multiply-adds, on 32-wide tiles (W = 312 -> 320)
                    "dominant_pipe_frac": wino_executed_share(3, W) * flop / (ms_w * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS})
                del xg, gg
            except Exception as e:
                cfgs["cfg4_train_step"]["dominant_error"] = f"{type(e).__name__}: {e}"
            line["configs"] = cfgs
            try:
                line["off_fast_path"] = off_fast_path(device)
            except Exception as e:
                line["off_fast_path"] = {"error": f"{type(e).__name__}: {e}"}
            g3 = cfgs.get("cfg3_crops_96", {}).get("gather_projected")
            if g3:      # north_star's ">= 60 % HBM roofline on the warp/gather": the gather beside the cost-volume builders
                line["roofline_hbm"]["gather"] = {
                    "kernel": "voxel_gather_fwd_lds: _sample_2d_feat on 8 crops 96^3, F=32, GridProjector coordinates",
                    "achieved": g3["GBps"], "frac": g3["frac_hbm"], "avg_launch_ms": g3["ms"],
                    "bytes_per_launch": cfgs["cfg3_crops_96"]["gather_bytes_algorithmic"]}
            if rccl_note:
                line["train"]["rccl"] = rccl_note
    if not args.no_sustained:
        # the headline step again, back to back for >= 5 s and >= 2000 steps, AFTER the extras (a warm part): the sustained rate
        model = GlobalStack(C)
        model.load_state_dict(seeded_state(model))
        model.eval().to(device)
        with torch.no_grad():
            for _ in range(3):
                model.forward_pair(left, right, shift, 1)
            sus = sustained_leg(lambda: model.forward_pair(left, right, shift, 1), args.sustained_seconds, args.sustained_steps)
        del model
        torch.cuda.empty_cache()
        if dist is not None and world > 1:      # whole-job rate: the slowest rank's
            t = torch.tensor([sus["pairs_per_s"]], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            sus["pairs_per_s_slowest_rank"] = float(t.item())
            sus["pairs_per_s"] = world * float(t.item())
        print(json.dumps({"rank_record_sustained": dict(sus, rank=rank)}), file=sys.stderr, flush=True)
        if rank == 0:
            sus["note"] = (f"`value` is the driver's 20-step window after a {PREWARM_S} s pre-warm; this is the same step for >= 5 s on a part already "
                           "warm from the other legs -- what a deployment running the step continuously sees")
            sus["vs_value"] = sus["pairs_per_s"] / line["value"]
            line["sustained"] = sus
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            oracle_out = {}
            line["cpu_baseline"] = cpu_baseline(outputs=oracle_out)
            # the oracle ran on exactly the timed legs' inputs and weights: every leg's last output against it, all 5.75 M values
            par = {k: parity_vs(v, oracle_out["cost"]) for k, v in outs.items() if v is not None}
            line["parity_vs_cpu_baseline"] = dict(par.get("value", {}), legs=par, tolerance="north_star: 1e-3 relative fp32; "
                                                  "tests/test_gpu_fullsize_oracle.py asserts rel_err <= 1e-4 and the elementwise rule")
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _breakdown(model, left, right, shift, build_cost_volume):
    with torch.no_grad():
        t, vol = timed_ms(lambda: build_cost_volume(left, right, shift, 1), 5)
        print(f"[breakdown] build_cost_volume      {t:8.3f} ms  {CV_BYTES / (t * 1e-3) / 1e9:8.1f} GB/s", file=sys.stderr)
        t, v1 = timed_ms(lambda: model.conv1(vol), 5)
        print(f"[breakdown] conv1 k3 64->32        {t:8.3f} ms  {CONV1_FLOP / (t * 1e-3) / 1e12:8.1f} TFLOP/s", file=sys.stderr)
        del vol
        t, v2 = timed_ms(lambda: model.conv2(v1), 5)
        print(f"[breakdown] conv2 k3 32->32        {t:8.3f} ms  {CONV1_FLOP / 2 / (t * 1e-3) / 1e12:8.1f} TFLOP/s", file=sys.stderr)
        t, _ = timed_ms(lambda: model.hg_conv3d(v2, None, None, residual=v2), 5)
        print(f"[breakdown] hourglass(32)          {t:8.3f} ms  {377.6e9 / (t * 1e-3) / 1e12:8.1f} TFLOP/s", file=sys.stderr)
        t, _ = timed_ms(lambda: model.classifier(v2), 5)
        print(f"[breakdown] classifier 1x1x1       {t:8.3f} ms", file=sys.stderr)


if __name__ == "__main__":
    main()
