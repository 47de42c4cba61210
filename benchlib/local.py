"""The local (V-A) model's legs of bench.py: BASELINE configs[2] (cfg3), the released shape, configs[4] (cfg5), what a caller pays off
the fast path, and the sharded 64-crop job (`--config cfg3`)."""
import gc
import time
import types

import numpy as np
import torch

from .common import (C, CV_BYTES, D, H, PEAK_F16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, W, make_inputs, parity_vs, prewarm,
                     projected_coordinates, seeded_state, timed_ms, wino_executed_share)
from .cpu import local_inputs, local_oracle
from .emit import rank_note

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
                b, b_first = S_._ROUTES["x3_gn_tail"], S_._ROUTES["gn_sheared_first_conv"]
                ms, _ = timed_ms(lambda: g.forward_pair(left, right, shift, 1), reps, 3)
            finally:
                S_.X3_GROUP_NORM[0] = True
            out[tag] = {"ms_per_step": ms, "pairs_per_s": 1e3 / ms, "split_mode": S_._ROUTES["x3_gn_tail"] > b,
                        "sheared_first_layer": S_._ROUTES["gn_sheared_first_conv"] > b_first,
                        "note": "GlobalStack(gn=True): every norm needs its conv result's statistics -- each layer is convolution -> "
                                "statistics -> affine pass in split mode; r6: on uniformly spaced planes the first layer is the sheared one "
                                "with its statistics from snvc_sheared_expand_stats (one channel per group), the 1.47 GB volume is not built"}
        del g
        torch.cuda.empty_cache()
        # 3. downsample = 2: features at twice the resolution, the volume sampled at every second pixel
        g = GlobalStack(C)
        g.load_state_dict(seeded_state(g))
        g.eval().to(device)
        r = np.random.default_rng(3)
        l2 = torch.from_numpy(r.standard_normal((1, C, 2 * H, 2 * W)).astype(np.float32)).to(device)
        r2 = torch.from_numpy(r.standard_normal((1, C, 2 * H, 2 * W)).astype(np.float32)).to(device)
        b4 = S_._ROUTES["ds_sheared_first_conv"]
        ms, _ = timed_ms(lambda: g(build_cost_volume(l2, r2, shift, 2)), reps, 3)
        out["cfg2_downsample_2"] = {"ms_per_step": ms, "pairs_per_s": 1e3 / ms, "sheared_first_layer": S_._ROUTES["ds_sheared_first_conv"] > b4,
                                    "note": "model(build_cost_volume(left [1,32,192,624], right, shift, 2)) with cfg2's half-pixel shift "
                                            "array (quarter-pixel planes of the volume: four phases): the eager op + conv1 over all 64 channels"}
        # r6: the sweep that covers cfg2's disparity range at this resolution -- planes ONE input pixel apart -- takes the sheared first layer
        shift1 = torch.arange(D, dtype=torch.float32, device=device)[None].contiguous()
        b2 = S_._ROUTES["ds_sheared_first_conv"]
        ms, _ = timed_ms(lambda: g.forward_pair(l2, r2, shift1, 2), reps, 3)
        out["cfg2_downsample_2_whole_pixel_planes"] = {
            "ms_per_step": ms, "pairs_per_s": 1e3 / ms, "sheared_first_layer": S_._ROUTES["ds_sheared_first_conv"] > b2,
            "note": "forward_pair(left [1,32,192,624], right, shift = 0..191, 2): same volume as cfg2 (the same sampling positions), the "
                    "row-subsampled right feature stands where the half-pixel upsampled one stands at downsample 1"}
        del g, l2, r2
        torch.cuda.empty_cache()
        # 4. fp64: the cost-volume op (the 3D stack has no fp64 kernels; neither does a user of the reference get one from cuDNN at speed)
        ld, rd = left.double(), right.double()
        # three measurements, the fastest kept: each call allocates its 2.96 GB result, and a run in which the caching allocator had to go
        # back to the driver for it measured 12.8 ms once (the kernel: 0.74-0.88)
        ms, vol = min((timed_ms(lambda: ops_cost_volume(ld, rd, shift.double()), 5, 3) for _ in range(3)), key=lambda t: t[0])
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
    rank_note("rank_record_cfg3", {"rank": rank, "device": str(device), "world": world, "crops_this_rank": hi - lo, "steps": steps,
                                   "ms_per_step_max_over_ranks": res["ms_per_step"]})
    del m, lf, rf, gl, gr, out
    torch.cuda.empty_cache()
    return res
