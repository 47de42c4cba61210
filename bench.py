#!/usr/bin/env python3
"""Headline benchmark: stereo-pairs/sec, cost-volume build + 3D CNN forward (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (N=1): BASELINE.json configs[1], "Global scene model: 1242x375, 192 disparities, full 3D
hourglass fwd, batch=1 on 1 MI355X", synthesised as SURVEY.md section 8(d) cfg2:
left/right features [1,32,96,312] (1242x375 padded to 1248x384, stride 4), shift =
linspace(0, 95.5, 192), downsample 1 -> concat volume [1,64,192,96,312] -> GlobalStack(32)
(conv 64->32, conv 32->32, hourglass(32) + residual, 1x1x1 classifier), eval-mode BatchNorm,
random-init weights, fp32.  A step = one pair through build_cost_volume + the 3D stack, inputs
resident in HBM.  Multi-GPU = one process per GPU, each with its own pair (batch sharding, no
data-path collective): weak scaling.

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel, the first 3x3x3
convolution (factored: 32->32 over the warped half of the volume, 318 GFLOP per launch; Winograd
F(4,3) along W on the fp32 MFMA pipe), timed with events on the launch stream inside the timed
loop: `achieved` prices the algorithmic FLOPs, `mfma_pipe_frac` the executed ones.  `materialized`
repeats the measurement with the full concat volume built and convolved (64->32, 636 GFLOP).
`cpu_baseline` times the CPU oracle (C cost volume + torch-CPU stack) on a bounded sample on rank 0
at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

C, H, W, D = 32, 96, 312, 192
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
CONV1_FLOP = 2.0 * D * H * W * (2 * C) * C * 27           # algorithmic FLOP of the dominant launch
WINO_EXECUTED = (6.0 / 12.0) * (320.0 / 312.0)                  # F(4,3) MFMA share x 64-wide tile padding of W=312
STEP_FLOP = 1332.0e9                                      # SURVEY.md section 8(d), cfg2 3D stack
STEP_BYTES = 1479.9e6                                     # cost-volume build, algorithmic bytes


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
        elif v.dim() == 5:
            fan_in = int(np.prod(v.shape[1:]))
            sd[k] = torch.from_numpy((g.standard_normal(tuple(v.shape)) * np.sqrt(2.0 / fan_in)).astype(np.float32))
    return sd


def cpu_baseline(d_sample=64, repeats=1):
    """CPU oracle on cfg1 (D = 64 planes of the same pair): C cost volume + torch-CPU 3D stack."""
    from oracle import native as O
    from oracle import torch_ref as T
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    left, right, shift = make_inputs(0, "cpu", d_sample)
    ref = T.GlobalStack(C)
    ref.load_state_dict(seeded_state(ref))
    ref.eval()
    ln, rn, sn = left.numpy(), right.numpy(), shift.numpy()
    best = None
    t_cv = t_cnn = 0.0
    for _ in range(repeats):
        t0 = time.perf_counter()
        vol = O.cost_volume_forward(ln, rn, sn, 1)
        t1 = time.perf_counter()
        with torch.no_grad():
            ref(torch.from_numpy(vol))
        t2 = time.perf_counter()
        if best is None or (t2 - t0) < best:
            best, t_cv, t_cnn = t2 - t0, t1 - t0, t2 - t1
    # scale the 64-plane time to a 192-plane pair (both stages are linear in D)
    pairs_per_s = 1.0 / (best * D / d_sample)
    return {
        "value": pairs_per_s, "unit": "stereo-pairs/s", "cores": cores, "kind": "port",
        "sample": f"1 pair at D={d_sample} of {D} planes (cfg1 size), {best:.2f}s "
                  f"(cost volume C oracle 1 thread {t_cv:.2f}s + torch-CPU 3D stack {cores} threads {t_cnn:.2f}s), "
                  f"scaled x{D // d_sample} to a {D}-plane pair; torch {torch.__version__}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="per-layer timing on stderr (extra untimed pass)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from snvc_amd.extension.build_cost_volume import build_cost_volume
    from snvc_amd.models.stereo_volume import GlobalStack

    model = GlobalStack(C)
    model.load_state_dict(seeded_state(model))
    model.eval().to(device)
    left, right, shift = make_inputs(rank, device)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(factored):
        """W warm-up + K timed steps; returns (seconds for the K steps, mean ms of the first-conv launch)."""
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        with torch.no_grad():
            for _ in range(args.warmup):
                model.forward_pair(left, right, shift, 1, factored=factored)
            barrier()
            t0 = time.perf_counter()
            for i in range(args.steps):
                # events go to torch's current stream == the stream the kernels are launched on
                out = model.forward_pair(left, right, shift, 1, factored=factored, timing=(ev0[i], ev1[i]))
            barrier()
            elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert torch.isfinite(out).all()
        return elapsed, float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))

    # Headline: factored first convolution (GlobalStack.forward_pair).  For transparency the same step
    # with the concat volume fully materialised (build_cost_volume + conv1 over all 64 channels) is
    # timed in the same process and reported alongside.
    elapsed, conv_ms = run(True)
    elapsed_mat, conv_ms_mat = run(False)
    dom_flop = CONV1_FLOP / 2                       # right half: 32 -> 32 channels, 27 taps
    achieved = dom_flop / (conv_ms * 1e-3) / 1e12
    achieved_mat = CONV1_FLOP / (conv_ms_mat * 1e-3) / 1e12
    # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this
    # process; they come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, gfx950 correction)
    # committed in profiles/r1/traffic.json.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r1", "traffic.json")) as fh:
            traffic = json.load(fh).get("conv1_right_wino43_dma_k3_32to32_cfg2", {}).get("hbm_bytes_corrected")
    except Exception:
        pass

    if args.breakdown and rank == 0:
        _breakdown(model, left, right, shift, build_cost_volume)

    if rank == 0:
        line = {
            "metric": "stereo-pairs/sec (cost-volume build + 3D CNN fwd)",
            "value": world * args.steps / elapsed,
            "unit": "stereo-pairs/s",
            "n_gpus": world,
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
                "first_conv": "factored: the left half of the concat volume is d-invariant -> 3 depth-class planes + "
                              "3D conv over the warped right half only; output identical to the materialised path "
                              "(tests/test_gpu_parity.py::test_global_pair_end_to_end_vs_oracle)",
                "pairs_per_gpu_per_step": 1,
                "sharding": f"batch x{world}, no collective",
                "step_gflop_algorithmic": STEP_FLOP / 1e9,
                "step_cost_volume_mb_algorithmic": STEP_BYTES / 1e6,
            },
            "roofline": {
                "kernel": "conv3d_wino_dma_kernel<4x4x32 tile, KC2, 3 WG/CU, planes>: first conv over the right half of the volume, "
                          "32->32 on 192x96x312, + depth-class planes (Winograd F(4,3) along W, fp32 MFMA, LDS-DMA staged)",
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                "traffic": traffic,
                "traffic_source": "profiles/r1/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                "flop_per_launch": dom_flop,
                "avg_launch_ms": conv_ms,
                # `achieved` prices the ALGORITHMIC multiply-adds of the convolution (contract); F(4,3)
                # issues 6 MFMAs where the direct form needs 12, so the matrix pipe executes half of them
                # and `frac` may exceed 1.  The pipe's own utilisation is mfma_pipe_frac.
                "executed_flop_per_launch": dom_flop * WINO_EXECUTED,
                "mfma_pipe_frac": achieved * WINO_EXECUTED / PEAK_F32_MFMA_TFLOPS,
            },
            "materialized": {
                "note": "same step with the full concat volume built by build_cost_volume and conv1 over all 64 channels",
                "value": world * args.steps / elapsed_mat,
                "ms_per_step": 1e3 * elapsed_mat / args.steps,
                "conv1_tflops": achieved_mat,
                "conv1_frac": achieved_mat / PEAK_F32_MFMA_TFLOPS,
                "conv1_flop_per_launch": CONV1_FLOP,
            },
            "step_tflops_algorithmic": STEP_FLOP / (elapsed / args.steps) / 1e12,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _breakdown(model, left, right, shift, build_cost_volume):
    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            r = fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n, r

    with torch.no_grad():
        t, vol = timed(lambda: build_cost_volume(left, right, shift, 1))
        print(f"[breakdown] build_cost_volume      {t:8.3f} ms  {STEP_BYTES / (t * 1e-3) / 1e9:8.1f} GB/s", file=sys.stderr)
        t, v1 = timed(lambda: model.conv1(vol))
        print(f"[breakdown] conv1 k3 64->32        {t:8.3f} ms  {CONV1_FLOP / (t * 1e-3) / 1e12:8.1f} TFLOP/s", file=sys.stderr)
        del vol
        t, v2 = timed(lambda: model.conv2(v1))
        print(f"[breakdown] conv2 k3 32->32        {t:8.3f} ms  {CONV1_FLOP / 2 / (t * 1e-3) / 1e12:8.1f} TFLOP/s", file=sys.stderr)
        t, _ = timed(lambda: model.hg_conv3d(v2, None, None, residual=v2))
        print(f"[breakdown] hourglass(32)          {t:8.3f} ms  {377.6e9 / (t * 1e-3) / 1e12:8.1f} TFLOP/s", file=sys.stderr)
        t, _ = timed(lambda: model.classifier(v2))
        print(f"[breakdown] classifier 1x1x1       {t:8.3f} ms", file=sys.stderr)


if __name__ == "__main__":
    main()
