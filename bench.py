#!/usr/bin/env python3
"""Headline benchmark: stereo-pairs/sec, cost-volume build + 3D CNN forward (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (starts N ranks itself when N > 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N --mode train                    (cfg4: fwd + bwd + RCCL gradient all-reduce)
    python bench.py --gpus N --config cfg3                   (BASELINE configs[2]: 64 RoI crops 96^3 sharded over N ranks)
    python bench.py --headline-only                          (the headline leg + roofline + cpu_baseline, nothing else)

Workload (N=1): BASELINE.json configs[1], "Global scene model: 1242x375, 192 disparities, full 3D hourglass fwd, batch=1 on 1
MI355X", synthesised as SURVEY.md section 8(d) cfg2: left/right features [1,32,96,312] (1242x375 padded to 1248x384, stride 4),
shift = linspace(0, 95.5, 192), downsample 1 -> concat volume [1,64,192,96,312] -> GlobalStack(32) (conv 64->32, conv 32->32,
hourglass(32) + residual, 1x1x1 classifier), eval-mode BatchNorm, random-init weights, fp32.  A step = one pair through
build_cost_volume + the 3D stack, inputs resident in HBM.  Multi-GPU = one process per GPU, each with its own pair (batch
sharding, no data-path collective): weak scaling.

Output (benchlib/emit.py holds the rule): ONE bare-JSON line, the result, LAST on stdout of rank 0 -- compact; everything else
any rank prints is prefixed `[tag] ` (per-rank records on stderr, the long form of the result as `[bench_detail] {...}` and in
gpurun_out/bench_detail.json).  A provisional result line (headline + roofline + cpu_baseline) goes out as soon as those exist, so
that a crash in a later leg still leaves a parsable headline; the final line replaces it by coming later.
  value / ms_per_step     GlobalStack.forward_pair, the contract's timing (W warm-up, K steps barrier to barrier, max over ranks)
  dtype / value_fp32_mfma the arithmetic the step computes in (split mode: fp32 products as three f16 MFMAs) and the strict fp32-MFMA rate
  roofline                dominant kernel (conv2 + side head), HIP events on the launch stream inside the timed loop
  roofline_hbm            the HBM-bound kernels: the first layer's expand pass, the any-shift expand, a1 / a2 builders, gather, a10
  cpu_baseline            CPU oracle (C/OpenMP cost volume + torch-CPU stack) on one full cfg2 pair, rank 0, N=1
  parity_vs_cpu_baseline  every timed leg's output against that oracle's on the same inputs and weights
  configs / train / off_fast_path / sustained   the other BASELINE configs on this GPU, cfg4's step, the slow paths, the >= 5 s rate
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

from benchlib import emit  # noqa: E402
from benchlib.common import *  # noqa: E402,F401,F403  (tests and tools use bench.make_inputs, bench.seeded_state, bench.D, ...)
from benchlib.common import (C, CONV1_FLOP, CV_BYTES, D, H, PEAK_F32_MFMA_TFLOPS, PREWARM_S, W, parity_vs, seeded_state,  # noqa: E402
                             sustained_leg, timed_ms, wino_executed_share, x3_power_probe)
from benchlib.cpu import cpu_baseline, local_inputs, local_oracle  # noqa: E402,F401
from benchlib.launch import init_group, spawn_ranks  # noqa: E402
from benchlib.local import (cfg3_crop_inputs, cfg3_shard_inputs, cfg3_shard_step, local_config, local_model, local_parity,  # noqa: E402,F401
                            off_fast_path, run_cfg3)
from benchlib.train import TrainStep, run_train  # noqa: E402,F401


def parse_args(argv=None):
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
    ap.add_argument("--no-extras", action="store_true", help="skip the `configs` / `train` / `off_fast_path` legs")
    ap.add_argument("--headline-only", action="store_true", help="value + roofline + cpu_baseline only (no other legs)")
    ap.add_argument("--breakdown", action="store_true", help="per-layer timing on stderr (extra untimed pass)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 5 s sustained-rate leg")
    ap.add_argument("--sustained-seconds", type=float, default=5.0)
    ap.add_argument("--sustained-steps", type=int, default=2000)
    ap.add_argument("--rehearse", action="store_true",
                    help="launch / timing-protocol / output rehearsal on the CPU (gloo group, a sleep as the step): NOT a measurement")
    return ap.parse_args(argv)


def rehearse(args, rank, world):
    """The launcher, the group, the contract's timing protocol and the emission rule end to end WITHOUT a GPU and without any
    kernel: the step is a 1 ms sleep.  For tests/test_bench_emit.py (world 2 on gloo); the line says what it is."""
    dist, joined, note = init_group("gloo", rank, world) if world > 1 else (None, 1, "no group")
    fail_rank = int(os.environ.get("SNVC_REHEARSE_FAIL_RANK", "-1"))

    def barrier():
        if dist is not None:
            dist.barrier()
    for _ in range(args.warmup):
        time.sleep(1e-3)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    mine = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    if rank == fail_rank:
        emit.rank_note("rank_record", {"rank": rank, "failing": True})
        raise SystemExit(3)
    recs = [{"rank": rank, "ms_per_step_this_rank": 1e3 * mine / args.steps}]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        vec = torch.tensor([float(rank), 1e3 * mine / args.steps], dtype=torch.float64)
        got = [torch.empty_like(vec) for _ in range(world)]
        dist.all_gather(got, vec)
        recs = [{"rank": int(v[0]), "ms_per_step_this_rank": float(v[1])} for v in got]
    emit.rank_note("rank_record", recs[rank if dist is not None else 0])
    if rank == 0:
        line = {"metric": "REHEARSAL (no kernel ran: a 1 ms sleep as the step)", "value": world * args.steps / elapsed, "unit": "sleeps/s",
                "n_gpus": joined, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                "config": {"workload": "rehearsal of bench.py's launcher / timing protocol / output rule on the CPU", "group": note},
                "ranks": recs, "not_finite_example": float("nan")}
        emit.detail_note(line)
        emit.emit_result(line, required=emit.CONTRACT)
    if dist is not None:
        dist.destroy_process_group()
    return 0


def _breakdown(model, left, right, shift):
    from snvc_amd.extension.build_cost_volume import build_cost_volume
    rows = {}
    with torch.no_grad():
        t, vol = timed_ms(lambda: build_cost_volume(left, right, shift, 1).materialize(), 5)
        rows["build_cost_volume"] = {"ms": t, "GBps": CV_BYTES / (t * 1e-3) / 1e9}
        t, v1 = timed_ms(lambda: model.conv1(vol), 5)
        rows["conv1 k3 64->32"] = {"ms": t, "TFLOPs": CONV1_FLOP / (t * 1e-3) / 1e12}
        del vol
        t, v2 = timed_ms(lambda: model.conv2(v1), 5)
        rows["conv2 k3 32->32"] = {"ms": t, "TFLOPs": CONV1_FLOP / 2 / (t * 1e-3) / 1e12}
        t, _ = timed_ms(lambda: model.hg_conv3d(v2, None, None, residual=v2), 5)
        rows["hourglass(32)"] = {"ms": t, "TFLOPs": 377.6e9 / (t * 1e-3) / 1e12}
        t, _ = timed_ms(lambda: model.classifier(v2), 5)
        rows["classifier 1x1x1"] = {"ms": t}
    emit.rank_note("breakdown (module by module, fp32-MFMA kernels, materialised volume)", rows)


def run_extras(args, line, rank, world, device, dist, barrier, rccl_note):
    """The legs beside the headline: cfg4's step (every N: the one leg with a collective), and at N = 1 the other BASELINE configs,
    the slow paths and the SURVEY 8(a) HBM rows the headline step does not launch.  An extra never takes the headline down."""
    from benchlib import hbm_rows
    from benchlib import local as L_
    tr = run_train(rank, world, device, dist, 10, 3, barrier)
    if rank == 0:
        line["train"] = tr
        if rccl_note:
            line["train"]["rccl"] = rccl_note
    if world != 1:
        return
    cfgs = {}
    for name, grid, F, crops, heads, prec in (("cfg3_crops_96", (96, 96, 96), 32, 8, False, "f32"),
                                              ("released_32x128x192", (32, 128, 192), 32, 2, True, "f32"),
                                              ("released_32x128x192_f16", (32, 128, 192), 32, 2, True, "f16"),
                                              ("cfg5_highres_80x160x160", (80, 160, 160), 64, 1, False, "f16"),
                                              ("cfg5_highres_80x160x160_f32", (80, 160, 160), 64, 1, False, "f32")):
        try:
            cfgs[name] = local_config(name, grid, F, crops, device, heads=heads, precision=prec)
        except Exception as e:
            cfgs[name] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and "error" not in cfgs[name]:
            try:        # the same model on the oracle's inputs: whole-tensor parity + the CPU rate beside the GPU rate
                cfgs[name].update(local_parity(grid, F, device, prec, sample_grid=(48, 80, 80) if grid == (80, 160, 160) else None))
            except Exception as e:
                cfgs[name]["parity_vs_cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    L_._ORACLES.clear()
    try:        # BASELINE configs[2] as the N > 1 runs shard it (`--config cfg3`), here all 64 crops on one rank
        cfgs["cfg3_64crops_sharded"] = run_cfg3(rank, world, device, dist, 2, 1, barrier)
    except Exception as e:
        cfgs["cfg3_64crops_sharded"] = {"error": f"{type(e).__name__}: {e}"}
    cfgs["cfg4_train_step"] = {k: tr[k] for k in ("ms_per_step", "fwd_ms", "bwd_ms", "step_tflops_algorithmic") if k in tr}
    try:        # the weight gradient of a 32->32 layer on the full grid (conv2's: the largest of the step's wgrad launches), both forms
        from snvc_amd import _lib, ops
        from benchlib.common import PEAK_F16_MFMA_TFLOPS
        xg = torch.relu(torch.randn(1, C, D, H, W, device=device))
        gg = torch.randn(1, C, D, H, W, device=device) * 1e-4
        flop = CONV1_FLOP / 2                          # 32 of conv1's 64 input channels
        ax, ag = ops.amax_word(device), ops.amax_word(device)      # the maxima as the training step has them: left by the producer passes
        ax[0:1] = xg.abs().max().reshape(1).view(torch.int32)
        ag[0:1] = gg.abs().max().reshape(1).view(torch.int32)
        ms_x3, _ = timed_ms(lambda: ops.conv3d_wgrad(xg, gg, 3, 1, 1, 1, amax_x=ax, amax_g=ag), 5, 3)
        ms_x3_own, _ = timed_ms(lambda: ops.conv3d_wgrad(xg, gg, 3, 1, 1, 1), 5, 3)
        with ops.conv_variant(_lib.ALGO_WGRAD_FP32):
            ms_w, _ = timed_ms(lambda: ops.conv3d_wgrad(xg, gg, 3, 1, 1, 1), 3)
        cfgs["cfg4_train_step"].update({
            "wgrad_kernel": "conv3d_wgrad_x3_kernel 32->32 on 192x96x312 (r6: operands split into (hi, lo) halves on the way into LDS, three "
                            "v_mfma_f32_16x16x32_f16 per fp32 product, deterministic) + wgrad_reduce_x3_kernel",
            "wgrad_ms": ms_x3, "wgrad_ms_finding_its_own_maxima": ms_x3_own, "wgrad_gflop_algorithmic": flop / 1e9,
            # necessary matrix-pipe flops: three half-precision MFMA flops per algorithmic multiply-add, against the f16 peak
            "wgrad_pipe_frac": 3.0 * flop / (ms_x3 * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS,
            "wgrad_fp32_form_ms": ms_w,
            "wgrad_fp32_form_pipe_frac": wino_executed_share(3, W) * flop / (ms_w * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS})
        del xg, gg
    except Exception as e:
        cfgs["cfg4_train_step"]["wgrad_error"] = f"{type(e).__name__}: {e}"
    line["configs"] = cfgs
    try:
        line["off_fast_path"] = off_fast_path(device)
    except Exception as e:
        line["off_fast_path"] = {"error": f"{type(e).__name__}: {e}"}
    g3 = cfgs.get("cfg3_crops_96", {})
    if g3.get("gather_projected"):      # north_star's ">= 60 % HBM roofline on the warp/gather": the gather beside the cost-volume builders
        for key, what in (("gather_projected", "GridProjector coordinates"), ("gather_uniform", "SURVEY 8(d)'s uniform coordinates")):
            line["roofline_hbm"]["gather" if key == "gather_projected" else key] = {
                "kernel": f"voxel_gather_fwd_lds (a3): _sample_2d_feat on 8 crops 96^3, F=32, {what}",
                "achieved": g3[key]["GBps"], "frac": g3[key]["frac_hbm"], "avg_launch_ms": g3[key]["ms"],
                "bytes_per_launch": g3["gather_bytes_algorithmic"]}
    for key, fn in (("cost_volume_backward", hbm_rows.cost_volume_backward_row), ("roiaware_pool3d", hbm_rows.roiaware_row)):
        try:
            line["roofline_hbm"][key] = fn(device)
        except Exception as e:
            line["roofline_hbm"][key] = {"error": f"{type(e).__name__}: {e}"}


def run_sustained(args, line, rank, world, device, dist):
    """the headline step again, back to back for >= 5 s and >= 2000 steps, AFTER the extras (a warm part): the sustained rate"""
    from snvc_amd.models.stereo_volume import GlobalStack
    from benchlib.common import make_inputs
    model = GlobalStack(C)
    model.load_state_dict(seeded_state(model))
    model.eval().to(device)
    left, right, shift = make_inputs(rank, device)
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
    emit.rank_note("rank_record_sustained", dict(sus, rank=rank))
    if rank == 0:
        sus["note"] = (f"`value` is the driver's K-step window after a {PREWARM_S} s pre-warm; this is the same step for >= 5 s on a part "
                       "already warm from the other legs -- what a deployment running the step continuously sees")
        sus["vs_value"] = sus["pairs_per_s"] / line["value"]
        line["sustained"] = sus


def main(argv=None):
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), args.gpus,
                                     need_gpus=not args.rehearse))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse:
        raise SystemExit(rehearse(args, rank, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist, joined, rccl_note = init_group("nccl", rank, world, device)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def finish():
        if dist is not None:
            dist.destroy_process_group()

    if args.mode == "train":
        tr = run_train(rank, world, device, dist, args.steps, args.warmup, barrier)
        if rank == 0:
            line = {"metric": "stereo-pairs/sec (training step: cost-volume + 3D CNN fwd+bwd + gradient all-reduce)",
                    "value": tr["pairs_per_s"], "unit": "stereo-pairs/s", "n_gpus": joined, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": tr["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": tr.get("arithmetic", "f32"), "data": "synthetic",
                    "config": {"workload": tr["workload"], "sharding": f"batch x{world}; RCCL all-reduce of {tr['allreduce_bytes']} "
                                                                       "gradient bytes per step"},
                    "train": tr}
            emit.detail_note(line, ROOT)
            emit.emit_result(line, required=emit.CONTRACT)
        return finish()

    if args.config == "cfg3":
        c3 = run_cfg3(rank, world, device, dist, args.steps, args.warmup, barrier, args.crops, args.crops_per_call, not args.no_gather)
        if rank == 0:
            line = {"metric": "RoI-crops/sec (feature->voxel gather + 3D trunk fwd, 96^3 crops)", "value": c3["crops_per_s"],
                    "unit": "RoI-crops/s", "n_gpus": joined, "steps": args.steps, "warmup": args.warmup, "ms_per_step": c3["ms_per_step"],
                    "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": c3["workload"], "sharding": f"{args.crops} crops on dim 0 over {world} rank(s), no data-path "
                                                                       "collective; " + c3["outputs_gathered"]},
                    "cfg3": c3}
            emit.detail_note(line, ROOT)
            emit.emit_result(line, required=emit.CONTRACT)
        return finish()

    # ---- the headline: BASELINE configs[1] through GlobalStack.forward_pair
    from benchlib.headline import Headline
    hd = Headline(args, rank, world, device, dist, barrier)
    hd.run_core()
    records = hd.all_rank_records(joined)
    emit.rank_note("rank_record", hd.rank_record(joined))         # every rank leaves its own line on stderr (prefixed: not JSON)
    line = hd.core_line(joined) if rank == 0 else None
    need = emit.CONTRACT + ("roofline",)
    oracle_out = {}
    if rank == 0:
        line["ranks"] = records
        if world == 1 and not args.no_cpu_baseline:
            need = emit.REQUIRED
            line["cpu_baseline"] = cpu_baseline(outputs=oracle_out)
            # the oracle ran on exactly the timed legs' inputs and weights: the timed leg's last output against it, all 5.75 M values
            line["parity_vs_cpu_baseline"] = dict(parity_vs(hd.outs["value"], oracle_out["cost"]),
                                                  tolerance="north_star: 1e-3 relative fp32; tests/test_gpu_fullsize_oracle.py asserts "
                                                            "rel_err <= 1e-4 and the elementwise rule")
        emit.emit_result(line, provisional=True, required=need)      # a crash in a later leg still leaves this one

    if not args.headline_only:
        hd.run_more()
        if rank == 0:
            hd.add_more(line)
            if oracle_out:
                par = {k: parity_vs(v, oracle_out["cost"]) for k, v in hd.outs.items() if v is not None}
                line["parity_vs_cpu_baseline"]["legs"] = par
    if args.breakdown and rank == 0:
        _breakdown(hd.model, hd.left, hd.right, hd.shift)
    x3_taken = hd.x3_taken
    hd.release()
    if not args.headline_only and not args.no_extras:
        if x3_taken and rank == 0:
            try:
                line["roofline"]["power_probe"] = x3_power_probe(device)
            except Exception as e:
                line["roofline"]["power_probe"] = {"error": f"{type(e).__name__}: {e}"}
        run_extras(args, line, rank, world, device, dist, barrier, rccl_note)
    if not args.headline_only and not args.no_sustained:
        run_sustained(args, line, rank, world, device, dist)
    if rank == 0:
        emit.detail_note(line, ROOT)
        emit.emit_result(line, required=need)      # LAST: nothing is printed after this line by this process
    finish()


if __name__ == "__main__":
    main()
