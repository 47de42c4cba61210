"""The headline of bench.py: BASELINE configs[1] (cfg2) -- one stereo pair through cost-volume build + 3D CNN forward -- its legs
(the same step on the other entry points / arithmetic), and the assembly of the result line from their measurements."""
import gc
import json
import os
import time

import numpy as np
import torch

from .common import (C, CONV1_FLOP, CV_BYTES, CV_RIGHT_BYTES, D, H, PEAK_F16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, PREWARM_S,
                     ROOT, STEP_FLOP, W, X3Q_SOURCE, kernel_source_hash, make_inputs, prewarm, read_gpu_clock_mhz, seeded_state,
                     wino_executed_share)

V1_BYTES = 4.0 * C * D * H * W                  # the first layer's output, written once by the expand pass
DOM_FLOP = CONV1_FLOP / 2                       # a 32 -> 32 channel 3x3x3 layer on the full grid (conv2; conv1's right half)
DTYPE_SPLIT = "f32 (split f16x3 operands, fp32 accumulate)"


def _div(a, b):
    return None if (a is None or b is None or not b) else a / b


def _gbs(nbytes, ms):
    return None if not ms else nbytes / (ms * 1e-3) / 1e9


def _tflops(flop, ms):
    return None if not ms else flop / (ms * 1e-3) / 1e12


class Headline:
    """Owns the cfg2 model and inputs of this rank and runs the timed legs.  ``leg`` is the contract's timing protocol: W warm-up
    steps, then exactly K steps bracketed by a barrier + synchronize on both sides, the MAX over ranks."""

    def __init__(self, args, rank, world, device, dist, barrier):
        from snvc_amd.models.stereo_volume import GlobalStack
        self.args, self.rank, self.world, self.device, self.dist, self.barrier = args, rank, world, device, dist, barrier
        self.model = GlobalStack(C)
        self.model.load_state_dict(seeded_state(self.model))
        self.model.eval().to(device)
        self.left, self.right, self.shift = make_inputs(rank, device)
        self.outs, self.local_elapsed, self.legs = {}, {}, {}

    # ------------------------------------------------------------------------------------------ timing
    def _timed(self, step, tag, events=None):
        """-> seconds for K steps, max over ranks.  ``step(i)`` runs step i (and records its event brackets)."""
        a = self.args
        with torch.no_grad():
            # the cyclic garbage collector stays out of the timed region (as timeit does): a generation-2 pass over the ~1e6 objects
            # torch keeps alive costs ~35 ms, i.e. 17 steps, whenever its counter happens to trip.  It is run BEFORE the warm-up: a
            # 50 ms host pause between the warm-up and the timed steps lets the GPU fall out of its sustained clocks again
            gc.collect()
            gc.disable()
            try:
                prewarm(lambda: step(None))
                for _ in range(a.warmup):
                    step(None)
                self.barrier()
                t0 = time.perf_counter()
                for i in range(a.steps):
                    out = step(i)
                torch.cuda.synchronize()
                self.local_elapsed[tag] = time.perf_counter() - t0     # this rank's own K steps (before it waits for the others)
                self.barrier()
                elapsed = time.perf_counter() - t0
            finally:
                gc.enable()
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert torch.isfinite(out).all()
        self.outs[tag] = out.cpu().numpy() if self.rank == 0 and self.world == 1 else None    # 23 MB, compared with the CPU oracle
        return elapsed

    def leg(self, tag, factored=True, sheared=True, commuted=True, arithmetic=None, brackets=("volume", "conv1", "conv2")):
        """One leg through GlobalStack.forward_pair.  Returns {"s": seconds for K steps, "<bracket>": mean ms or None}.
        sheared path: volume = Rq + the 2D convolution G + the edge slab, conv1 = the expand pass; general path: volume = the
        right-half cost-volume launch, conv1 = the first 3D convolution; conv2 = the second 3D convolution (+ side head)."""
        a, m = self.args, self.model
        # events go to torch's current stream == the stream the kernels are launched on.  The headline leg records the dominant
        # kernel's bracket only: three brackets (six event records per step) cost 0.66 % of the step, one costs nothing
        ev = [{k: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for k in brackets} for _ in range(a.steps)]

        def step(i):
            return m.forward_pair(self.left, self.right, self.shift, 1, factored=factored, timing=None if i is None else ev[i],
                                  sheared=sheared, commuted=commuted, arithmetic=arithmetic)
        res = {"s": self._timed(step, tag)}
        for k in ("volume", "conv1", "conv2"):
            try:
                res[k] = float(np.mean([e[k][0].elapsed_time(e[k][1]) for e in ev]))
            except (RuntimeError, ValueError, KeyError):      # a bracket this path / this leg does not record
                res[k] = None
        self.legs[tag] = res
        return res

    def reference_api_leg(self, tag="reference_api"):
        """the reference's call sequence, verbatim: volume = build_cost_volume(l, r, s, 1); cost = model(volume)"""
        from snvc_amd.extension.build_cost_volume import build_cost_volume
        res = {"s": self._timed(lambda i: self.model(build_cost_volume(self.left, self.right, self.shift, 1)), tag)}
        self.legs[tag] = res
        return res

    # ------------------------------------------------------------------------------------------ the legs
    def run_core(self):
        """`value` (+ which routes it took) and the two legs the roofline entries need."""
        from snvc_amd.models import submodule as S_
        r0, r1 = S_._ROUTES["sheared_first_conv"], S_._ROUTES["x3_tail"]
        self.leg("value", brackets=("conv2",))
        self.sheared_taken = S_._ROUTES["sheared_first_conv"] > r0
        self.x3_taken = S_._ROUTES["x3_tail"] > r1          # conv2 + hourglass on the split-mode (f16x3) kernels
        self.leg("first_layer_brackets")                     # the first layer's own brackets (prep chains, expand pass) for `roofline_hbm`
        # the same step with conv2 and the hourglass on the fp32-MFMA kernels (r1-r3's arithmetic: Winograd F(4,3), v_mfma_f32_32x32x2_f32)
        self.leg("fp32_mfma", arithmetic="fp32")
        st = self.model.__dict__.get("_snvc_x3")
        self.x3_overflow = int(st["flag"].item()) if st is not None else None       # 0: no value was clamped to half's range
        self.x3_exponents = dict(st["exp"]) if st is not None else None

    def run_more(self):
        """The same step on the other entry points (all N; cheap: 20 steps each)."""
        from snvc_amd.models import submodule as S_
        m = self.model
        m.overflow_check = "deferred"          # what reading the split-mode overflow flag INSIDE the call costs: the flag only posted
        self.leg("deferred_overflow_check", brackets=("conv2",))
        m.check_overflow()
        m.overflow_check = "call"
        m.fused_tail = False                   # r4's tail (conv5 -> fp32 `post` -> the one-channel transposed layer as its own kernel)
        self.leg("two_launch_tail")
        m.fused_tail = True
        self.leg("general_shift", sheared=False)                        # any shift array: warp after convolution
        self.leg("built_right_half", sheared=False, commuted=False)     # right half built + 3D convolution over it
        self.leg("materialized", factored=False)
        self.reference_api_leg()
        self.redone = int(S_._ROUTES["x3_overflow_redo"])
        self.lazy_stale = int(S_._ROUTES["lazy_prefetch_stale"])

    def release(self):
        self.model = None
        torch.cuda.empty_cache()

    def rank_record(self, joined):
        a = self.args
        return {"rank": self.rank, "device": torch.cuda.get_device_name(self.device), "world_joined": joined, "steps": a.steps,
                "ms_per_step_this_rank": 1e3 * self.local_elapsed["value"] / a.steps,
                "ms_per_step_max_over_ranks": 1e3 * self.legs["value"]["s"] / a.steps,
                "split_mode": bool(self.x3_taken), "sclk_mhz": read_gpu_clock_mhz(self.device.index or 0)}

    def all_rank_records(self, joined):
        """Every rank's record on rank 0 (a fixed-size float vector per rank through the group the run already has: no pickling)."""
        mine = self.rank_record(joined)
        if self.dist is None or self.world == 1:
            return [mine]
        vec = torch.tensor([mine["rank"], mine["ms_per_step_this_rank"], mine["ms_per_step_max_over_ranks"], float(mine["split_mode"]),
                            mine["sclk_mhz"] if mine["sclk_mhz"] is not None else -1.0], dtype=torch.float64, device=self.device)
        got = [torch.empty_like(vec) for _ in range(self.world)]
        self.dist.all_gather(got, vec)
        return [{"rank": int(v[0]), "ms_per_step_this_rank": float(v[1]), "ms_per_step_max_over_ranks": float(v[2]),
                 "split_mode": bool(v[3]), "sclk_mhz": None if v[4] < 0 else float(v[4])} for v in (g.cpu() for g in got)]

    # ------------------------------------------------------------------------------------------ the line
    def _traffic(self):
        """HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process; separate rocprofv3 --pmc
        passes (FETCH_SIZE, WRITE_SIZE, gfx950 correction) are committed under profiles/.  The counters belong to the kernel text they
        were collected on: a changed kernel voids them."""
        rounds = ("r6", "r5", "r4", "r3", "r2", "r1")
        if self.x3_taken:
            for r in rounds:
                rel = f"profiles/{r}/traffic.json"
                try:
                    with open(os.path.join(ROOT, rel)) as fh:
                        ent = json.load(fh).get("layers", {}).get("x3_conv2", {})
                except Exception:
                    continue
                if ent.get("hbm_bytes_corrected") is None:
                    continue
                then, now = ent.get("kernel_source_sha256_16"), kernel_source_hash(*X3Q_SOURCE)
                if then is None or then != now:
                    return None, (f"{rel}: collected on kernel source {then}, the kernel is now {now}: re-run the --pmc passes "
                                  "(tools/pmc_traffic.sh + tools/make_traffic_json.py)")
                return ent["hbm_bytes_corrected"], (f"{rel}, layer x3_conv2 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                                    "kernel source hash checked)")
            return None, "no traffic.json with an x3_conv2 entry"
        key = ("layers", "conv2_side") if self.sheared_taken else ("conv1_right_wino43_dma_k3_32to32_cfg2",)
        for r in rounds:
            rel = f"profiles/{r}/traffic.json"
            try:
                with open(os.path.join(ROOT, rel)) as fh:
                    ent = json.load(fh)
                for k in key:
                    ent = ent.get(k, {})
            except Exception:
                continue
            if ent.get("hbm_bytes_corrected") is not None:
                return ent["hbm_bytes_corrected"], rel + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        return None, None

    def core_line(self, joined, power_probe=None):
        """The result line from the core legs alone (metric, value, roofline, roofline_hbm, fp32_mfma)."""
        a, L, world = self.args, self.legs, self.world
        x3, sheared = self.x3_taken, self.sheared_taken
        elapsed = L["value"]["s"]
        share = wino_executed_share(3, W)               # F(4,3): 6 of 12 multiplies x padding of W=312 to 320
        # split mode: three half-precision MFMAs per product, 28 tap slots for 27 taps (two taps per MFMA), W = 312 on 32-wide tiles
        share_x3 = 3.0 * (28.0 / 27.0) * ((-(-W // 32) * 32) / float(W))
        dom_ms = L["value"]["conv2"]
        # what `frac` prices: the flops the arithmetic NEEDS on the pipe it runs on -- split mode: three half-precision MFMA flops per
        # fp32 product (no padding); fp32 Winograd form: the algorithm's 6 of 12 multiplies -- not what the tiling pads on top
        need = _tflops(DOM_FLOP * (3.0 if x3 else 0.5), dom_ms)
        executed = _tflops(DOM_FLOP * (share_x3 if x3 else share), dom_ms)
        alg = _tflops(DOM_FLOP, dom_ms)
        peak = PEAK_F16_MFMA_TFLOPS if x3 else PEAK_F32_MFMA_TFLOPS
        f32_ms = L["fp32_mfma"]["conv2"]
        exec_f32 = _tflops(DOM_FLOP * share, f32_ms)
        traffic, traffic_src = self._traffic()
        fl = L["first_layer_brackets"]
        line = {
            "metric": "stereo-pairs/sec (cost-volume build + 3D CNN fwd)",
            "value": world * a.steps / elapsed,
            "unit": "stereo-pairs/s",
            "n_gpus": joined,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # the arithmetic type the path computes in: fp32 tensors in and out; in split mode every fp32 product is three f16 MFMAs on
            # (hi, lo) half pairs with fp32 accumulation (22 significand bits).  `value_fp32_mfma` is the strict fp32-MFMA figure.
            "dtype": DTYPE_SPLIT if x3 else "f32",
            "value_fp32_mfma": world * a.steps / L["fp32_mfma"]["s"],
            "data": "synthetic",
            "config": {
                "workload": "cfg2 global scene model: 1 pair/GPU, features [1,32,96,312] (1242x375 /4), 192 disparities -> concat volume "
                            "[1,64,192,96,312] -> conv3d x2 + hourglass(32) + classifier",
                "arithmetic": ("fp32 tensors in and out; conv2 + hourglass in SPLIT MODE (" + ("taken" if x3 else "NOT taken") + "): activations / "
                               "weights travel as (hi, lo) pairs of halves (22 significant bits), each fp32 product = three half-precision MFMAs "
                               "with fp32 accumulation; `value_fp32_mfma` / `fp32_mfma` repeat the step on the fp32-MFMA kernels"),
                "arithmetic_note": ("split mode holds the fp32 layers to fp32 accuracy (5e-7 of the range vs float64 per layer; the fp32 Winograd "
                                    "kernels: 2e-6) and to the SAME per-layer 2e-5 / stack 1e-4 tolerances as the fp32 kernels "
                                    "(tests/test_gpu_fullsize_oracle.py, parity_vs_cpu_baseline below)"),
                "prewarm_s": PREWARM_S,
                "prewarm": (f"{PREWARM_S} s of the same step, untimed, in front of every leg's W warm-up steps: after an idle stretch the GPU "
                            "needs a few hundred ms of load to reach its sustained clocks.  `value` is meant to be the sustained rate: hold it "
                            "against `sustained` (>= 5 s, >= 2000 steps)"),
                "split_mode": {"taken": bool(x3), "overflow_flag": self.x3_overflow, "tensor_exponents": self.x3_exponents,
                               "rule": "2^e * (|beta| + 64 |gamma|) <= 2^15 per tensor (folded eval BatchNorm); a value beyond it is clamped "
                                       "and flagged, the call is then redone on the fp32-MFMA kernels before anything is returned"},
                "sheared_first_layer": bool(sheared),
                "entry_points": {
                    "value": "GlobalStack.forward_pair(left, right, shift): fused entry point.  Left half of the concat volume: d-invariant -> 3 "
                             "depth-class planes.  Warped right half: the disparity planes are uniformly spaced (shift = d/2), so it is a shear "
                             "of one 2D image and conv1 over it is a 2D convolution evaluated along the shear (csrc/sheared_conv.hip)",
                    "general_shift": "the same entry point for ANY shift array (sheared=False): interpolation along w commutes with the "
                                     "convolution -- three 2D convolutions of the right feature + three interpolations per output voxel",
                    "built_right_half": "forward_pair(..., sheared=False, commuted=False): right half of the volume built, factored first 3D "
                                        "convolution over it (r2's path)",
                    "reference_api": "model(build_cost_volume(left, right, shift, 1)): same kernels as `value` (lazy volume)",
                    "materialized": "the full concat volume built in HBM, then the modules (conv1 over all 64 channels)"},
                "pairs_per_gpu_per_step": 1,
                "sharding": f"batch x{world}, no collective",
                "step_gflop_algorithmic": STEP_FLOP / 1e9,
                "step_cost_volume_mb_algorithmic": CV_BYTES / 1e6,
            },
            "roofline": {
                "kernel": ("conv3d_x3q_kernel<side head>: conv2 32->32 on 192x96x312 + classifier side head, split mode (f16x3), 4x4x32 tile, "
                           "2 WG/CU, 3 x v_mfma_f32_16x16x32_f16 per fp32 product (csrc/conv3d_f16.hip)" if x3 else
                           "conv3d_wino_dma_kernel<4x4x32, KC2, 3 WG/CU, side head>: conv2 32->32 on 192x96x312 (Winograd F(4,3), fp32 MFMA)"
                           if sheared else
                           "conv3d_wino_dma_kernel<4x4x32, KC2, 3 WG/CU, planes>: conv1 over the right half, 32->32 on 192x96x312"),
                "bound": "mfma",
                # `achieved` = the matrix-pipe flops the layer's arithmetic NEEDS per second (see `need` above); `executed_tflops` adds
                # what the tiling pads on top (+6.3 %, = SQ_INSTS_MFMA x 16384 flop); `algorithmic_tflops` = 2*voxels*Cin*Cout*27 / time
                "achieved": need, "peak": peak, "unit": "TFLOP/s", "frac": _div(need, peak),
                "executed_tflops": executed, "executed_frac": _div(executed, peak),
                "algorithmic_tflops": alg, "algorithmic_over_fp32_mfma_peak": _div(alg, PEAK_F32_MFMA_TFLOPS),
                "flop_per_launch_algorithmic": DOM_FLOP, "flop_per_launch_executed": DOM_FLOP * (share_x3 if x3 else share),
                "avg_launch_ms": dom_ms,
                "traffic": traffic, "traffic_source": traffic_src,
                "power_probe": power_probe,
                "fp32_mfma_form": {"kernel": "conv3d_wino_dma_kernel<4x4x32, side head> (Winograd F(4,3), v_mfma_f32_32x32x2_f32)",
                                   "avg_launch_ms": f32_ms, "achieved": exec_f32, "peak": PEAK_F32_MFMA_TFLOPS,
                                   "frac": _div(exec_f32, PEAK_F32_MFMA_TFLOPS)},
            },
            "roofline_hbm": {
                # the headline path's own HBM-bound kernel: conv1's result written along the shear (one 0.74 GB write stream)
                "kernel": "sheared_expand_split_kernel: the first layer's output of the headline path, written along the shear",
                "bound": "hbm", "achieved": _gbs(V1_BYTES, fl["conv1"]), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": _div(_gbs(V1_BYTES, fl["conv1"]), PEAK_HBM_GBS), "bytes_per_launch": V1_BYTES,
                "avg_launch_ms": fl["conv1"], "prep_ms": fl["volume"],
                "measured_in": "a repeat of the headline leg with the first layer's event brackets on (`value` itself records the conv2 "
                               "bracket only: three brackets cost 0.66 % of the step)",
                "prep": "Rq on two grids + the depth-1 3x7 convolutions G (all columns) and G' (last column), 3 depth classes each",
            },
            "fp32_mfma": {
                "note": "same step, same entry point, with conv2 and the hourglass on the fp32-MFMA kernels (forward_pair(..., "
                        "arithmetic='fp32'): Winograd F(4,3) / polyphase kernels on v_mfma_f32_32x32x2_f32 -- rounds 1-3's arithmetic)",
                "value": world * a.steps / L["fp32_mfma"]["s"], "ms_per_step": 1e3 * L["fp32_mfma"]["s"] / a.steps,
                "conv2_ms": f32_ms, "expand_ms": L["fp32_mfma"]["conv1"],
            },
            "step_tflops_algorithmic": STEP_FLOP / (elapsed / a.steps) / 1e12,
        }
        return line

    def add_more(self, line):
        """The entries of ``run_more``'s legs."""
        a, L, world = self.args, self.legs, self.world
        x3 = self.x3_taken
        share = wino_executed_share(3, W)
        share_x3 = 3.0 * (28.0 / 27.0) * ((-(-W // 32) * 32) / float(W))

        def rate(tag):
            return world * a.steps / L[tag]["s"]

        def ms(tag):
            return 1e3 * L[tag]["s"] / a.steps
        gen, built, mat = L["general_shift"], L["built_right_half"], L["materialized"]
        line["roofline_hbm"].update({
            "warped_expand": {"kernel": "warped_expand_split_kernel: the same layer for ANY shift array (general_shift leg), same 0.74 GB write",
                              "achieved": _gbs(V1_BYTES, gen["conv1"]), "frac": _div(_gbs(V1_BYTES, gen["conv1"]), PEAK_HBM_GBS),
                              "bytes_per_launch": V1_BYTES, "avg_launch_ms": gen["conv1"], "prep_ms": gen["volume"]},
            "right_half_builder": {"kernel": "cost_volume_fwd_rows: right (warped) half only (built_right_half leg; on no default path)",
                                   "achieved": _gbs(CV_RIGHT_BYTES, built["volume"]),
                                   "frac": _div(_gbs(CV_RIGHT_BYTES, built["volume"]), PEAK_HBM_GBS),
                                   "bytes_per_launch": CV_RIGHT_BYTES, "avg_launch_ms": built["volume"]},
            "full_volume": {"kernel": "cost_volume_fwd_rows: build_cost_volume, both halves (a1; materialized leg)",
                            "achieved": _gbs(CV_BYTES, mat["volume"]), "frac": _div(_gbs(CV_BYTES, mat["volume"]), PEAK_HBM_GBS),
                            "bytes_per_launch": CV_BYTES, "avg_launch_ms": mat["volume"]}})
        alg_mat = _tflops(CONV1_FLOP, mat["conv1"])
        line.update({
            "overflow_check": {
                "note": "split mode clamps a value beyond the range its BatchNorm parameters promise and raises a device flag.  `value` reads "
                        "that flag INSIDE the call (4-byte copy queued behind the last layer that can clamp, waited for after the rest of "
                        "the call is queued) and redoes a flagged call on the fp32-MFMA kernels: no clamped result is ever returned "
                        "(tests/test_gpu_overflow.py).  `deferred` = the same leg with the flag only posted (r4's behaviour)",
                "checked_ms_per_step": ms("value"), "deferred_ms_per_step": ms("deferred_overflow_check"),
                "cost_ms_per_step": ms("value") - ms("deferred_overflow_check"), "redone_calls": self.redone},
            "two_launch_tail": {
                "note": "same step with r4's tail: conv5 writes `post` (64 channels, fp32), deconv3d_cout1_kernel reads it; `value` contracts "
                        "conv5's result with the folded tail's 27 taps in conv5's epilogue + snvc_deconv_tail_gather",
                "value": rate("two_launch_tail"), "ms_per_step": ms("two_launch_tail")},
            "reference_api": {
                "note": "the reference's call sequence verbatim -- volume = build_cost_volume(left, right, shift, 1); cost = model(volume) -- "
                        "under torch.no_grad(): build_cost_volume returns a LazyCostVolume (snvc_amd/lazy.py) that GlobalStack.forward "
                        "consumes on the fused path; any other use of it builds the real volume",
                "value": rate("reference_api"), "ms_per_step": ms("reference_api"), "stale_prefetches": self.lazy_stale},
            "general_shift": {
                "note": "same step on the path any shift array takes (forward_pair(..., sheared=False)): warp after convolution",
                "value": rate("general_shift"), "ms_per_step": ms("general_shift"), "expand_ms": gen["conv1"], "prep_ms": gen["volume"]},
            "built_right_half": {
                "note": "same step with the right half of the volume built (cost_volume_fwd_rows) and the factored first 3D convolution over it",
                "value": rate("built_right_half"), "ms_per_step": ms("built_right_half"), "conv1_ms": built["conv1"],
                "conv1_pipe_frac": _div(_tflops(DOM_FLOP * share, built["conv1"]), PEAK_F32_MFMA_TFLOPS), "volume_ms": built["volume"]},
            "materialized": {
                "note": "same step with the full concat volume built in HBM (what the reference does, and what this library does whenever "
                        "the volume is written to): build_cost_volume_forward, conv1 runs over all 64 channels",
                "value": rate("materialized"), "ms_per_step": ms("materialized"), "conv1_ms": mat["conv1"],
                "conv1_tflops_algorithmic": alg_mat,
                # split mode: three half-precision MFMAs per product against the f16 peak; fp32 form: Winograd's 6/12 against the fp32 peak
                "conv1_pipe_frac": (_div(alg_mat, PEAK_F16_MFMA_TFLOPS / share_x3) if x3 else _div(alg_mat, PEAK_F32_MFMA_TFLOPS / share)),
                "conv1_kernel": ("conv3d_f16_kernel<k3, split mode> 64->32 after a layout pass of the 1.47 GB volume" if x3 else
                                 "conv3d_wino_dma_kernel 64->32 (fp32 Winograd F(4,3))"),
                "conv1_flop_per_launch": CONV1_FLOP}})
        return line
