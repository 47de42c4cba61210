"""BASELINE configs[3] (cfg4): the training step -- cost volume + 3D CNN forward and backward on the HIP kernels + the flat-bucket
gradient all-reduce (RCCL)."""
import gc
import time

import numpy as np
import torch

from .common import C, STEP_FLOP, make_inputs, prewarm, seeded_state

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
        "arithmetic": "f32 contract; conv2 and the hourglass (7 of the 9 layers): forward, data and weight gradients on split f16x3 operands "
                      "(value = hi + lo halves, three f16 MFMAs per product, fp32 accumulate; twins / scales from device-side maxima, r6); first "
                      "layer and classifier: fp32 MFMA; BatchNorm statistics and backward coefficients in fp64",
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
        # r6: the same step with the 3x3x3 layers' forward / data gradients on the fp32 Winograd kernels (r5's route; weight gradients as above)
        S.X3_TRAIN[0] = False
        try:
            tg = TrainStep(rank, device)
            for _ in range(3):
                tg()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            k = max(5, steps // 2)
            for _ in range(k):
                tg()
                torch.cuda.synchronize()
            res["ms_per_step_fp32_forward_dgrad"] = 1e3 * (time.perf_counter() - t1) / k
            del tg
            torch.cuda.empty_cache()
        finally:
            S.X3_TRAIN[0] = True
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
