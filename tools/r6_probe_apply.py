"""Why are conv6's backward passes slower inside the step than conv2's?  Re-run each act_backward_apply / act_backward_reduce call in place and print pointers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from snvc_amd import ops
from snvc_amd.models import submodule as S
from benchlib.train import TrainStep
dev = torch.device("cuda:0")
ts = TrainStep(0, dev)
for _ in range(3):
    ts()
torch.cuda.synchronize()
orig_apply, orig_reduce = ops.act_backward_apply, ops.act_backward_reduce
def t(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, r
def apply(raw, gy, *a, **k):
    if raw.numel() > 1e8:
        us, r = t(lambda: orig_apply(raw, gy, *a, **k))
        nz = (gy == 0).float().mean().item()
        print(f"apply  {tuple(raw.shape)} flags {a[6]} twin {k.get('twin_mul') is not None}: {us:.0f} us; raw {raw.data_ptr():#x} gy {gy.data_ptr():#x} gy strides {gy.stride()} zeros in gy {nz:.3f} "
              f"finite {torch.isfinite(gy).all().item()} max|gy| {gy.abs().max().item():.3e} tiny(<1e-38) {(gy.abs() < 1.2e-38).float().mean().item():.3f}")
        return r
    return orig_apply(raw, gy, *a, **k)
def reduce(raw, gy, *a, **k):
    if raw.numel() > 1e8:
        us, r = t(lambda: orig_reduce(raw, gy, *a, **k))
        print(f"reduce {tuple(raw.shape)} flags {a[3]}: {us:.0f} us")
        return r
    return orig_reduce(raw, gy, *a, **k)
ops.act_backward_apply, ops.act_backward_reduce = apply, reduce
ts()
torch.cuda.synchronize()
