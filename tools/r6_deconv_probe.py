"""r6: what bounds the split transposed layer with a float32 result?  The same layer with a split (C8) result, and with 64 output channels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from snvc_amd import ops
dev = torch.device("cuda:0")
for name, ci, co, sp in (("64->32 half->full", 64, 32, (96, 48, 156)), ("64->64 quarter->half", 64, 64, (48, 24, 78)), ("64->64 half->full", 64, 64, (96, 48, 156))):
    x = torch.relu(torch.randn(1, ci, *sp, device=dev))
    w = torch.randn(ci, co, 3, 3, 3, device=dev) * 0.05
    lay = ops.Conv3dLayerX3(w, 3, 2, 1, 1, True)
    mul = ops.split_scale_of(x)
    xs = ops.to_split(x, mul_dev=mul)
    t_f32, _ = bench.timed_ms(lambda: lay(xs, 0, None, None, to_f32=True, x_mul_dev=mul), 10, 3)
    out = torch.empty((1, 2, co // 8) + tuple(2 * s for s in sp) + (8,), dtype=torch.float16, device=dev)
    t_c8, _ = bench.timed_ms(lambda: lay(xs, 0, None, None, out=out, out_exp=-8), 10, 3)
    flop = 2 * 27 * ci * co * sp[0] * sp[1] * sp[2]
    print(f"{name}: float32 result {t_f32*1e3:.0f} us | split result {t_c8*1e3:.0f} us | {flop/1e9:.1f} GFLOP, result {co*8*sp[0]*sp[1]*sp[2]*4/1e6:.0f} MB", flush=True)
    del x, xs, out
    torch.cuda.empty_cache()
