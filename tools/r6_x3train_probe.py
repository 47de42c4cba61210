"""Would the training step's half- / quarter-resolution and transposed layers gain from the inference split kernels behind a layout
pass (to_split + Conv3dLayerX3(to_f32))?  fp32 form vs that route, per layer shape of the cfg4 hourglass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from snvc_amd import ops
dev = torch.device("cuda:0")
cases = {"hg conv2 fwd/dgrad: 64->64 s1 @96x48x156": (64, 64, (96, 48, 156), 1, False),
         "hg conv4: 64->64 s1 @48x24x78": (64, 64, (48, 24, 78), 1, False),
         "hg conv6 fwd: deconv 64->32 @96x48x156 -> full": (64, 32, (96, 48, 156), 2, True),
         "hg conv5 fwd / conv3 dgrad: deconv 64->64 @48x24x78 -> half": (64, 64, (48, 24, 78), 2, True),
         "hg conv3 fwd / conv5 dgrad: s2 64->64 @96x48x156 -> quarter": (64, 64, (96, 48, 156), 2, False),
         "conv2 fwd/dgrad: 32->32 s1 @192x96x312": (32, 32, (192, 96, 312), 1, False)}
for name, (ci, co, sp, st, tr) in cases.items():
    x = torch.relu(torch.randn(1, ci, *sp, device=dev))
    w = torch.randn((ci, co, 3, 3, 3) if tr else (co, ci, 3, 3, 3), device=dev) * 0.05
    f32 = ops.Conv3dLayer(w, 3, st, 1, 1, tr)
    x3 = ops.Conv3dLayerX3(w, 3, st, 1, 1, tr)
    ms_f, yf = bench.timed_ms(lambda: f32(x, None, None, None, 0, None), 10, 3)
    def route():
        mul = ops.split_scale_of(x)
        xs = ops.to_split(x, mul_dev=mul)
        return x3(xs, 0, None, None, flags=0, out_exp=0, to_f32=True, x_mul_dev=mul)
    ms_x, yx = bench.timed_ms(route, 10, 3)
    mul = ops.split_scale_of(x); xs = ops.to_split(x, mul_dev=mul)
    ms_k, _ = bench.timed_ms(lambda: x3(xs, 0, None, None, flags=0, out_exp=0, to_f32=True, x_mul_dev=mul), 10, 3)
    err = ((yx - yf).abs().max() / yf.abs().max()).item()
    print(f"{name:62s} fp32 {ms_f:.3f} ms | x3 route {ms_x:.3f} (kernel alone {ms_k:.3f}) | rel diff {err:.1e}", flush=True)
    del x, w, f32, x3, yf, yx, xs
    torch.cuda.empty_cache()
