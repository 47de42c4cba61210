"""r6: the twin-writing passes at the cfg4 layer sizes (GB/s of algorithmic bytes).  (The workgroup-count / chunking knobs of item 19 were environment variables of an experimental build; the product has the defaults.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from snvc_amd import ops
dev = torch.device("cuda:0")
for name, c, sp in (("full 32ch", 32, (192, 96, 312)), ("half 64ch", 64, (96, 48, 156)), ("quarter 64ch", 64, (48, 24, 78))):
    raw = torch.randn(1, c, *sp, device=dev); gy = torch.randn(1, c, *sp, device=dev)
    scale, shift = torch.rand(1, c, device=dev) + 0.5, torch.randn(1, c, device=dev)
    A, B, Cc = torch.randn(c, device=dev), torch.randn(c, device=dev) * 0.1, torch.randn(c, device=dev) * 0.01
    mul = torch.full((1,), 256.0, device=dev)
    nb = raw.numel() * 4
    t0, _ = bench.timed_ms(lambda: ops.affine_act(raw, scale, shift, None, ops.EPI_RELU, amax=ops.amax_word(dev)), 10, 3)
    t1, _ = bench.timed_ms(lambda: ops.affine_act(raw, scale, shift, None, ops.EPI_RELU, amax=ops.amax_word(dev), twin_mul=mul), 10, 3)
    t2, _ = bench.timed_ms(lambda: ops.act_backward_apply(raw, gy, None, scale, shift, A, B, Cc, ops.EPI_RELU, False, False, amax=ops.amax_word(dev)), 10, 3)
    t3, _ = bench.timed_ms(lambda: ops.act_backward_apply(raw, gy, None, scale, shift, A, B, Cc, ops.EPI_RELU, False, False, amax=ops.amax_word(dev), twin_mul=mul), 10, 3)
    t4, _ = bench.timed_ms(lambda: ops.act_backward_reduce(raw, gy, None, scale, shift, ops.EPI_RELU, False), 10, 3)
    t5, _ = bench.timed_ms(lambda: ops.act_backward_reduce(raw, gy, None, scale, shift, ops.EPI_RELU, False, amax_gy=ops.amax_word(dev)), 10, 3)
    t6, _ = bench.timed_ms(lambda: ops.norm_stats(raw, None, None, c, False, 1e-5), 10, 3)
    print(f"{name}: affine_act {t0*1e3:.0f} us ({2*nb/t0/1e9:.2f} TB/s) | +twin {t1*1e3:.0f} us ({3*nb/t1/1e9:.2f}) || bwd_apply {t2*1e3:.0f} us ({3*nb/t2/1e9:.2f}) "
          f"| +twin {t3*1e3:.0f} us ({4*nb/t3/1e9:.2f}) || bwd_reduce {t4*1e3:.0f} us ({2*nb/t4/1e9:.2f}) | +amax {t5*1e3:.0f} || norm_stats {t6*1e3:.0f} us ({nb/t6/1e9:.2f})", flush=True)
    del raw, gy
    torch.cuda.empty_cache()
