"""r6: the training step's forward / data-gradient convolutions on the split kernels (submodule.X3_TRAIN).
  1. the twin-writing passes against to_split of their float32 result;
  2. an hourglass in train mode, route on vs off: outputs and every gradient;
  3. the cfg4 step with the route off / on (and with the weights touched every step, so that the split layers re-pack)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snvc_amd import ops
from snvc_amd.models import submodule as S

dev = torch.device("cuda:0")
what = sys.argv[1:] or ["unit", "hg", "step"]
for w_ in what:
    if w_.startswith("cc="):
        S.X3_TRAIN_MIN_CC[0] = int(w_[3:])


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


if "unit" in what:
    torch.manual_seed(1)
    n, c, sp = 2, 64, (6, 8, 20)
    raw = torch.randn(n, c, *sp, device=dev) * 3
    res = torch.randn(n, c, *sp, device=dev)
    scale, shift = torch.randn(1, c, device=dev), torch.randn(1, c, device=dev)
    for flags, r in ((ops.EPI_RELU, None), (ops.EPI_RELU | ops.EPI_ADD_PRE, res), (ops.EPI_ADD_POST, res), (0, None)):
        ref = ops.affine_act(raw, scale, shift, r, flags)
        mul = ops.split_scale_of(ref)
        am = ops.amax_word(dev)
        got = ops.affine_act(raw, scale, shift, r, flags, amax=am, twin_mul=mul)
        pair, m2 = ops.twin_of(got)
        back = ops.from_split(pair) / mul
        want = ops.from_split(ops.to_split(ref, mul_dev=mul)) / mul
        print(f"affine_act twin flags {flags}: y equal {torch.equal(got, ref)}, twin == to_split {torch.equal(back, want)}, "
              f"twin vs y {rel(back, ref):.1e}, amax {am.max().view(torch.float32).item():.4f} vs {ref.abs().max().item():.4f}")
    gy = torch.randn(n, c, *sp, device=dev)
    A, B, Cc = torch.randn(c, device=dev), torch.randn(c, device=dev) * 0.1, torch.randn(c, device=dev) * 0.01
    for flags, r in ((ops.EPI_RELU, None), (ops.EPI_RELU | ops.EPI_ADD_PRE, res)):
        d0, g0 = ops.act_backward_apply(raw, gy, r, scale, shift, A, B, Cc, flags, False, True)
        amg = ops.amax_word(dev)
        ops.act_backward_reduce(raw, gy, r, scale, shift, flags, False, amax_gy=amg)
        l1 = torch.full((c,), 2.0, device=dev)
        ax = ops.amax_word(dev); ax[0:1] = raw.abs().max().reshape(1).view(torch.int32) if False else (raw.abs().max() / 2).reshape(1).view(torch.int32)
        mul = ops.split_scale_bound(c, c, dev, a=A, amax_p=amg, b=B, l1=l1, amax_x=ax, cc=Cc)
        bound = (A.abs() * gy.abs().max() + B.abs() * raw.abs().max() + Cc.abs()).max()
        d1, g1 = ops.act_backward_apply(raw, gy, r, scale, shift, A, B, Cc, flags, False, True, twin_mul=mul)
        pair, _ = ops.twin_of(d1)
        back = ops.from_split(pair) / mul
        print(f"act_bwd_apply twin flags {flags}: draw equal {torch.equal(d0, d1)}, g equal {torch.equal(g0, g1)}, twin vs draw {rel(back, d0):.1e}, "
              f"mul {mul.item():g} bound {bound.item():.3f} -> bound*mul {bound.item() * mul.item():.0f} (in [8192, 16384)), max|gy| "
              f"{amg.max().view(torch.float32).item():.4f} vs {gy.abs().max().item():.4f}")
    # fp32 residual on the split kernels' fp32 output
    x = torch.relu(torch.randn(1, 64, 6, 8, 32, device=dev))
    w = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
    extra = torch.randn(1, 32, 12, 16, 64, device=dev)
    lay = ops.Conv3dLayerX3(w, 3, 2, 1, 1, True, w_mul_dev=ops.split_scale_of(w))
    mul = ops.split_scale_of(x)
    xs = ops.to_split(x, mul_dev=mul)
    y0 = lay(xs, 0, None, None, to_f32=True, x_mul_dev=mul)
    y1 = lay(xs, 0, None, None, to_f32=True, x_mul_dev=mul, residual_f32=extra)
    yr = torch.nn.functional.conv_transpose3d(x.double(), w.double(), stride=2, padding=1, output_padding=1).float()
    print(f"deconv on the split kernel (device-scaled weights) vs float64 {rel(y0, yr):.1e}; + fp32 residual {rel(y1, yr + extra):.1e}")

if "hg" in what:
    from snvc_amd.models.submodule import hourglass
    torch.manual_seed(2)
    hg = hourglass(32).to(dev).train()
    x0 = torch.relu(torch.randn(2, 32, 8, 16, 40, device=dev))
    outs = {}
    for on in (False, True, True):
        S.X3_TRAIN[0] = on
        for p in hg.parameters():
            p.grad = None
        x = x0.clone().requires_grad_()
        xa = x * 1.0                              # a non-leaf with a producer-less history, as conv2's output is for the hourglass
        b = dict(S._ROUTES)
        o, pre, post = hg(xa, None, None, residual=xa)
        (o.pow(2).mean() + pre.mean() * 0.1).backward()
        routes = {k: v - b.get(k, 0) for k, v in S._ROUTES.items() if v - b.get(k, 0) and k.startswith("x3_train")}
        outs[on] = (o.detach(), x.grad.clone(), {n_: p.grad.clone() for n_, p in hg.named_parameters()})
        print("route", on, routes)
    o0, gx0, gp0 = outs[False]
    o1, gx1, gp1 = outs[True]
    print(f"hourglass train-mode: out {rel(o1, o0):.1e}, dx {rel(gx1, gx0):.1e}, worst parameter gradient "
          f"{max(rel(gp1[k], gp0[k]) for k in gp0):.1e}")
    S.X3_TRAIN[0] = True

if "step" in what:
    import bench
    from benchlib.train import TrainStep
    combos = ((False, False, None), (True, False, None), (False, True, None), (True, True, None))
    if "on" in what: combos = ((True, False, None),)
    if "off" in what: combos = ((False, False, None),)
    if "touch" in what: combos = ((True, True, None),)
    for w_ in what:
        if w_.startswith("ab="):            # ab=1024,2048,1024,2048: the route's channel threshold, alternating in one process
            combos = tuple((True, False, int(v)) for v in w_[3:].split(","))
    for on, touch, cc in combos:
        S.X3_TRAIN[0] = on
        if cc is not None:
            S.X3_TRAIN_MIN_CC[0] = cc
        ts = TrainStep(0, dev)
        for _ in range(4):
            ts()
        torch.cuda.synchronize()
        b = dict(S._ROUTES)
        import time
        t0 = time.perf_counter()
        K = 30
        acc = [0.0, 0.0, 0.0]
        for _ in range(K):
            if touch:
                with torch.no_grad():
                    torch._foreach_mul_(list(ts.model.parameters()), 1.0)
            loss = ts()
            torch.cuda.synchronize()
            for i, v in enumerate(ts.phases_ms()):
                acc[i] += v
        ms = (time.perf_counter() - t0) / K * 1e3
        routes = {k: (v - b.get(k, 0)) // K for k, v in S._ROUTES.items() if v - b.get(k, 0) and k.startswith("x3_train")}
        print(f"cfg4 step (min Cin*Cout {S.X3_TRAIN_MIN_CC[0]}), split forward/dgrad {'ON ' if on else 'OFF'}{' (weights touched every step)' if touch else ''}: {ms:.2f} ms "
              f"(fwd {acc[0] / K:.2f}, bwd {acc[1] / K:.2f}), loss {loss.item():.6e}, per step {routes}", flush=True)
        del ts
        torch.cuda.empty_cache()
