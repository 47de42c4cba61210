"""r6: the training step's forward and data-gradient convolutions on the split kernels (models/submodule.py: X3_TRAIN).
  * the passes that write a tensor's split TWIN beside it (snvc_affine_act_twin / snvc_act_backward_apply_twin), the scale from a
    device-side bound (snvc_split_scale_bound), max|gy| from the reduction pass (snvc_act_backward_reduce_amax);
  * the split kernels' float32 output with a float32 residual and with the batch statistics of the result from the same launch
    (snvc_f16x3_conv3d_forward with y_f32, snvc_f16x3_conv3d_forward_stats), device-scaled weights;
  * the route itself: an hourglass in train mode (reference snvc/models/submodule.py:85-168) on the split kernels against the fp32
    kernels and against torch-CPU autograd; tests/test_gpu_parity.py::test_training_step_* run the whole cfg4 step with the route on.
Reference semantics: nn.Conv3d / nn.ConvTranspose3d + train-mode nn.BatchNorm3d + ReLU as snvc/models/submodule.py:32-50,127-146 composes
them; the twin is a second representation of the same float32 tensor (x * 2^k = hi + lo in half), checked bit for bit."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_parity import check, dev, seeded

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())


@pytest.mark.parametrize("flags_res", [("relu", False), ("relu+pre", True), ("post", True), ("none", False)])
@pytest.mark.parametrize("shape", [(2, 64, (6, 8, 20)), (1, 32, (3, 5, 68)), (1, 8, (2, 3, 258))])
def test_affine_act_twin(flags_res, shape):
    """y is the non-twin pass's y bit for bit; the pair is to_split(y) bit for bit (full waves through the LDS transpose, tail waves
    stored directly); the maximum left in the words is max|y|"""
    from snvc_amd import ops
    name, with_res = flags_res
    flags = {"relu": ops.EPI_RELU, "relu+pre": ops.EPI_RELU | ops.EPI_ADD_PRE, "post": ops.EPI_ADD_POST, "none": 0}[name]
    n, c, sp = shape
    r = np.random.default_rng(5)
    raw, res = _t(r.standard_normal((n, c) + sp) * 3), (_t(r.standard_normal((n, c) + sp)) if with_res else None)
    scale, shift = _t(r.uniform(0.5, 2, (1, c))), _t(r.standard_normal((1, c)))
    ref = ops.affine_act(raw, scale, shift, res, flags)
    mul = ops.split_scale_of(ref)
    am = ops.amax_word(dev())
    got = ops.affine_act(raw, scale, shift, res, flags, amax=am, twin_mul=mul)
    assert torch.equal(got, ref)
    pair, m2 = ops.twin_of(got)
    assert m2 is mul and torch.equal(pair, ops.to_split(ref, mul_dev=mul))
    assert am.max().view(torch.float32).item() == ref.abs().max().item()
    got.add_(1.0)                                        # an in-place update voids the tag
    assert ops.twin_of(got) is None


@pytest.mark.parametrize("flags_res", [("relu", False), ("relu+pre", True), ("pre", True)])
@pytest.mark.parametrize("want_g", [False, True])
def test_act_backward_apply_twin(flags_res, want_g):
    from snvc_amd import ops
    name, with_res = flags_res
    flags = {"relu": ops.EPI_RELU, "relu+pre": ops.EPI_RELU | ops.EPI_ADD_PRE, "pre": ops.EPI_ADD_PRE}[name]
    n, c, sp = 2, 64, (5, 6, 44)
    r = np.random.default_rng(6)
    raw, gy = _t(r.standard_normal((n, c) + sp) * 3), _t(r.standard_normal((n, c) + sp) * 1e-4)
    res = _t(r.standard_normal((n, c) + sp)) if with_res else None
    scale, shift = _t(r.uniform(0.5, 2, (1, c))), _t(r.standard_normal((1, c)))
    A, B, Cc = _t(r.standard_normal(c)), _t(r.standard_normal(c) * 1e-5), _t(r.standard_normal(c) * 1e-6)
    d0, g0 = ops.act_backward_apply(raw, gy, res, scale, shift, A, B, Cc, flags, False, want_g)
    # max|gy| from the reduction pass, the sums unchanged by it
    amg = ops.amax_word(dev())
    s1 = ops.act_backward_reduce(raw, gy, res, scale, shift, flags, False, amax_gy=amg)
    s0 = ops.act_backward_reduce(raw, gy, res, scale, shift, flags, False)
    assert torch.equal(s0, s1) and amg.max().view(torch.float32).item() == gy.abs().max().item()
    l1 = _t(np.full(c, 2.0))
    ax = ops.amax_word(dev())
    ax[3:4] = (raw.abs().max() / 2).reshape(1).view(torch.int32)          # l1 * X = max|raw|
    mul = ops.split_scale_bound(c, c, dev(), a=A, amax_p=amg, b=B, l1=l1, amax_x=ax, cc=Cc)
    bound = float((A.abs() * gy.abs().max() + B.abs() * raw.abs().max() + Cc.abs()).max())
    m = mul.item()
    assert m == 2.0 ** round(np.log2(m)) and 8192.0 <= bound * m < 16384.0 * (1 + 1e-6)
    d1, g1 = ops.act_backward_apply(raw, gy, res, scale, shift, A, B, Cc, flags, False, want_g, twin_mul=mul)
    assert torch.equal(d0, d1) and (not want_g or torch.equal(g0, g1))
    pair, _ = ops.twin_of(d1)
    assert torch.equal(pair, ops.to_split(d0, mul_dev=mul))
    assert float(d0.abs().max()) <= bound * (1 + 1e-6)


def test_split_scale_bound_edge_cases():
    from snvc_amd import ops
    z = ops.amax_word(dev())
    assert ops.split_scale_bound(1, 1, dev(), amax_x=z).item() == 1.0                 # an all-zero tensor: scale 1
    w = ops.amax_word(dev())
    w[0:1] = torch.tensor([float("inf")], device=dev()).view(torch.int32)
    assert ops.split_scale_bound(1, 1, dev(), amax_x=w).item() == 1.0                 # a non-finite maximum: scale 1
    for v in (1e-30, 3.0, 8191.9, 8192.0, 1e20):
        w = ops.amax_word(dev())
        w[5:6] = torch.tensor([v], device=dev()).view(torch.int32)
        m = ops.split_scale_bound(1, 1, dev(), amax_x=w).item()
        e = int(np.floor(np.log2(np.float32(v)))) + 1
        assert m == 2.0 ** max(-24, min(40, 14 - e)), (v, m)
    # per-sample rows, the residual's maximum added
    sc, sh = _t([[1.0, -2.0], [0.5, 3.0]]), _t([[0.1, 0.2], [0.3, -0.4]])
    l1 = _t([2.0, 4.0])
    x, r = ops.amax_word(dev()), ops.amax_word(dev())
    x[0:1] = torch.tensor([1.5], device=dev()).view(torch.int32)
    r[9:10] = torch.tensor([7.0], device=dev()).view(torch.int32)
    m = ops.split_scale_bound(4, 2, dev(), b=sc.reshape(-1).contiguous(), l1=l1, amax_x=x, cc=sh.reshape(-1).contiguous(), amax_r=r).item()
    bound = max(1 * 2 * 1.5 + 0.1, 2 * 4 * 1.5 + 0.2, 0.5 * 2 * 1.5 + 0.3, 3 * 4 * 1.5 + 0.4) + 7.0
    assert 8192.0 <= bound * m < 16384.0


_LAYERS = {"s1 64->64 (16x16x32 form)": (64, 64, (8, 8, 64), 1, False), "s1 32->32": (32, 32, (6, 9, 40), 1, False),
           "s1 64->32 small": (64, 32, (4, 4, 32), 1, False), "s2 32->64": (32, 64, (8, 8, 64), 2, False),
           "s2 64->64": (64, 64, (6, 12, 40), 2, False), "deconv 64->32": (64, 32, (4, 6, 40), 2, True),
           "deconv 64->64": (64, 64, (3, 5, 36), 2, True)}


@pytest.mark.parametrize("case", list(_LAYERS))
def test_split_layer_f32_output_residual_and_statistics(case):
    """Conv3dLayerX3 with device-scaled weights (no host read of max|w|): float32 result against float64 torch; + a float32 residual =
    result + residual bit for bit; forward_stats = the same result bit for bit with scale / shift / mean / var as norm_stats of it"""
    from snvc_amd import ops
    ci, co, sp, st, tr = _LAYERS[case]
    r = np.random.default_rng(9)
    x = torch.relu(_t(r.standard_normal((2, ci) + sp)))
    w = _t(r.standard_normal(((ci, co) if tr else (co, ci)) + (3, 3, 3)) * 0.05)
    lay = ops.Conv3dLayerX3(w, 3, st, 1, 1, tr, w_mul_dev=ops.split_scale_of(w))
    mul = ops.split_scale_of(x)
    xs = ops.to_split(x, mul_dev=mul)
    y = lay(xs, 0, None, None, to_f32=True, x_mul_dev=mul)
    if tr:
        ref = F.conv_transpose3d(x.double().cpu(), w.double().cpu(), stride=2, padding=1, output_padding=1)
    else:
        ref = F.conv3d(x.double().cpu(), w.double().cpu(), stride=st, padding=1)
    check(y.cpu().numpy(), ref.float().numpy(), 2e-6, case)
    extra = _t(r.standard_normal(tuple(y.shape)))
    y2 = lay(xs, 0, None, None, to_f32=True, x_mul_dev=mul, residual_f32=extra)
    assert torch.equal(y2, y + extra)
    # snvc_f16x3_conv3d_forward_f32: the input's scale taken out inside the kernel (a device pointer) instead of folded into the scale vector
    assert torch.equal(lay.forward_f32(xs, mul), y) and torch.equal(lay.forward_f32(xs, mul, residual_f32=extra), y2)
    assert torch.equal(lay.forward_f32(xs, mul, relu=True), torch.relu(y))
    gamma, beta = _t(r.uniform(0.5, 2, co)), _t(r.standard_normal(co))
    got = lay.forward_stats(xs, mul, gamma, beta, 1e-5)
    assert got is not None, "this kernel form carries the statistics epilogue"
    raw, scale, shift, mean, var = got
    assert torch.equal(raw, y)
    s0, h0, m0, v0 = ops.norm_stats(y, gamma, beta, co, False, 1e-5)
    for a, b, what in ((scale, s0, "scale"), (shift, h0, "shift"), (mean, m0, "mean"), (var, v0, "var")):
        check(a.cpu().numpy(), b.cpu().numpy(), 2e-6, f"{case}: {what} from the epilogue")


def _hourglass_grads(hg, x0, on):
    from snvc_amd.models import submodule as S
    S.X3_TRAIN[0] = on
    try:
        for p in hg.parameters():
            p.grad = None
        x = x0.clone().requires_grad_()
        xa = x * 1.0
        o, pre, post = hg(xa, None, None, residual=xa)
        (o.pow(2).mean() + 0.1 * pre.mean() + 0.05 * post.pow(2).mean()).backward()
        return o.detach(), x.grad.clone(), {k: p.grad.clone() for k, p in hg.named_parameters()}
    finally:
        S.X3_TRAIN[0] = True


def test_hourglass_train_route_vs_fp32_kernels_and_torch_autograd():
    """One train-mode hourglass step (snvc/models/submodule.py:85-168: stride-2, stride-1 and transposed layers, both skip connections)
    on the split kernels: against the fp32 kernels of the same package, and against torch-CPU autograd on the oracle's restatement.
    From the second step on the operands come through twins their producers wrote."""
    from oracle import torch_ref as T
    from snvc_amd.models import submodule as S
    ref = seeded(T.hourglass(32), 91).train()
    ours = seeded(S.hourglass(32), 91).to(dev()).train()
    r = np.random.default_rng(92)
    X = np.maximum(r.standard_normal((2, 32, 8, 16, 40)), 0).astype(np.float32)
    xr = torch.from_numpy(X).requires_grad_()
    xa = xr * 1.0
    o, pre, post = ref(xa, None, None)
    o = o + xa                                             # the caller's residual (vernier.py:370), folded into conv6 on our side
    (o.pow(2).mean() + 0.1 * pre.mean() + 0.05 * post.pow(2).mean()).backward()
    x0 = torch.from_numpy(X).to(dev())
    o0, gx0, gp0 = _hourglass_grads(ours, x0, False)
    b = dict(S._ROUTES)
    o1, gx1, gp1 = _hourglass_grads(ours, x0, True)
    assert S._ROUTES["x3_train_dgrad"] == b.get("x3_train_dgrad", 0) + 6
    b = dict(S._ROUTES)
    o2, gx2, gp2 = _hourglass_grads(ours, x0, True)        # the producers now write the twins their consumers asked for
    # the caller's tensor has no producer on the path (no twin, no maximum): its own operand uses and conv1's draw take a layout pass
    assert S._ROUTES["x3_train_layout_pass"] - b.get("x3_train_layout_pass", 0) <= 3
    assert S._ROUTES["x3_train_twin"] - b.get("x3_train_twin", 0) >= 9
    check(o2.cpu().numpy(), o1.cpu().numpy(), 2e-6, "twins (scale from a bound) vs layout passes (scale from the maximum)")
    check(o1.cpu().numpy(), o0.cpu().numpy(), 2e-5, "out: split vs fp32 kernels")
    check(gx2.cpu().numpy(), gx0.cpu().numpy(), 2e-5, "dx: split vs fp32 kernels")
    check(o1.cpu().numpy(), o.detach().numpy(), 1e-4, "out vs torch")
    check(gx2.cpu().numpy(), xr.grad.numpy(), 1e-4, "dx vs torch autograd")
    for (k, p) in ref.named_parameters():
        check(gp2[k].cpu().numpy(), p.grad.numpy(), 2e-4, f"d{k} vs torch autograd")
        check(gp2[k].cpu().numpy(), gp0[k].cpu().numpy(), 5e-5, f"d{k}: split vs fp32 kernels")


def test_which_layers_take_the_route():
    """whole 32-channel blocks on both sides, 3x3x3, stride 1 / 2 / transposed; an odd extent under a stride-2 layer keeps the fp32 route
    (its crop / pad handling lives there); the one-channel classifier and the 7^3 / 5^3 layers of the local trunk stay where they are"""
    import torch.nn as nn
    from snvc_amd.models import submodule as S
    x = torch.zeros(1, 32, 6, 10, 36, device=dev())
    assert S._x3_train_route(nn.Conv3d(32, 32, 3, 1, 1, bias=False), x)
    assert S._x3_train_route(nn.Conv3d(32, 64, 3, 2, 1, bias=False), x)
    assert S._x3_train_route(nn.ConvTranspose3d(32, 64, 3, 2, 1, output_padding=1, bias=False), x)
    assert not S._x3_train_route(nn.Conv3d(32, 64, 3, 2, 1, bias=False), x[:, :, :5])          # odd depth
    assert not S._x3_train_route(nn.Conv3d(32, 1, 3, 1, 1, bias=False), x)
    assert not S._x3_train_route(nn.Conv3d(32, 32, 5, 1, 2, bias=False), x)
    assert not S._x3_train_route(nn.Conv3d(32, 32, 3, 1, 1, bias=False), x.cpu())
    S.X3_TRAIN[0] = False
    try:
        assert not S._x3_train_route(nn.Conv3d(32, 32, 3, 1, 1, bias=False), x)
    finally:
        S.X3_TRAIN[0] = True


def test_global_stack_step_takes_no_layout_pass_from_the_second_step_on():
    """cfg4 in miniature (build_cost_volume + GlobalStack, train-mode BatchNorm): conv2 and the hourglass's six layers run forward and
    data gradient on the split kernels (7 layers x 2 operands = 14 twins per step), the fused first layer hands conv2 its twin
    (snvc_sheared_expand_split), and no tensor goes through a layout pass of its own; the second step's loss and gradients equal the
    first step's bit for bit (same inputs: the twins are a representation, not a different arithmetic)."""
    from snvc_amd.models import submodule as S
    from snvc_amd.models.stereo_volume import GlobalStack
    r = np.random.default_rng(95)
    C, H, W, D = 32, 8, 40, 8
    model = seeded(GlobalStack(C), 96).to(dev()).train()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.momentum = 0.0                      # the running statistics do not enter a train-mode step; keep them fixed anyway
    L, R = _t(r.standard_normal((1, C, H, W))).requires_grad_(), _t(r.standard_normal((1, C, H, W))).requires_grad_()
    sh = torch.arange(D, dtype=torch.float32, device=dev())[None].contiguous()
    outs = []
    for step in range(3):
        for p in model.parameters():
            p.grad = None
        L.grad = R.grad = None
        b = dict(S._ROUTES)
        loss = model.forward_pair(L, R, sh, 1).pow(2).mean()
        loss.backward()
        d = {k: S._ROUTES[k] - b.get(k, 0) for k in ("x3_train_layout_pass", "x3_train_twin", "x3_train_dgrad")}
        outs.append((loss.detach().clone(), L.grad.clone(), d))
    assert outs[0][2]["x3_train_dgrad"] == 7
    for step in (1, 2):
        assert outs[step][2] == {"x3_train_layout_pass": 0, "x3_train_twin": 14, "x3_train_dgrad": 7}, outs[step][2]
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])
    check(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy(), 1e-5, "step with twins vs step with layout passes")


@pytest.mark.parametrize("shape", [(1, 32, 4, 12, 20), (2, 32, 8, 8, 36)])
def test_hourglass_route_with_tensors_that_cannot_carry_a_twin(shape):
    """voxel counts that are not a multiple of 4 at the quarter-resolution level (15 voxels; 36 at the second shape is one, its rows are
    not): the producing passes skip the twin where ops.twin_ok says so, the consumer converts by itself -- same results as the fp32 kernels"""
    from snvc_amd.models import submodule as S
    hg = seeded(S.hourglass(32), 97).to(dev()).train()
    x0 = torch.relu(_t(np.random.default_rng(98).standard_normal(shape)))
    o0, gx0, gp0 = _hourglass_grads(hg, x0, False)
    _hourglass_grads(hg, x0, True)
    b = dict(S._ROUTES)
    o1, gx1, gp1 = _hourglass_grads(hg, x0, True)
    assert S._ROUTES["x3_train_dgrad"] - b.get("x3_train_dgrad", 0) == 6
    check(o1.cpu().numpy(), o0.cpu().numpy(), 2e-5, "out")
    check(gx1.cpu().numpy(), gx0.cpu().numpy(), 3e-5, "dx")
    for k in gp0:
        check(gp1[k].cpu().numpy(), gp0[k].cpu().numpy(), 1e-4, f"d{k}")


@pytest.mark.parametrize("norm", ["frozen_bn", "groupnorm", "none"])
def test_single_layer_route_with_other_norms(norm):
    """the route under a frozen (eval-mode) BatchNorm3d (draw = A * g: no raw term in the twin's bound), under GroupNorm (per-sample scale /
    shift rows in the bound) and without a norm (draw is gy itself: no pass writes a twin, the data gradient converts): dx, dW and the
    norm's gradients against the fp32 kernels"""
    import torch.nn as nn
    from snvc_amd.models import submodule as S
    layer = seeded(S.ConvBNReLU3d(S.convbn_3d(64, 64, 3, 1, 1, gn=(norm == "groupnorm")), nn.ReLU(inplace=True)), 99).to(dev()).train()
    if norm == "frozen_bn":
        layer[0][1].eval()
    if norm == "none":
        layer = seeded(S.HipConv3d(64, 64, 3, 1, 1, bias=False), 99).to(dev()).train()
    x0 = torch.relu(_t(np.random.default_rng(100).standard_normal((2, 64, 4, 8, 36))))
    prev = seeded(S.ConvBNReLU3d(S.convbn_3d(64, 64, 3, 1, 1), nn.ReLU(inplace=True)), 101).to(dev()).train()      # a producer on the path
    outs = {}
    for on in (False, True, True):
        S.X3_TRAIN[0] = on
        try:
            for p in list(layer.parameters()) + list(prev.parameters()):
                p.grad = None
            x = x0.clone().requires_grad_()
            y = layer.fused(prev.fused(x)) if hasattr(layer, "fused") else layer(prev.fused(x))
            y.pow(2).mean().backward()
            outs[on] = (y.detach(), x.grad.clone(), [p.grad.clone() for p in layer.parameters()])
        finally:
            S.X3_TRAIN[0] = True
    check(outs[True][0].cpu().numpy(), outs[False][0].cpu().numpy(), 2e-5, f"{norm}: out")
    check(outs[True][1].cpu().numpy(), outs[False][1].cpu().numpy(), 3e-5, f"{norm}: dx")
    for a, b in zip(outs[True][2], outs[False][2]):
        check(a.cpu().numpy(), b.cpu().numpy(), 1e-4, f"{norm}: parameter gradient")


@pytest.mark.parametrize("cin", [1, 2])
@pytest.mark.parametrize("flags", ["none", "relu+post"])
def test_pointwise_layer_from_one_or_two_channels(cin, flags):
    """nn.Conv3d(k1) from <= 2 channels (r6: the classifier's data gradient, snvc/models/submodule.py classifier; gx[c] = w[c] * gy) on the
    streaming kernel: against float64 torch, and equal to the MFMA tile form (ALGO_DIRECT keeps it) -- a one-term sum has one rounding"""
    from snvc_amd import _lib, ops
    r = np.random.default_rng(110 + cin)
    x = _t(r.standard_normal((2, cin, 4, 6, 40)))
    w = _t(r.standard_normal((32, cin, 1, 1, 1)))
    res = _t(r.standard_normal((2, 32, 4, 6, 40))) if flags != "none" else None
    fl = (ops.EPI_RELU | ops.EPI_ADD_POST) if res is not None else 0
    lay = ops.Conv3dLayer(w, 1, 1, 0, 1, False)
    y = lay(x, None, None, res, fl, None)
    ref = F.conv3d(x.double().cpu(), w.double().cpu())
    if res is not None:
        ref = torch.relu(ref) + res.double().cpu()
    check(y.cpu().numpy(), ref.float().numpy(), 1e-6, "pointwise expand")
    with ops.conv_variant(_lib.ALGO_DIRECT):
        y2 = ops.Conv3dLayer(w, 1, 1, 0, 1, False)(x, None, None, res, fl, None)
    if cin == 1:
        assert torch.equal(y, y2)
    else:
        check(y.cpu().numpy(), y2.cpu().numpy(), 1e-6, "streaming vs MFMA tile form")


def test_bn_bookkeeping_in_one_launch():
    """snvc_bn_track against nn.BatchNorm3d's own train-mode update (torch/nn/modules/batchnorm.py): the same running statistics bit for
    bit, the counter advanced, and the tensors' version counters bumped (the folded eval-mode BatchNorm is cached on them)"""
    import torch.nn as nn
    from snvc_amd import ops
    from snvc_amd.models import submodule as S
    r = np.random.default_rng(120)
    x = _t(r.standard_normal((2, 32, 3, 4, 20)) * 2 + 0.5)
    for momentum in (0.1, 0.7):
        ref, ours = nn.BatchNorm3d(32, momentum=momentum).to(dev()).train(), nn.BatchNorm3d(32, momentum=momentum).to(dev()).train()
        ref(x)
        mean, var = x.mean((0, 2, 3, 4)), x.var((0, 2, 3, 4), unbiased=False)
        v0 = ours.running_mean._version
        assert ops.bn_track(ours, mean.contiguous(), var.contiguous(), x.numel() / 32)
        assert ours.running_mean._version > v0 and int(ours.num_batches_tracked) == 1
        check(ours.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy(), 1e-6, "running_mean")
        check(ours.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), 1e-6, "running_var")
        # torch's own two lerp_ on the same inputs: bit for bit
        t = nn.BatchNorm3d(32, momentum=momentum).to(dev())
        t.running_mean.lerp_(mean, momentum)
        t.running_var.lerp_(var * (x.numel() / 32 / (x.numel() / 32 - 1)), momentum)
        assert torch.equal(t.running_mean, ours.running_mean) and torch.equal(t.running_var, ours.running_var)
    cum = nn.BatchNorm3d(32, momentum=None).to(dev()).train()
    assert not ops.bn_track(cum, mean.contiguous(), var.contiguous(), 100.0)        # cumulative average: the torch path keeps it
    # a layer trained one step, then evaluated: the folded BatchNorm sees the new statistics
    layer = seeded(S.ConvBNReLU3d(S.convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True)), 121).to(dev())
    xin = torch.relu(_t(r.standard_normal((1, 32, 4, 6, 20))))
    layer.eval()
    with torch.no_grad():
        y_before = layer.fused(xin).clone()
    layer.train()
    layer.fused(xin.clone().requires_grad_()).mean().backward()
    layer.eval()
    with torch.no_grad():
        y_after = layer.fused(xin)
        ref_after = torch.relu(layer[0][1](torch.nn.functional.conv3d(xin, layer[0][0].weight, padding=1)))
    assert not torch.equal(y_before, y_after)
    check(y_after.cpu().numpy(), ref_after.cpu().numpy(), 1e-5, "eval after one training step")


@pytest.mark.parametrize("q", [1, 2, 4])
def test_sheared_expand_leaves_the_maximum(q):
    """snvc_sheared_expand_amax: the same tensor as snvc_sheared_expand bit for bit, and max|y| in the words, bit for bit"""
    from snvc_amd import ops
    from snvc_amd.models import submodule as S
    r = np.random.default_rng(130 + q)
    n, c, d, h, w, m0 = 2, 32, 12, 5, 40, 3
    off, wu, off_col, wu_col = S.sheared_geometry(q, m0, d, w)      # any consistent geometry: the kernel reads zeros outside [0, WG)
    g, gcol = _t(r.standard_normal((n, 3 * c, h, wu))), _t(r.standard_normal((n, 3 * c, h, wu_col)))
    planes = _t(r.standard_normal((n, c, 3, h, w)))
    scale, shift = _t(r.uniform(0.5, 2, c)), _t(r.standard_normal(c))
    y0 = torch.empty((n, c, d, h, w), device=dev())
    y1 = torch.empty_like(y0)
    ops.sheared_expand(g, gcol, planes, scale, shift, y0, q, m0, off, off_col, ops.EPI_RELU)
    am = ops.amax_word(dev())
    ops.sheared_expand(g, gcol, planes, scale, shift, y1, q, m0, off, off_col, ops.EPI_RELU, amax=am)
    assert torch.equal(y0, y1)
    assert am.max().view(torch.float32).item() == y0.abs().max().item()


def test_cfg4_full_size_step_split_route_vs_fp32_route():
    """BASELINE.json configs[3] at its full size (features [1,32,96,312], 192 half-pixel planes: a 5.75 M-voxel grid, 736 MB per
    32-channel tensor): the training step with conv2 + hourglass on the split kernels against the same step on the fp32 kernels --
    the loss, every parameter gradient and both feature gradients; size-independent properties of the route: no layout pass and 14
    twins from the second step on, the running statistics moved once per step either way."""
    from snvc_amd.models import submodule as S
    from snvc_amd.models.stereo_volume import GlobalStack
    r = np.random.default_rng(140)
    C, H, W, D = 32, 96, 312, 192
    left, right = _t(r.standard_normal((1, C, H, W))).requires_grad_(), _t(r.standard_normal((1, C, H, W))).requires_grad_()
    shift = torch.from_numpy(np.linspace(0.0, (D - 1) / 2.0, D, dtype=np.float32)[None].copy()).to(dev())
    res = {}
    for on in (False, True):
        S.X3_TRAIN[0] = on
        try:
            model = seeded(GlobalStack(C), 141).to(dev()).train()
            for step in range(2):
                for p in model.parameters():
                    p.grad = None
                left.grad = right.grad = None
                b = dict(S._ROUTES)
                loss = model.forward_pair(left, right, shift, 1).pow(2).mean()
                loss.backward()
            d = {k: S._ROUTES[k] - b.get(k, 0) for k in ("x3_train_layout_pass", "x3_train_twin", "x3_train_dgrad")}
            res[on] = (loss.item(), left.grad.clone(), right.grad.clone(), {k: p.grad.clone() for k, p in model.named_parameters()}, d,
                       {k: v.clone() for k, v in model.state_dict().items() if "running_mean" in k})
            del model
            torch.cuda.empty_cache()
        finally:
            S.X3_TRAIN[0] = True
    assert res[True][4] == {"x3_train_layout_pass": 0, "x3_train_twin": 14, "x3_train_dgrad": 7}, res[True][4]
    assert res[False][4] == {"x3_train_layout_pass": 0, "x3_train_twin": 0, "x3_train_dgrad": 0}
    assert np.isfinite(res[True][0]) and abs(res[True][0] - res[False][0]) <= 2e-6 * abs(res[False][0])
    # the bounds of tests/test_gpu_fullsize.py::test_training_step_full_size_properties (DIRECT vs Winograd fp32 forms of the same step): two
    # arithmetics ~2e-6 of the range apart per layer flip ~1e-5 of the 1e9 ReLU masks; a parameter gradient sums over every voxel
    # (flips average out), a feature-map gradient at one pixel sums ~1.7e5 terms (one flip moves it by a fraction of a percent)
    def err(a, b):
        return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    e_l, e_r = err(res[True][1], res[False][1]), err(res[True][2], res[False][2])
    assert e_l < 5e-2 and e_r < 5e-2, (e_l, e_r)
    worst = max((err(res[True][3][k], res[False][3][k]), k) for k in res[False][3])
    assert worst[0] < 5e-3, worst
    for k in res[False][5]:
        assert err(res[True][5][k], res[False][5][k]) < 1e-4, k           # BatchNorm bookkeeping moved the same way (two steps each)
