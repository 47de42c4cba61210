"""GPU parity of the 2D BEV neck + heads, layer by layer: the depth-1 HIP conv kernels behind
``fused_conv2d`` / ``fused_deconv2d`` / ``zero_stuff2x`` and the blocks built from them (``BasicBlock2d``,
``hourglass2d``, ``hourglass2d_downsample_16``; reference snvc/models/submodule.py:11-29,270-361, hrnet.py:25-69,
vernier.py:68-93) against torch's fp32 operators on the CPU and the pinned oracle modules, at the tolerance an exact
fp32 FMA chain in another summation order meets.  Every block test also asserts through ``_ROUTES`` that the HIP route
-- not the modules' torch forward -- is what ran, and that anything with a gradient to compute keeps torch's route.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_parity import TIGHT, check, dev, seeded

pytestmark = pytest.mark.gpu


def _t(r, shape):
    return torch.from_numpy(r.standard_normal(shape).astype(np.float32))


def _bn_eval(y, bn):
    return F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)


@pytest.mark.parametrize("k,stride", [(1, 1), (1, 2), (3, 1), (3, 2)])
@pytest.mark.parametrize("cin,cout,hw", [(32, 64, (24, 40)), (9, 32, (6, 4)), (64, 27, (13, 34)), (256, 64, (16, 24)),
                                         # >= 32 x 32 with 16-byte rows and whole channel groups: k3 / stride 1 takes the depth-1
                                         # Winograd form (1 x 16 x 32 tiles: ragged in H and W here, odd channel count)
                                         (64, 64, (40, 72)), (7, 32, (33, 36)), (256, 64, (64, 96))])
def test_fused_conv2d_vs_torch(k, stride, cin, cout, hw):
    from snvc_amd.models import submodule as S
    r = np.random.default_rng(1000 + 10 * k + stride + cin)
    seq = seeded(S.convbn(cin, cout, k, stride, (k - 1) // 2, 1), 7 + k)
    conv, bn = seq[0], seq[1]
    x = _t(r, (2, cin) + hw)
    with torch.no_grad():
        raw = F.conv2d(x, conv.weight, None, stride, (k - 1) // 2)
        ref = _bn_eval(raw, bn)
        res = _t(r, tuple(ref.shape))
        seq = seq.to(dev())
        xd, rd = x.to(dev()), res.to(dev())
        conv, bn = seq[0], seq[1]
        check(S.fused_conv2d(conv, bn, xd).cpu().numpy(), ref.numpy(), TIGHT, "bn(conv)")
        check(S.fused_conv2d(conv, None, xd).cpu().numpy(), raw.numpy(), TIGHT, "conv")
        check(S.fused_conv2d(conv, bn, xd, relu=True).cpu().numpy(), F.relu(ref).numpy(), TIGHT, "relu(bn(conv))")
        check(S.fused_conv2d(conv, bn, xd, relu=True, residual=rd).cpu().numpy(), F.relu(ref + res).numpy(), TIGHT,
              "relu(bn(conv) + res)")
        check(S.fused_conv2d(conv, bn, xd, relu=True, residual=rd, residual_after_act=True).cpu().numpy(),
              (F.relu(ref) + res).numpy(), TIGHT, "relu(bn(conv)) + res")
        check(S.fused_conv2d(conv, bn, xd, sigmoid=True).cpu().numpy(), torch.sigmoid(ref).numpy(), TIGHT, "sigmoid(bn(conv))")
        if k == 3 and stride == 1 and min(hw) >= 32:
            # the depth-1 Winograd form (taken by itself from two jobs per CU on; forced here): bit-different from the direct form
            from snvc_amd import _lib, ops
            direct = S.fused_conv2d(conv, bn, xd, relu=True, residual=rd)
            with ops.conv_variant(_lib.ALGO_WINO_TILE_BIG):
                wino = S.fused_conv2d(conv, bn, xd, relu=True, residual=rd)
                check(wino.cpu().numpy(), F.relu(ref + res).numpy(), TIGHT, "relu(bn(conv) + res), depth-1 Winograd")
                check(S.fused_conv2d(conv, None, xd).cpu().numpy(), raw.numpy(), TIGHT, "conv, depth-1 Winograd")
            if cout % 32 == 0 and hw[1] % 4 == 0:
                assert not torch.equal(wino, direct), "the forced form is a different kernel"


def test_fused_conv2d_transposed_input_is_a_swapped_kernel():
    """hm2(feats.permute(0, 1, 3, 2)) (vernier.py:441-442) without the copy: the convolution of feats with the kernel's
    spatial axes swapped, returned as a transposed view."""
    from snvc_amd.models import submodule as S
    r = np.random.default_rng(1150)
    for cin, cout, hw, bias in ((64, 9, (24, 16), False), (16, 32, (13, 40), True)):
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=bias)
        conv.weight.data.copy_(_t(r, tuple(conv.weight.shape)) * 0.1)
        if bias:
            conv.bias.data.copy_(_t(r, (cout,)))
        x = _t(r, (2, cin) + hw)
        with torch.no_grad():
            ref = F.conv2d(x.permute(0, 1, 3, 2), conv.weight, conv.bias, 1, 1)
            got = S.fused_conv2d(conv.to(dev()), None, x.to(dev()), transposed_input=True)
        assert tuple(got.shape) == tuple(ref.shape)
        check(got.cpu().numpy(), ref.numpy(), TIGHT, f"conv(x^T) {cin}->{cout}")


def test_fused_conv2d_bias_and_whole_extent_layer():
    """Conv2d with its own bias (hm2, the coordinate head's last layer) and the (6,4) kernel that covers its whole input
    (vernier.py:87-88), run as a 1x1 layer over the flattened input, + Sigmoid."""
    from snvc_amd.models import submodule as S
    r = np.random.default_rng(1100)
    for cin, cout, ks, hw, pad in ((64, 9, (1, 1), (12, 20), 0), (32, 18, (6, 4), (6, 4), 0), (16, 18, (3, 3), (6, 4), 1)):
        conv = torch.nn.Conv2d(cin, cout, ks, 1, pad, bias=True)
        conv.weight.data.copy_(_t(r, tuple(conv.weight.shape)) * 0.1)
        conv.bias.data.copy_(_t(r, (cout,)))
        x = _t(r, (3, cin) + hw)
        with torch.no_grad():
            ref = F.conv2d(x, conv.weight, conv.bias, 1, pad)
            cd = conv.to(dev())
            check(S.fused_conv2d(cd, None, x.to(dev())).cpu().numpy(), ref.numpy(), TIGHT, f"conv+bias {ks}")
            check(S.fused_conv2d(cd, None, x.to(dev()), sigmoid=True).cpu().numpy(), torch.sigmoid(ref).numpy(), TIGHT,
                  f"sigmoid(conv+bias) {ks}")
            bn = seeded(torch.nn.BatchNorm2d(cout), 5).to(dev())
            exp = F.relu(_bn_eval(ref, bn.cpu()))
            check(S.fused_conv2d(cd, bn.to(dev()), x.to(dev()), relu=True).cpu().numpy(), exp.numpy(), TIGHT, f"relu(bn(conv+bias)) {ks}")


def test_zero_stuff_and_fused_deconv2d_vs_torch():
    from snvc_amd import ops
    from snvc_amd.models import submodule as S
    r = np.random.default_rng(1200)
    x = _t(r, (2, 5, 7, 9)).to(dev())
    up = ops.zero_stuff2x(x)
    exp = torch.zeros(2, 5, 14, 18, device=dev())
    exp[:, :, ::2, ::2] = x
    assert torch.equal(up, exp)
    for cin, cout, hw in ((64, 64, (6, 10)), (64, 32, (12, 17)), (8, 5, (1, 1)), (128, 64, (3, 2)), (64, 64, (64, 96)), (32, 48, (20, 44))):
        seq = seeded(S._deconvbn_2d(cin, cout, False), 31)
        x = _t(r, (2, cin) + hw)
        with torch.no_grad():
            raw = F.conv_transpose2d(x, seq[0].weight, None, 2, 1, 1)
            ref = _bn_eval(raw, seq[1])
            res = _t(r, tuple(ref.shape))
            sd = seq.to(dev())
            check(S.fused_deconv2d(sd[0], sd[1], x.to(dev())).cpu().numpy(), ref.numpy(), TIGHT, f"bn(deconv2d) {cin}->{cout}")
            check(S.fused_deconv2d(sd[0], None, x.to(dev())).cpu().numpy(), raw.numpy(), TIGHT, f"deconv2d {cin}->{cout}")
            check(S.fused_deconv2d(sd[0], sd[1], x.to(dev()), relu=True, residual=res.to(dev())).cpu().numpy(),
                  F.relu(ref + res).numpy(), TIGHT, f"relu(bn(deconv2d) + res) {cin}->{cout}")


def _routes():
    from snvc_amd.models import submodule as S
    return S._ROUTES["neck2d_hip"], S._ROUTES["neck2d_torch"]


def test_fused_conv2d_group_norm_vs_torch():
    """cfg.gn: Conv2d / ConvTranspose2d + GroupNorm(32, C) (+ residual, + ReLU) on the HIP kernels against torch."""
    from snvc_amd.models import submodule as S
    r = np.random.default_rng(1500)
    for tr, cin, cout, stride, hw in ((False, 32, 64, 1, (12, 20)), (False, 64, 64, 2, (16, 24)), (True, 64, 32, 2, (6, 10))):
        seq = S._deconvbn_2d(cin, cout, True) if tr else S.convbn(cin, cout, 3, stride, 1, 1, gn=True)
        seq[0].weight.data.copy_(_t(r, tuple(seq[0].weight.shape)) * 0.1)
        seq[1].weight.data.copy_(torch.from_numpy(r.uniform(0.5, 1.5, cout).astype(np.float32)))
        seq[1].bias.data.copy_(torch.from_numpy(r.uniform(-0.2, 0.2, cout).astype(np.float32)))
        x = _t(r, (2, cin) + hw)
        with torch.no_grad():
            ref = seq(x)
            res = _t(r, tuple(ref.shape))
            sd, xd, rd = seq.to(dev()), x.to(dev()), res.to(dev())
            f = S.fused_deconv2d if tr else S.fused_conv2d
            check(f(sd[0], sd[1], xd).cpu().numpy(), ref.numpy(), 5e-5, f"gn(conv) transposed={tr}")
            check(f(sd[0], sd[1], xd, relu=True, residual=rd).cpu().numpy(), F.relu(ref + res).numpy(), 5e-5,
                  f"relu(gn(conv) + res) transposed={tr}")


@pytest.mark.parametrize("block", ["basic", "basic_down", "hg2d", "hg2d_skips", "hg2d_16", "hg2d_gn", "hg2d_16_gn"])
def test_neck_blocks_vs_oracle_modules(block):
    """The blocks against the oracle's torch modules (pinned to the imported reference by make_golden.py) with the same
    seeded state dict; the HIP route must be the one taken under no_grad."""
    from oracle import torch_ref as T
    from snvc_amd.models import submodule as S
    r = np.random.default_rng(1300)
    if block == "basic":
        ours, ref, x, extra = S.BasicBlock2d(32, 32), T.BasicBlock2d(32, 32), _t(r, (2, 32, 12, 8)), ()
    elif block == "basic_down":
        ours = S.BasicBlock2d(16, 32, 2, S.basicdownsample(16, 32))
        ref = T.BasicBlock2d(16, 32, 2, T.basicdownsample(16, 32))
        x, extra = _t(r, (2, 16, 12, 8)), ()
    elif block == "hg2d_gn":
        ours, ref, x, extra = S.hourglass2d(32, gn=True), T.hourglass2d(32, gn=True), _t(r, (2, 32, 16, 24)), (None, None)
    elif block == "hg2d_16_gn":
        ours, ref, x, extra = (S.hourglass2d_downsample_16(32, gn=True), T.hourglass2d_downsample_16(32, gn=True),
                               _t(r, (2, 32, 32, 48)), ())
    elif block == "hg2d":
        ours, ref, x, extra = S.hourglass2d(32), T.hourglass2d(32), _t(r, (2, 32, 16, 24)), (None, None)
    elif block == "hg2d_skips":
        ours, ref, x = S.hourglass2d(32), T.hourglass2d(32), _t(r, (1, 32, 8, 12))
        extra = (_t(r, (1, 64, 4, 6)), _t(r, (1, 64, 4, 6)))
    else:
        ours, ref, x, extra = S.hourglass2d_downsample_16(32), T.hourglass2d_downsample_16(32), _t(r, (2, 32, 32, 48)), ()
    sd = T.seeded_state_dict(ref, 77)
    ref.load_state_dict(sd)
    ours.load_state_dict(sd, strict=True)
    ref.eval()
    ours.eval().to(dev())
    with torch.no_grad():
        exp = ref(x, *extra)
        hip0, torch0 = _routes()
        got = ours(x.to(dev()), *[e.to(dev()) if e is not None else None for e in extra])
        hip1, torch1 = _routes()
    assert hip1 > hip0 and torch1 == torch0, "the block must run on the HIP kernels under no_grad with eval BatchNorm"
    exp = exp if isinstance(exp, tuple) else (exp,)
    got = got if isinstance(got, tuple) else (got,)
    for i, (g, e) in enumerate(zip(got, exp)):
        check(g.cpu().numpy(), e.numpy(), 5e-5, f"{block} output {i}")


def test_neck_keeps_torch_route_when_anything_needs_grad():
    """eval-mode BatchNorm + an input that does not require grad + trainable head weights (head-only
    fine-tuning on a frozen trunk) must NOT take the kernels without a backward: the parameters get their gradients."""
    from snvc_amd.models import submodule as S
    r = np.random.default_rng(1400)
    blk = seeded(S.BasicBlock2d(16, 16), 3).to(dev())      # eval mode, parameters require grad
    x = _t(r, (1, 16, 8, 8)).to(dev())                      # detached input
    hip0, torch0 = _routes()
    y = blk(x)
    hip1, torch1 = _routes()
    assert (hip1, torch1) == (hip0, torch0 + 1)
    y.square().mean().backward()
    assert blk.conv1.weight.grad is not None and blk.conv2.weight.grad.abs().sum() > 0
    with torch.no_grad():                                   # same module, nothing to differentiate: HIP route, same values
        y2 = blk(x)
    assert _routes()[0] == hip1 + 1
    check(y2.cpu().numpy(), y.detach().cpu().numpy(), 5e-5, "HIP route vs torch route")
    for p in blk.parameters():                              # frozen parameters: HIP route even with autograd on
        p.requires_grad_(False)
    hip2 = _routes()[0]
    blk(x)
    assert _routes()[0] == hip2 + 1
