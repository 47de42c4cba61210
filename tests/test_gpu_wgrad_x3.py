"""r6: the split-operand (f16x3) weight gradients (csrc/conv3d_bwd.hip: conv3d_wgrad_x3_kernel, conv3d_wgrad_x3s2_kernel) with the
operands' maxima SUPPLIED (snvc_conv3d_wgrad_amax), as the training step runs them: the maximum left by the pass that wrote the
tensor (snvc_affine_act_amax / snvc_act_backward_apply_amax), an upper bound of it, and the layer-level plumbing (tags that an
in-place update voids).  Reference: torch autograd through nn.Conv3d in float64 (snvc/models/submodule.py:32-50 composes them).
tests/test_gpu_parity.py::test_conv3d_wgrad_vs_float64 covers the same kernels finding the maxima themselves."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_parity import check, dev

pytestmark = pytest.mark.gpu


def _ref(x, g, stride):
    w = torch.zeros(g.shape[1], x.shape[1], 3, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv3d(x.double(), w, stride=stride, padding=1)
    assert y.shape == g.shape
    (y * g.double()).sum().backward()
    return w.grad.float().numpy()


def _word(value, device):
    from snvc_amd import ops
    w = ops.amax_word(device)
    w[7:8] = torch.tensor([value], dtype=torch.float32, device=device).view(torch.int32)     # any slot: the consumer takes the maximum
    return w


@pytest.mark.parametrize("stride,shape", [(1, (2, 40, 24, (5, 10, 44))), (1, (1, 64, 64, (6, 10, 78))), (2, (1, 32, 64, (6, 12, 40))),
                                          (2, (1, 64, 64, (8, 6, 156)))])
@pytest.mark.parametrize("slack", [1.0, 1000.0])
def test_wgrad_with_supplied_maxima(stride, shape, slack):
    """exact maxima, and upper bounds 2^10 too large (10 of the 39 bits below the maximum: still fp32-grade)"""
    from snvc_amd import ops
    n, ci, co, sp = shape
    r = np.random.default_rng(17)
    x = torch.from_numpy((r.standard_normal((n, ci) + sp) * 3.0).astype(np.float32))
    g = torch.from_numpy((r.standard_normal((n, co) + tuple(s // stride for s in sp)) * 1e-5).astype(np.float32))
    exp = _ref(x, g, stride)
    xd, gd = x.to(dev()), g.to(dev())
    ax, ag = _word(float(x.abs().max()) * slack, dev()), _word(float(g.abs().max()) * slack, dev())
    dw = ops.conv3d_wgrad(xd, gd, 3, stride, 1, 1, amax_x=ax, amax_g=ag)
    dw_own = ops.conv3d_wgrad(xd, gd, 3, stride, 1, 1)
    check(dw.cpu().numpy(), exp, 2e-5, f"wgrad s{stride} supplied maxima x{slack:g}")
    if slack == 1.0:
        assert torch.equal(dw, dw_own), "the same scales -> the same bits"
    # only one of the two supplied
    dw_half = ops.conv3d_wgrad(xd, gd, 3, stride, 1, 1, amax_x=ax)
    check(dw_half.cpu().numpy(), exp, 2e-5, "one maximum supplied")


def test_producer_passes_leave_the_maximum():
    """snvc_affine_act_amax / snvc_act_backward_apply_amax: the maximum over the slots is max|output|, bit for bit"""
    from snvc_amd import ops
    r = np.random.default_rng(3)
    raw = torch.from_numpy(r.standard_normal((2, 32, 4, 6, 40)).astype(np.float32)).to(dev())
    sc = torch.from_numpy(r.uniform(0.5, 2, 32).astype(np.float32)).to(dev())
    sh = torch.from_numpy(r.uniform(-1, 1, 32).astype(np.float32)).to(dev())
    w = ops.amax_word(dev())
    y = ops.affine_act(raw, sc, sh, None, ops.EPI_RELU, amax=w)
    assert int(w.max()) == int(y.abs().max().view(torch.int32)) and int((w != 0).sum()) > 1
    gy = torch.from_numpy(r.standard_normal(tuple(raw.shape)).astype(np.float32)).to(dev())
    cg, cr, cc = (torch.from_numpy(r.standard_normal(32).astype(np.float32)).to(dev()) for _ in range(3))
    w2 = ops.amax_word(dev())
    draw, _ = ops.act_backward_apply(raw, gy, None, sc, sh, cg, cr, cc, ops.EPI_RELU, False, False, amax=w2)
    assert int(w2.max()) == int(draw.abs().max().view(torch.int32))
    # rows that are not 16-byte aligned take the scalar loop of both passes
    raw3 = raw[..., :39].contiguous()
    w3 = ops.amax_word(dev())
    y3 = ops.affine_act(raw3, sc, sh, None, ops.EPI_RELU, amax=w3)
    assert int(w3.max()) == int(y3.abs().max().view(torch.int32))


def test_layer_backward_uses_tags_and_matches_the_fp32_forms():
    """Two stacked conv + BatchNorm + ReLU layers under autograd: the second layer's weight gradient takes x's maximum from the tag the first
    layer's pass left and g's from its own epilogue backward; same gradients as with the fp32 weight-gradient forms within the layer
    tolerances; a tensor written to in place loses its tag."""
    from snvc_amd import _lib, ops
    from snvc_amd.models import submodule as S
    torch.manual_seed(0)
    # (frozen BatchNorm statistics: behind a TRAIN-mode BatchNorm the convolution's weight gradient is a difference of nearly equal
    # terms -- the norm removes what a change of scale would do -- and any two summation orders differ by 1e-3 of its 1e-7 size)
    l1, l2 = S.convbn_3d(32, 32, 3, 1, 1).to(dev()).eval(), S.convbn_3d(32, 32, 3, 1, 1).to(dev()).eval()
    x = torch.randn(1, 32, 6, 8, 40, device=dev(), requires_grad=True)

    def run(bits):
        for p in list(l1.parameters()) + list(l2.parameters()):
            p.grad = None
        with ops.conv_variant(bits):
            y1 = l1(x)
            tagged = ops.amax_of(y1) is not None
            out = l2(y1)
            out.square().mean().backward()
        return tagged, [p.grad.clone() for p in list(l1.parameters()) + list(l2.parameters())]
    tagged, g_x3 = run(0)
    assert tagged, "the forward pass of a training layer tags its output with the maximum"
    _, g_f32 = run(_lib.ALGO_WGRAD_FP32)
    for a, b in zip(g_x3, g_f32):
        check(a.cpu().numpy(), b.cpu().numpy(), 5e-5, "gradients, split-operand vs fp32 weight-gradient forms")
    y1 = l1(x)
    assert ops.amax_of(y1) is not None
    with torch.no_grad():
        y1.mul_(2.0)
    assert ops.amax_of(y1) is None, "an in-place update voids the tag"
