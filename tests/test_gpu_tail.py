"""The global stack's folded tail without storing `post` (r5; snvc_f16x3_deconv3d_tail_forward + snvc_deconv_tail_gather).

Reference graph: hourglass conv5 / conv6 (snvc/models/submodule.py:127-146,161-166) and the composition
``classifier(v + hourglass(v)[0])`` of snvc/models/vernier.py:366-371.  conv6 has no activation, so
``classifier(bn(conv6(post)) + v) = deconv'(post) + b' + classifier(v)`` with deconv' a transposed layer to ONE channel.
Here conv5's epilogue contracts its own result with deconv's 27 taps per voxel (T), and a gather sums T over the
(voxel, tap) pairs of each output.  Checked against torch's float64 ``conv_transpose3d`` of the float64 layer.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_parity import TIGHT, check

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(2, 3, 5), (3, 4, 33), (5, 2, 64), (1, 1, 1)])
@pytest.mark.parametrize("n", [1, 2])
def test_tail_gather_is_the_scatter_half_of_a_one_channel_transposed_layer(shape, n):
    """T formed on the host in float64 from a random `post`; the gather must reproduce conv_transpose3d(post, W') + b + res."""
    from snvc_amd import ops
    g = np.random.default_rng(1)
    nd, nh, nw = shape
    c = 8
    post = g.standard_normal((n, c, 2 * nd, 2 * nh, 2 * nw))
    w = g.standard_normal((c, 1, 3, 3, 3))
    res = g.standard_normal((n, 1, 4 * nd, 4 * nh, 4 * nw)).astype(np.float32)
    b = np.float32(0.37)
    ref = F.conv_transpose3d(torch.from_numpy(post), torch.from_numpy(w), None, 2, 1, 1).numpy() + b + res
    t = np.einsum("ck,ncdhw->nkdhw", w.reshape(c, 27), post)                       # [n, 27, D, H, W] on post's grid
    t = t.reshape(n, 27, nd, 2, nh, 2, nw, 2).transpose(0, 1, 3, 5, 7, 2, 4, 6)      # class-major [n, 27, rd, rh, rw, pd, ph, pw]
    t = np.ascontiguousarray(t.reshape(n, 27, 8, nd, nh, nw)).astype(np.float32)
    got = ops.deconv_tail_gather(torch.from_numpy(t).to(dev()), torch.tensor([b], device=dev()), torch.from_numpy(res).to(dev()))
    check(got.cpu().numpy(), ref.astype(np.float32), 1e-6, f"tail gather {shape} x{n}")
    got = ops.deconv_tail_gather(torch.from_numpy(t).to(dev()))
    check(got.cpu().numpy(), (ref - b - res).astype(np.float32), 1e-6, f"tail gather {shape} x{n}, no bias / residual")


@pytest.mark.parametrize("form", ["full", "small"])
@pytest.mark.parametrize("case", ["64_64", "64_64_ragged", "64_32", "32_64_odd"])
def test_split_transposed_layer_with_tail_projection_vs_float64(case, form):
    """conv5 (+ folded BatchNorm + pre, ReLU) with its result contracted in the epilogue, then the gather: against the float64
    evaluation of relu(bn(deconv(x)) + pre) followed by the float64 one-channel transposed layer, at the exact-fp32 tolerance;
    and T itself against the to_f32 form of the same layer (what r4 stored) contracted in float64."""
    from snvc_amd import ops
    torch.manual_seed(len(case))
    cin, cout, shape = {"64_64": (64, 64, (4, 4, 32)), "64_64_ragged": (64, 64, (5, 7, 45)), "64_32": (64, 32, (3, 5, 34)),
                        "32_64_odd": (32, 64, (2, 3, 7))}[case]
    n = 2
    x = torch.relu(torch.randn(n, cin, *shape, device=dev())) * 1.5
    w = torch.randn(cin, cout, 3, 3, 3, device=dev()) * np.sqrt(2.0 / (cin * 27 / 8))
    scale, bias = torch.rand(cout, device=dev()) + 0.5, torch.randn(cout, device=dev()) * 0.3
    out_sp = tuple(2 * s for s in shape)
    pre = torch.relu(torch.randn(n, cout, *out_sp, device=dev()))
    wt = torch.randn(cout, 1, 3, 3, 3, device=dev()) * 0.2
    from snvc_amd import _lib as L_
    layer = ops.Conv3dLayerX3(w, 3, 2, 1, 1, True, algo=L_.ALGO_X3_SMALL if form == "small" else 0)      # 2x4x32 / 4x4x32 input tiles
    tail = ops.TailWeightsX3(wt)
    x_exp, e_y, e_res = 3, 2, 4
    xs, rs = ops.to_split(x, x_exp), ops.to_split(pre, e_res)
    flag = torch.zeros(1, dtype=torch.int32, device=dev())
    t = layer.forward_tail(xs, x_exp, scale, bias, tail, residual=rs, res_exp=e_res, flags=ops.EPI_RELU | ops.EPI_ADD_PRE, out_exp=e_y,
                           overflow=flag)
    assert int(flag.item()) == 0 and t.shape == (n, 27, 8) + shape
    hres = torch.randn(n, 1, *(4 * s for s in shape), device=dev())
    tb = torch.tensor([0.25], device=dev())
    got = ops.deconv_tail_gather(t, tb, hres)
    post64 = torch.relu(F.conv_transpose3d(x.double().cpu(), w.double().cpu(), None, 2, 1, 1) * scale.double().cpu().view(1, -1, 1, 1, 1)
                        + bias.double().cpu().view(1, -1, 1, 1, 1) + pre.double().cpu())
    ref = F.conv_transpose3d(post64, wt.double().cpu(), None, 2, 1, 1) + 0.25 + hres.double().cpu()
    check(got.cpu().numpy(), ref.numpy(), TIGHT, f"{case}: fused tail vs float64")
    # T against the stored form of the same layer
    post = layer(xs, x_exp, scale, bias, residual=rs, res_exp=e_res, flags=ops.EPI_RELU | ops.EPI_ADD_PRE, out_exp=e_y, to_f32=True)
    check(post.cpu().numpy(), post64.numpy(), TIGHT, f"{case}: the layer itself (fp32 output)")
    tref = torch.einsum("ck,ncdhw->nkdhw", wt.double().cpu().reshape(cout, 27), post.double().cpu())
    d_, h_, w_ = shape
    tref = tref.reshape(n, 27, d_, 2, h_, 2, w_, 2).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(n, 27, 8, d_, h_, w_)
    check(t.cpu().numpy(), tref.numpy(), TIGHT, f"{case}: T vs the stored layer contracted in float64")
    # a result beyond half's range is clamped AND flagged (the guard's last look is behind this launch)
    layer.forward_tail(xs, x_exp, scale * 4096.0, bias, tail, residual=rs, res_exp=e_res, flags=ops.EPI_RELU | ops.EPI_ADD_PRE, out_exp=8,
                       overflow=flag)
    assert int(flag.item()) == 1


def test_global_stack_fused_tail_equals_the_two_launch_tail():
    import bench
    from snvc_amd.models import submodule as S
    from snvc_amd.models.stereo_volume import GlobalStack
    m = GlobalStack(32)
    m.load_state_dict(bench.seeded_state(m, 5))
    m.eval().to(dev())
    g = np.random.default_rng(2)
    left = torch.from_numpy(g.standard_normal((2, 32, 12, 44)).astype(np.float32)).to(dev())
    right = torch.from_numpy(g.standard_normal((2, 32, 12, 44)).astype(np.float32)).to(dev())
    shift = torch.from_numpy(np.tile(np.arange(20, dtype=np.float32) * 0.5, (2, 1))).to(dev())
    with torch.no_grad():
        before = S._ROUTES["x3_fused_tail"]
        a = m.forward_pair(left, right, shift, 1)
        assert S._ROUTES["x3_fused_tail"] == before + 1
        m.fused_tail = False
        b = m.forward_pair(left, right, shift, 1)
        assert S._ROUTES["x3_fused_tail"] == before + 1
        c = m.forward_pair(left, right, shift, 1, arithmetic="fp32")
    check(a.cpu().numpy(), b.cpu().numpy(), 2e-6, "fused tail vs conv5 -> fp32 post -> one-channel transposed layer")
    check(a.cpu().numpy(), c.cpu().numpy(), 1e-4, "fused tail vs the fp32-MFMA stack")
