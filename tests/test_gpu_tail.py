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

from test_gpu_parity import TIGHT, check  # noqa: E402

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


# ------------------------------------------------------------------------------------------------------------------------------
# r5: the sheared first layer's depth-1 layers in split mode (snvc_f16x3_split_scale, snvc_sheared_upsample_split, snvc_f16x3_conv2d_*)
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 3, 4, 1023, 4096 * 37 + 2, 3 * 1000 * 1000])
def test_split_scale_in_one_launch_equals_the_torch_expression(n):
    from snvc_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n)
    for amp in (1.0, 3.7e-5, 9.1e6):
        x = (torch.randn(n, generator=g) * amp).to(dev())
        a = ops.split_scale_of(x)
        # r4's torch expression (split_scale_for's fallback path), spelled out: the one-launch kernel must agree bit for bit
        lo, hi = torch.aminmax(x)
        e = torch.floor(torch.log2(16384.0 / torch.maximum(-lo, hi).float().reshape(1)))
        b = torch.exp2(torch.where(torch.isfinite(e), e, torch.zeros_like(e)).clamp_(-24, 40))
        assert torch.equal(a, b), (n, amp, a.item(), b.item())
        assert torch.equal(ops.split_scale_for(x), a)                  # contiguous fp32: the same launch
        y = (x * 0.25)[: max(n // 2, 1)]
        assert torch.equal(ops.split_scale_for(x, y), a)               # several tensors: the minimum of their scales
        xs = x[::2] if n > 1 else x                                     # not contiguous: the torch expression
        assert ops.split_scale_for(xs).item() * xs.abs().max().item() < 16384.0
        m = float(a.item()) * x.abs().max().item()
        assert 8192.0 <= m < 16384.0
    z = torch.zeros(n, device=dev())
    assert ops.split_scale_of(z).item() == 1.0                     # all zero: scale 1
    z[n // 2] = float("inf")
    assert ops.split_scale_of(z).item() == 1.0                     # non-finite maximum: scale 1
    z[n // 2] = float("nan")
    z[0] = -5.0
    assert ops.split_scale_of(z).item() == 2048.0                  # a NaN does not take part: max|x| = 5 -> 5 * 2048 = 10240
    # (an exact power of two lands ON 2^14: floor(log2(16384 / 2)) = 13, as in the torch expression); the scratch words were left zero
    assert ops.split_scale_of(torch.full((n,), 2.0, device=dev())).item() == 8192.0


@pytest.mark.parametrize("q", [1, 2])
def test_sheared_upsample_split_equals_upsample_then_split(q):
    from snvc_amd import ops
    torch.manual_seed(q)
    right = torch.randn(2, 32, 9, 40, device=dev()) * 3.0
    mul = ops.split_scale_of(right)
    for wu, off in ((q * 39 + 12, 4), (56, -11)):
        a = ops.sheared_upsample_split(right, q, wu, off, mul)
        b = ops.to_split(ops.sheared_upsample(right, q, wu, off).unsqueeze(2), mul_dev=mul)
        assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("case", ["3x7_32_96", "3x7_8_32_ragged", "3x3_32_96", "3x3_64_64_tiny"])
def test_depth1_split_layer_vs_float64(case):
    from snvc_amd import ops
    torch.manual_seed(len(case))
    kh, kw, cin, cout, shape = {"3x7_32_96": (3, 7, 32, 96, (96, 200)), "3x7_8_32_ragged": (3, 7, 8, 32, (17, 45)),
                                "3x3_32_96": (3, 3, 32, 96, (40, 70)), "3x3_64_64_tiny": (3, 3, 64, 64, (2, 3))}[case]
    n = 2
    x = torch.randn(n, cin, *shape, device=dev()) * 2.5
    w = torch.randn(cout, cin, kh, kw, device=dev()) * np.sqrt(2.0 / (cin * kh * kw))
    lay = ops.Conv2dLayerX3(w)
    mul = ops.split_scale_of(x)
    xs = ops.to_split(x.unsqueeze(2), mul_dev=mul)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), None, 1, (kh // 2, kw // 2))
    got = lay(xs, mul)
    check(got.cpu().numpy(), ref.numpy(), TIGHT, f"{case}: depth-1 split layer vs float64")
    scale, bias = torch.rand(cout, device=dev()) + 0.5, torch.randn(cout, device=dev())
    got = lay(xs, mul, scale, bias, flags=ops.EPI_RELU)
    ref2 = torch.relu(ref * scale.double().cpu().view(1, -1, 1, 1) + bias.double().cpu().view(1, -1, 1, 1))
    check(got.cpu().numpy(), ref2.numpy(), TIGHT, f"{case}: + affine + ReLU")
    xs5 = ops.to_split(x.unsqueeze(2), 5)                              # a fixed exponent instead of the device scale
    check(lay(xs5, None, x_exp=5).cpu().numpy(), ref.numpy(), TIGHT, f"{case}: fixed exponent")


def test_global_stack_split_prep_equals_fp32_prep():
    import bench
    from snvc_amd.models import submodule as S
    from snvc_amd.models.stereo_volume import GlobalStack
    m = GlobalStack(32)
    m.load_state_dict(bench.seeded_state(m, 9))
    m.eval().to(dev())
    g = np.random.default_rng(4)
    left = torch.from_numpy(g.standard_normal((2, 32, 12, 44)).astype(np.float32)).to(dev())
    right = torch.from_numpy((5.0 * g.standard_normal((2, 32, 12, 44))).astype(np.float32)).to(dev())
    for q in (1, 2):
        shift = torch.from_numpy(np.tile(np.arange(20, dtype=np.float32) / q, (2, 1))).to(dev())
        with torch.no_grad():
            before = S._ROUTES["sheared_prep_x3"]
            a = m.forward_pair(left, right, shift, 1)
            v1a = m.last_first_layer()
            after = S._ROUTES["sheared_prep_x3"]
            assert after >= before + 1          # (+2 when the spacing guessed from the previous call was wrong and the prep ran twice)
            m.split_prep = False
            b = m.forward_pair(left, right, shift, 1)
            v1b = m.last_first_layer()
            assert S._ROUTES["sheared_prep_x3"] == after
            m.split_prep = True
            c = m.forward_pair(left, right, shift, 1, arithmetic="fp32")      # fp32 arithmetic: the fp32-MFMA prep
            assert S._ROUTES["sheared_prep_x3"] == after
        check(v1a.cpu().numpy(), v1b.cpu().numpy(), TIGHT, f"q={q}: first layer, split prep vs fp32 prep")
        check(a.cpu().numpy(), b.cpu().numpy(), 1e-5, f"q={q}: stack output, split prep vs fp32 prep")
        check(a.cpu().numpy(), c.cpu().numpy(), 1e-4, f"q={q}: vs the fp32-MFMA stack")
    # any shift array (warp after convolution): its three depth-1 3x3 layers P, Q, E and the left planes in split mode
    shift = torch.from_numpy(np.tile(np.arange(20, dtype=np.float32) * 0.73 + 0.2, (2, 1))).to(dev())
    with torch.no_grad():
        before = S._ROUTES["commuted_prep_x3"]
        a = m.forward_pair(left, right, shift, 1)
        v1a = m.last_first_layer()
        assert S._ROUTES["commuted_prep_x3"] == before + 1
        m.split_prep = False
        b = m.forward_pair(left, right, shift, 1)
        v1b = m.last_first_layer()
        assert S._ROUTES["commuted_prep_x3"] == before + 1
        m.split_prep = True
    check(v1a.cpu().numpy(), v1b.cpu().numpy(), TIGHT, "any shift: first layer, split prep vs fp32 prep")
    check(a.cpu().numpy(), b.cpu().numpy(), 1e-5, "any shift: stack output, split prep vs fp32 prep")


@pytest.mark.parametrize("res_mode", ["none", "pre", "post"])
def test_affine_act_split_vs_torch(res_mode):
    """snvc_f16x3_affine_from_ncdhw (the norm + residual + activation pass of a GroupNorm layer in split mode): per-(sample, channel)
    affine of an fp32 NCDHW tensor, residual a split pair in its own units, ReLU, result as a split pair; clamp + flag."""
    from snvc_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    n, c, sp = 2, 44, (3, 5, 37)          # 44 channels: a ragged last group (channels 44..47 do not exist)
    raw = torch.randn(n, c, *sp, generator=g).to(d) * 3
    sc = (torch.rand(n, c, generator=g) + 0.5).to(d)
    sh = torch.randn(n, c, generator=g).to(d)
    res = torch.randn(n, c, *sp, generator=g).to(d) * 2
    res_s = ops.to_split(res, 3)
    res_v = ops.from_split(res_s, 3, c)                     # what the pair holds (22 bits)
    flags = ops.EPI_RELU | {"none": 0, "pre": ops.EPI_ADD_PRE, "post": ops.EPI_ADD_POST}[res_mode]
    flag = torch.zeros(1, dtype=torch.int32, device=d)
    y = ops.affine_act_split(raw, sc, sh, 2, residual=None if res_mode == "none" else res_s, res_exp=3, flags=flags, overflow=flag)
    v = raw * sc[:, :, None, None, None] + sh[:, :, None, None, None]
    if res_mode == "pre":
        v = v + res_v
    v = torch.relu(v)
    if res_mode == "post":
        v = v + res_v
    got = ops.from_split(y, 2, c)
    err = (got - v).abs().max().item() / v.abs().max().item()
    assert err < 2e-6, err
    assert flag.item() == 0
    assert torch.isfinite(y.float()).all()
    assert torch.all(y[:, :, 5, ..., 4:8].float() == 0)      # channels 44..47 of the ragged group are zeros
    # clamp + flag: a scale that pushes values out of half's range
    y2 = ops.affine_act_split(raw * 1e4, sc, sh, 4, flags=0, overflow=flag)
    assert flag.item() == 1 and torch.isfinite(y2.float()).all()
