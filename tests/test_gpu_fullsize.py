"""Parity at BASELINE.json's FULL sizes (cfg2: volume [1,64,192,96,312]) through size-independent
properties and spot checks against the oracle on crops: the small-size parity tests cannot catch index
overflow, tile-edge or grid-limit mistakes that only appear at scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
C, H, W, D = 32, 96, 312, 192


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def pair():
    r = np.random.default_rng(2024)
    left = r.standard_normal((1, C, H, W)).astype(np.float32)
    right = r.standard_normal((1, C, H, W)).astype(np.float32)
    shift = np.linspace(0.0, 95.5, D, dtype=np.float32)[None].copy()
    return left, right, shift


def test_cost_volume_full_size(pair):
    from oracle import native as O
    from snvc_amd import ops
    from snvc_amd.extension.build_cost_volume import build_cost_volume
    left, right, shift = pair
    dl, dr, ds = (torch.from_numpy(a).to(dev()) for a in pair)
    vol = build_cost_volume(dl, dr, ds, 1)
    assert vol.shape == (1, 2 * C, D, H, W)
    # left half: every plane is the left feature (checksum of checksums + sampled planes, bit-exact)
    lsum = vol[0, :C].double().sum(dim=(2, 3))                      # [C, D]
    assert torch.equal(lsum, dl[0].double().sum(dim=(1, 2))[:, None].expand(C, D))
    for c, d in ((0, 0), (17, 95), (31, 191)):
        assert torch.equal(vol[0, c, d], dl[0, c])
    # right half: sampled channels against the C oracle run on that single channel (bit-exact)
    for c in (0, 13, 31):
        exp = O.cost_volume_forward(left[:, c:c + 1], right[:, c:c + 1], shift, 1)[0, 1]    # [D,H,W]
        assert np.array_equal(vol[0, C + c].cpu().numpy(), exp)
    # the right-half builder equals the full builder's right half
    assert torch.equal(ops.cost_volume_forward_right(dr, ds), vol[:, C:])
    # backward: adjoint identity <fwd(L,R), g> == <L, gL> + <R, gR> at full size
    g = torch.randn_like(vol)
    from snvc_amd.extension.build_cost_volume import build_cost_volume_cuda as CV
    gl, gr = CV.build_cost_volume_backward(g, ds, 1)
    lhs = (vol.double() * g.double()).sum().item()
    rhs = (dl.double() * gl.double()).sum().item() + (dr.double() * gr.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0) + 1e-2
    # and sampled channels of the backward against the oracle (bit-exact, deterministic)
    gsub = torch.cat([g[:, 5:6], g[:, C + 5:C + 6]], dim=1).cpu().numpy()
    el, er = O.cost_volume_backward(gsub, shift, 1)
    assert np.array_equal(gl[0, 5].cpu().numpy(), el[0, 0]) and np.array_equal(gr[0, 5].cpu().numpy(), er[0, 0])


def test_conv3d_full_size_crops_and_linearity():
    """First conv at cfg2 size: spot voxels against torch-CPU on input crops (corners, tile edges, the
    last partial 32-wide tile at W = 312) and linearity of the bare convolution."""
    from snvc_amd.models import submodule as S
    conv = S.HipConv3d(2 * C, C, 3, 1, 1, bias=False)
    torch.manual_seed(1)
    torch.nn.init.normal_(conv.weight, std=0.05)
    conv = conv.to(dev())
    x = torch.randn(1, 2 * C, D, H, W, device=dev())
    with torch.no_grad():
        y = conv(x)
        w_cpu = conv.weight.cpu()
        pts = [(0, 0, 0), (D - 1, H - 1, W - 1), (3, 4, 31), (4, 3, 32), (100, 50, 287), (100, 50, 288), (191, 0, 311),
               (7, 95, 160)]
        for (d, h, w) in pts:
            d0, d1, h0, h1, w0, w1 = max(d - 1, 0), min(d + 2, D), max(h - 1, 0), min(h + 2, H), max(w - 1, 0), min(w + 2, W)
            crop = x[:, :, d0:d1, h0:h1, w0:w1].cpu()
            pad = (1 - (w - w0), 1 - (w1 - 1 - w), 1 - (h - h0), 1 - (h1 - 1 - h), 1 - (d - d0), 1 - (d1 - 1 - d))
            ref = F.conv3d(F.pad(crop, pad), w_cpu)[0, :, 0, 0, 0]
            got = y[0, :, d, h, w].cpu()
            assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4), (d, h, w, (got - ref).abs().max())
        x2 = torch.randn_like(x)
        lin = conv(0.5 * x - 2.0 * x2)
        comb = 0.5 * y - 2.0 * conv(x2)
        err = (lin - comb).abs().max().item() / comb.abs().max().item()
        assert err < 1e-5, err


def test_global_pair_full_size_factored_equals_materialised(pair):
    """The benchmarked step at full size: the sheared first convolution (uniform disparity planes) and the general factored
    first convolution, both against the materialised concat volume; the first layer's output of the two compared directly."""
    from snvc_amd.models.stereo_volume import GlobalStack
    import bench
    m = GlobalStack(C)
    m.load_state_dict(bench.seeded_state(m))
    m.eval().to(dev())
    dl, dr, ds = (torch.from_numpy(a).to(dev()) for a in pair)
    from snvc_amd.models import submodule as S
    with torch.no_grad():
        before = S._ROUTES["sheared_first_conv"]
        s = m.forward_pair(dl, dr, ds, 1)                           # cfg2's shifts are d / 2: the sheared first convolution
        assert S._ROUTES["sheared_first_conv"] == before + 1
        v1s = m.last_first_layer()
        probe_s = v1s[0, ::7, ::5, ::9, ::11].clone()               # the first layer itself, strided over the whole volume,
        edge_s = [v1s[0, :, 0].clone(), v1s[0, :, D - 1].clone(), v1s[0, :, :, :, W - 1].clone(), v1s[0, :, :, :, 0].clone()]
        a = m.forward_pair(dl, dr, ds, 1, factored=True, sheared=False, commuted=False)
        v1g = m.last_first_layer()
        assert torch.allclose(probe_s, v1g[0, ::7, ::5, ::9, ::11], rtol=1e-4, atol=1e-4)
        for e, g in zip(edge_s, (v1g[0, :, 0], v1g[0, :, D - 1], v1g[0, :, :, :, W - 1], v1g[0, :, :, :, 0])):   # and its four borders whole
            assert torch.allclose(e, g, rtol=1e-4, atol=1e-4), (e - g).abs().max()   # 864-term fp32 sums in two orders, values up to ~5
        probe_g = v1g[0, ::7, ::5, ::9, ::11].clone()
        edge_g = [v1g[0, :, 0].clone(), v1g[0, :, D - 1].clone(), v1g[0, :, :, :, W - 1].clone(), v1g[0, :, :, :, 0].clone()]
        # the form any OTHER shift array takes (warp after convolution), here on cfg2's own shifts at full size
        before_c = S._ROUTES["commuted_first_conv"]
        cmt = m.forward_pair(dl, dr, ds, 1, sheared=False)
        assert S._ROUTES["commuted_first_conv"] == before_c + 1
        v1c = m.last_first_layer()
        assert torch.allclose(probe_g, v1c[0, ::7, ::5, ::9, ::11], rtol=1e-4, atol=1e-4)
        for e, g in zip(edge_g, (v1c[0, :, 0], v1c[0, :, D - 1], v1c[0, :, :, :, W - 1], v1c[0, :, :, :, 0])):
            assert torch.allclose(e, g, rtol=1e-4, atol=1e-4), (e - g).abs().max()
        b = m.forward_pair(dl, dr, ds, 1, factored=False)
    assert a.shape == (1, 1, D, H, W) and torch.isfinite(a).all()
    err = (cmt - b).abs().max().item() / b.abs().max().item()
    assert err < 1e-4, err
    err = (a - b).abs().max().item() / b.abs().max().item()
    assert err < 1e-4, err
    err = (s - b).abs().max().item() / b.abs().max().item()
    assert err < 1e-4, err


# ------------------------------------------------------------------------------------------------
# Local (V-A) model at the released shape (32,128,192) and at a cfg3 crop (96,96,96): every layer kind
# of trunk_3d checked at FULL size -- spot voxels against torch-CPU on input crops (so a wrong tile, a
# 32-bit offset overflow or a grid-limit mistake cannot hide), Winograd == direct kernel over the whole
# tensor, and the fused trunk against its own layer-by-layer evaluation.
# ------------------------------------------------------------------------------------------------
def _local_model(grid, F_=32):
    import types
    import bench
    from snvc_amd.models.vernier import VernierScale
    cfg = types.SimpleNamespace(vernier_type="BEV_type3", backbone="hrfeat", gn=False,
                                grid_resolution=[32, grid[1], 192], resolution=(256, 256),
                                x_range=(-1.0, 1.0), z_range=(-1.0, 1.0), num_parts=9)
    cfg.hrfeat = types.SimpleNamespace(output_channel=F_, name="identity")
    cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
    m = VernierScale(cfg)
    m.load_state_dict(bench.seeded_state(m))
    return m.eval().to(dev())


def _spot_points(shape, tile=(4, 4, 32)):
    """Corners, tile edges (last voxel of a tile / first of the next), the far corner and a few interior voxels."""
    d, h, w = shape
    pts = {(0, 0, 0), (d - 1, h - 1, w - 1), (0, h - 1, 0), (d - 1, 0, w - 1), (d // 2, h // 2, w // 2)}
    pts |= {(tile[0] - 1, tile[1] - 1, tile[2] - 1), (tile[0], tile[1], tile[2]), (d - 1, h // 2, tile[2] * (w // tile[2]) - 1),
            (d // 3, h - 1, w - tile[2]), (1, 2, w - 2), (d - 2, 1, 3)}
    return [p for p in pts if all(0 <= c < s for c, s in zip(p, shape))]


def _conv_at(x, weight, pt, k, dil, stride=1):
    """torch-CPU value of conv3d(x, weight)[0, :, pt] from the input crop around pt ('same' padding)."""
    r = dil * (k - 1) // 2
    shape = x.shape[2:]
    lo = [p * stride - r for p in pt]
    hi = [l + 2 * r + 1 for l in lo]
    sl = [slice(max(l, 0), min(h_, s)) for l, h_, s in zip(lo, hi, shape)]
    crop = x[:, :, sl[0], sl[1], sl[2]].cpu()
    pad = []
    for l, h_, s in reversed(list(zip(lo, hi, shape))):   # F.pad wants (W_lo, W_hi, H_lo, H_hi, D_lo, D_hi)
        pad += [max(-l, 0), max(h_ - s, 0)]
    return F.conv3d(F.pad(crop, pad), weight, None, 1, 0, dil)[0, :, 0, 0, 0]


def _check_layer_spots(seq, x, y, relu=True, residual=None, after_act=False, sigmoid=False, tol=2e-4, what=""):
    """seq: ConvBN3d (conv + eval BatchNorm) or a bare conv; y = the HIP result for input x."""
    conv, bn = (seq[0], seq[1]) if isinstance(seq, torch.nn.Sequential) else (seq, None)
    k, dil = conv.kernel_size[0], conv.dilation[0]
    w_cpu = conv.weight.detach().cpu()
    for pt in _spot_points(tuple(y.shape[2:])):
        ref = _conv_at(x, w_cpu, pt, k, dil)
        if bn is not None:
            ref = (ref - bn.running_mean.cpu()) / torch.sqrt(bn.running_var.cpu() + bn.eps) * bn.weight.detach().cpu() + bn.bias.detach().cpu()
        res = residual[0, :, pt[0], pt[1], pt[2]].cpu() if residual is not None else None
        if res is not None and not after_act:
            ref = ref + res
        if relu:
            ref = torch.relu(ref)
        if sigmoid:
            ref = torch.sigmoid(ref)
        if res is not None and after_act:
            ref = ref + res
        got = y[0, :, pt[0], pt[1], pt[2]].cpu()
        scale = max(ref.abs().max().item(), 1.0)
        assert (got - ref).abs().max().item() <= tol * scale, (what, pt, (got - ref).abs().max().item(), scale)


@pytest.mark.parametrize("grid", [(32, 128, 192), (96, 96, 96)], ids=["released_32x128x192", "cfg3_crop_96"])
def test_local_trunk_full_size(grid):
    import bench
    from snvc_amd import ops
    from snvc_amd.models.submodule import _folded_bn, _get_layer, _Plan
    m = _local_model(grid)
    m.precision = "f32"          # this test walks the fp32-MFMA kernels layer by layer; the split-mode trunk has its own test below
    f = 32
    r = np.random.default_rng(7)
    lf = torch.from_numpy(r.standard_normal((1, f, 64, 64)).astype(np.float32)).to(dev())
    rf = torch.from_numpy(r.standard_normal((1, f, 64, 64)).astype(np.float32)).to(dev())
    pl, pr = bench.projected_coordinates(1, grid, dev())
    v = grid[0] * grid[1] * grid[2]
    with torch.no_grad():
        # a3 at full size: the projected coordinates land inside the crop; gather == F.grid_sample on sampled voxels
        inside = ((pl >= 0) & (pl <= 256)).all(dim=1).float().mean().item()
        assert inside > 0.5, inside
        vox = m.construct_voxel(lf, rf, pl, pr)
        assert vox.shape == (1, 2 * f) + grid
        idx = torch.from_numpy(r.integers(0, v, 4096)).to(dev())
        for feat, pts, half in ((lf, pl, 0), (rf, pr, 1)):
            g = (pts[0][:, idx] / 256.0 * 2 - 1).t().reshape(1, 1, -1, 2)
            ref = F.grid_sample(feat, g, mode="bilinear", padding_mode="zeros", align_corners=False)[0, :, 0]   # [F, 4096]
            got = vox.view(1, 2 * f, v)[0, half * f:(half + 1) * f][:, idx]
            assert (got - ref).abs().max().item() < 1e-4
        # every layer of trunk_3d (vernier.py:415-438), each checked against torch-CPU on crops of ITS OWN input
        img = m.vimg_feat(vox)
        _check_layer_spots(m.vimg_feat[0], vox, img, what="vimg_feat k1")
        v1 = m.conv1(vox)
        _check_layer_spots(m.conv1[0], vox, v1, tol=4e-4, what="conv1 k7 (Winograd F(4,7))")
        plan = _Plan()
        sc, bi = _folded_bn(m.conv1[0][1], plan)
        v1d = _get_layer(m.conv1[0][0], plan)(vox, sc, bi, None, ops.EPI_RELU, None, exact=True)
        _check_layer_spots(m.conv1[0], vox, v1d, tol=2e-5, what="conv1 k7 (direct)")
        e = (v1 - v1d).abs().max().item() / v1d.abs().max().item()
        assert e < 3e-4, f"k7 Winograd vs direct over the whole tensor: {e:.2e}"
        del v1d
        v2 = m.conv2.fused(v1, residual=v1, residual_after_act=True)
        _check_layer_spots(m.conv2[0], v1, v2, residual=v1, after_act=True, what="conv2 k5")
        v3 = m.conv3.fused(v2, residual=v2, residual_after_act=True)
        _check_layer_spots(m.conv3[0], v2, v3, residual=v2, after_act=True, what="conv3 k5 dil2")
        for conv_seq, x_in in ((m.conv2, v1), (m.conv3, v2)):
            plan = _Plan()
            sc, bi = _folded_bn(conv_seq[0][1], plan)
            layer = _get_layer(conv_seq[0][0], plan)
            a = layer(x_in, sc, bi, None, ops.EPI_RELU, None)
            b = layer(x_in, sc, bi, None, ops.EPI_RELU, None, exact=True)
            e = (a - b).abs().max().item() / b.abs().max().item()
            assert e < 2e-5, f"k5 Winograd vs direct over the whole tensor: {e:.2e}"
            del a, b
        del v1, v2
        hg = m.hg_conv3d(v3, residual=v3)                              # hourglass_downsample_16 + v  (:420-423)
        # first / last layers of the hourglass on their own inputs
        o1 = m.hg_conv3d.conv1(v3)
        w1 = m.hg_conv3d.conv1[0][0].weight.detach().cpu()
        bn1 = m.hg_conv3d.conv1[0][1]
        for pt in _spot_points(tuple(o1.shape[2:]), tile=(2, 4, 32)):
            ref = _conv_at(v3, w1, pt, 3, 1, stride=2)
            ref = torch.relu((ref - bn1.running_mean.cpu()) / torch.sqrt(bn1.running_var.cpu() + bn1.eps) * bn1.weight.detach().cpu() + bn1.bias.detach().cpu())
            got = o1[0, :, pt[0], pt[1], pt[2]].cpu()
            assert (got - ref).abs().max().item() <= 2e-4 * max(ref.abs().max().item(), 1.0), ("hg conv1 s2", pt)
        del o1
        t = m.fg_cls_head[0].fused(hg, relu=True)
        _check_layer_spots(m.fg_cls_head[0], hg, t, what="fg_cls_head[0] k3")
        occ = m.fg_cls_head[2].fused(t, sigmoid=True)
        _check_layer_spots(m.fg_cls_head[2], t, occ, relu=False, sigmoid=True, what="occupancy head k3 -> 1 channel")
        cat = torch.cat([hg, img * occ], dim=1)
        v4 = m.conv4(cat)
        _check_layer_spots(m.conv4[0], cat, v4, what="conv4 k3 (Winograd F(4,3))")
        bev_ref = F.avg_pool3d(v4, (4, 1, 1), (4, 1, 1)).reshape(1, -1, grid[1], grid[2])
        del cat, t
        # the fused trunk (in-place concat, fused residuals) reproduces the layer-by-layer evaluation
        bev, occ2, _ = m.trunk_3d(vox)
        assert bev.shape == bev_ref.shape and torch.isfinite(bev).all()
        assert (bev - bev_ref).abs().max().item() <= 1e-5 * bev_ref.abs().max().item()
        assert (occ2 - occ).abs().max().item() <= 1e-6


@pytest.mark.parametrize("grid", [(32, 128, 192), (96, 96, 96)], ids=["released_32x128x192", "cfg3_crop_96"])
def test_local_trunk_split_mode_full_size(grid):
    """The local trunk in split mode (the default at inference) at full size: the 7^3 / 5^3 / dilated 5^3 / 1^3 layers
    on the gather's own output against torch-CPU on crops at the EXACT-fp32 tolerance (2e-5: the fp32 Winograd F(4,7) form needs
    4e-4), and the whole trunk against the fp32-MFMA trunk (whose own error, F(4,7)'s 1e-4, bounds the comparison)."""
    import bench
    from snvc_amd import ops
    from snvc_amd.models import submodule as S
    from snvc_amd.models.submodule import SplitT
    m = _local_model(grid)
    f = 32
    r = np.random.default_rng(7)
    lf = torch.from_numpy(r.standard_normal((1, f, 64, 64)).astype(np.float32)).to(dev())
    rf = torch.from_numpy(r.standard_normal((1, f, 64, 64)).astype(np.float32)).to(dev())
    pl, pr = bench.projected_coordinates(1, grid, dev())
    with torch.no_grad():
        vox = m.construct_voxel(lf, rf, pl, pr)
        before = S._ROUTES["x3_local_trunk"]
        bev, occ, _ = m.trunk_3d(vox)                                   # precision "auto": split mode
        assert S._ROUTES["x3_local_trunk"] == before + 1
        assert not m.__dict__.get("_snvc_x3_off") and all(int(g.flag.item()) == 0 for g in m.__dict__["_snvc_x3_guard"].values())
        m.precision = "f32"
        bev32, occ32, _ = m.trunk_3d(vox)
        assert S._ROUTES["x3_local_trunk"] == before + 1
        e = (bev - bev32).abs().max().item() / bev32.abs().max().item()
        assert e < 3e-4, f"split-mode trunk vs fp32-MFMA trunk (BEV features): {e:.2e}"
        assert (occ - occ32).abs().max().item() < 3e-4          # sigmoid outputs in [0, 1]; the fp32 trunk carries F(4,7)'s 1e-4
        # layer by layer on the same input, against torch-CPU on crops
        mul = ops.split_scale_for(vox)
        vs = SplitT(ops.to_split(vox, mul_dev=mul), 0, None, mul)
        img = m.vimg_feat.fused_x3(vs)
        _check_layer_spots(m.vimg_feat[0], vox, ops.from_split(img.t, img.exp), tol=2e-5, what="split vimg_feat k1")
        v1 = m.conv1.fused_x3(vs)
        v1f = ops.from_split(v1.t, v1.exp)
        _check_layer_spots(m.conv1[0], vox, v1f, tol=2e-5, what="split conv1 k7")
        v2 = m.conv2.fused_x3(v1, residual=v1, residual_after_act=True)
        v2f = ops.from_split(v2.t, v2.exp)
        _check_layer_spots(m.conv2[0], v1f, v2f, residual=v1f, after_act=True, tol=2e-5, what="split conv2 k5")
        v3 = m.conv3.fused_x3(v2, residual=v2, residual_after_act=True)
        _check_layer_spots(m.conv3[0], v2f, ops.from_split(v3.t, v3.exp), residual=v2f, after_act=True, tol=2e-5, what="split conv3 k5 dil2")


@pytest.mark.parametrize("grid,n", [((32, 128, 192), 2), ((96, 96, 96), 1), ((32, 64, 96), 3)], ids=["released_2crops", "cfg3_crop_96", "small_3crops"])
def test_split_gather_is_bitwise_the_fp32_gather_then_to_split(grid, n):
    """r4: ``construct_voxel_x3`` (snvc_voxel_gather_forward_split) -- the gather writing the split C8 pair itself, scaled by the
    features' own maximum -- equals ``to_split(construct_voxel(...), mul_dev=that scale)`` bit for bit (hi and lo planes), incl.
    coordinates outside the image; the model's forward takes it in split mode and gives the same outputs as from the fp32 gather."""
    import bench
    from snvc_amd import ops
    from snvc_amd.models import submodule as S
    m = _local_model(grid)
    f = 32
    r = np.random.default_rng(17)
    lf = torch.from_numpy((3.7 * r.standard_normal((n, f, 64, 64))).astype(np.float32)).to(dev())
    rf = torch.from_numpy(r.standard_normal((n, f, 64, 64)).astype(np.float32)).to(dev())
    pl, pr = bench.projected_coordinates(n, grid, dev())
    pl[:, :, ::97] -= 300.0                                      # some samples outside the image: zero taps
    with torch.no_grad():
        vs = m.construct_voxel_x3(lf, rf, pl, pr)
        assert vs is not None, "split mode qualifies: the gather writes the pair"
        vox = m.construct_voxel(lf, rf, pl, pr)
        mul = ops.split_scale_for(lf, rf)
        assert torch.equal(mul, vs.mul_dev) and 8192.0 <= float(mul.item()) * max(lf.abs().max().item(), rf.abs().max().item()) < 16384.0
        ref = ops.to_split(vox, mul_dev=mul)
        assert tuple(ref.shape) == tuple(vs.t.shape)
        assert torch.equal(ref, vs.t), "hi / lo planes of the fused gather differ from gather + layout pass"
        before = S._ROUTES["x3_local_trunk"]
        bev_a, occ_a, _ = m.trunk_3d(vs)
        bev_b, occ_b, _ = m.trunk_3d(vox)
        assert S._ROUTES["x3_local_trunk"] == before + 2
        # the only difference: the scale (features' maximum vs the voxels' own maximum, both powers of two: exact rescaling
        # unless the two exponents differ, then the lo parts round differently at 2^-22)
        e = (bev_a - bev_b).abs().max().item() / bev_b.abs().max().item()
        assert e < 2e-6, e
        assert (occ_a - occ_b).abs().max().item() < 2e-6
        m.precision = "f32"                                       # a pair handed to a trunk that left split mode: taken apart again
        bev_c, _, _ = m.trunk_3d(vs)
        bev_d, _, _ = m.trunk_3d(vox)
        # (the pair carries 22 bits of the fp32 voxel: 2^-22 on the input, a few 1e-6 behind the fp32 trunk's Winograd layers)
        assert (bev_c - bev_d).abs().max().item() / bev_d.abs().max().item() < 2e-5


def test_training_step_full_size_properties():
    """cfg4 at full size (1 pair, cfg2 shapes, train-mode BatchNorm): finite gradients for every parameter and both
    feature maps, equality of the step with the Winograd forms off (desc.algo = DIRECT), and the bilinear adjoint
    identities <conv(x,w), g> = <x, dgrad(g)> = <w, wgrad(x,g)> of one full-size layer."""
    import bench
    from snvc_amd import _lib, ops
    from snvc_amd.models import submodule as S
    from snvc_amd.models.stereo_volume import GlobalStack

    def step(bits):
        m = GlobalStack(C)
        m.load_state_dict(bench.seeded_state(m))
        m.train().to(dev())
        left, right, shift = bench.make_inputs(0, dev())
        left.requires_grad_(); right.requires_grad_()
        with ops.conv_variant(bits):
            out = m.forward_pair(left, right, shift, 1)
            loss = out.pow(2).mean()
            loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        grads["left"], grads["right"] = left.grad.clone(), right.grad.clone()
        return loss.item(), grads

    l0, g0 = step(0)
    l1, g1 = step(_lib.ALGO_DIRECT)
    assert np.isfinite(l0) and abs(l0 - l1) <= 1e-4 * abs(l1)
    for k in g0:
        assert torch.isfinite(g0[k]).all(), k
        assert g0[k].abs().max().item() > 0, k
        e = (g0[k] - g1[k]).abs().max().item() / max(g1[k].abs().max().item(), 1e-30)
        # Not bit-equal: the two runs differ by ~2e-6 of the range per layer, which flips ~1e-5 of the 1e9 ReLU
        # masks.  A parameter gradient sums over every voxel (flips average out: < 5e-3 of its max); a feature-map
        # gradient at one pixel sums only ~1.7e5 terms, so ONE flipped mask moves it by ~0.25 % of its typical
        # value and the worst of 1e6 pixels by ~1 % (measured 1.0e-2): bounded at 5e-2.
        assert e < (5e-2 if k in ("left", "right") else 5e-3), (k, e)
    del g0, g1
    torch.cuda.empty_cache()
    conv = S.HipConv3d(C, C, 3, 1, 1, bias=False).to(dev())
    torch.manual_seed(3)
    torch.nn.init.normal_(conv.weight, std=0.05)
    x = torch.randn(1, C, D, H, W, device=dev(), requires_grad=True)
    with ops.conv_variant(_lib.ALGO_DIRECT):       # the exact fp32 chains: the identities hold to accumulation rounding
        y = conv(x)
        g = torch.randn_like(y)
        y.backward(g)
    lhs = (y.detach().double() * g.double()).sum().item()
    via_x = (x.detach().double() * x.grad.double()).sum().item()
    via_w = (conv.weight.detach().double() * conv.weight.grad.double()).sum().item()
    assert abs(lhs - via_x) <= 1e-5 * abs(lhs) + 1e-2, (lhs, via_x)
    # <w, dw> sums 27648 terms of either sign: elements of dw carry independent fp32 accumulation errors (5.75e6 products
    # each; the float64 comparison in test_gpu_parity.py bounds them by ~1e-5 of their rms), so the identity holds to
    # 4 sigma = 4 * 1e-5 * rms(dw) * |w|_2 (0.8 here; measured 0.07), not to a fraction of the (cancelling) total
    dw_direct = conv.weight.grad.detach()
    tol_w = 4e-5 * dw_direct.double().pow(2).mean().sqrt().item() * conv.weight.detach().double().norm().item()
    assert abs(lhs - via_w) <= tol_w, (lhs, via_w, tol_w)
    # the Winograd-domain weight gradient over the same 46080 tiles: one tile lost or taken twice would move an element
    # by ~1e-3 of the largest one (sqrt(128 voxels) against 4.5 sigma of sqrt(5.75e6)); measured difference ~1e-5
    dw_wino = ops.conv3d_wgrad(x.detach(), g, 3, 1, 1, 1)
    assert not torch.equal(dw_wino, dw_direct), "the default form of this layer is the Winograd-domain one"
    e = (dw_wino - dw_direct).abs().max().item() / dw_direct.abs().max().item()
    assert e <= 5e-5, e
    via_w = (conv.weight.detach().double() * dw_wino.double()).sum().item()
    assert abs(lhs - via_w) <= 4 * tol_w, (lhs, via_w, tol_w)


def test_training_step_full_size_sheared_vs_general():
    """cfg4 at full size: the step through the sheared first layer with the train-mode BatchNorm folded around it
    (_ShearedFirstConvBNFn: no raw result, no raw gradient, statistics in the kernels' epilogues) against the same step on the
    general factored function (``sheared=False``: warped half built, 3D weight / data gradients, separate statistics passes):
    same loss, every parameter gradient and both feature gradients within the bounds of the DIRECT-vs-Winograd comparison
    above (the two paths differ by fp32 summation order, which flips ~1e-5 of the ReLU masks)."""
    import bench
    from snvc_amd.models import submodule as S
    from snvc_amd.models.stereo_volume import GlobalStack

    def step(sheared):
        m = GlobalStack(C)
        m.load_state_dict(bench.seeded_state(m))
        m.train().to(dev())
        left, right, shift = bench.make_inputs(0, dev())
        left.requires_grad_(); right.requires_grad_()
        before = S._ROUTES["sheared_first_conv_train_fused_bn"]
        out = m.forward_pair(left, right, shift, 1, sheared=sheared)
        loss = out.pow(2).mean()
        loss.backward()
        assert S._ROUTES["sheared_first_conv_train_fused_bn"] == before + (1 if sheared else 0)
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        grads["left"], grads["right"] = left.grad.clone(), right.grad.clone()
        stats = {k: v.detach().clone() for k, v in m.state_dict().items() if "running" in k}
        return loss.item(), grads, stats

    l0, g0, s0 = step(True)
    torch.cuda.empty_cache()
    l1, g1, s1 = step(False)
    assert np.isfinite(l0) and abs(l0 - l1) <= 1e-4 * abs(l1)
    for k in g0:
        assert torch.isfinite(g0[k]).all(), k
        e = (g0[k] - g1[k]).abs().max().item() / max(g1[k].abs().max().item(), 1e-30)
        assert e < (5e-2 if k in ("left", "right") else 5e-3), (k, e)
    for k in s0:        # BatchNorm bookkeeping moved the same way on both paths
        e = (s0[k] - s1[k]).abs().max().item() / max(s1[k].abs().max().item(), 1e-30)
        assert e < 1e-4, (k, e)
