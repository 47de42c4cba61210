"""fp16-storage mode (BASELINE.json configs[4], "High-res local model ... 64ch, fp16 with MFMA").

The reference has no half-precision path ("its tolerance must be set vs the fp32 result"),
so the parity targets are:
  * per layer: torch's fp32 convolution of the SAME half-rounded inputs and weights (what an exact
    fp16-storage / fp32-accumulate layer computes), elementwise |err| <= 1e-3*|ref| + 2e-4*rms(ref)
    (the result's own rounding to half is 4.9e-4 relative; fp32 accumulation order is ~1e-5 of rms);
  * gather: bit-equal to the fp32 gather rounded to half;
  * whole trunk: this library's fp32 path on the same inputs and parameters, max|err| <= 2e-2 * rms(ref)
    (20+ layers, each re-rounding its activations to half; measured values are printed).
"""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def h(t):
    """round to half and back: what the kernels see"""
    return t.half().float()


def close_f16(got, ref, what, rtol=1e-3, arms=2e-4):
    got, ref = got.double(), ref.double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    rms = ref.pow(2).mean().sqrt().item()
    bound = rtol * ref.abs() + arms * max(rms, 1e-30)
    ratio = ((got - ref).abs() / bound).max().item()
    assert ratio <= 1.0, f"{what}: worst |err| / (1e-3|ref| + 2e-4 rms) = {ratio:.2f}"


def seeded_bn(bn, g):
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(g.uniform(0.5, 1.5, bn.weight.shape).astype(np.float32)))
        bn.bias.copy_(torch.from_numpy(g.uniform(-0.2, 0.2, bn.bias.shape).astype(np.float32)))
        bn.running_mean.copy_(torch.from_numpy(g.uniform(-0.2, 0.2, bn.bias.shape).astype(np.float32)))
        bn.running_var.copy_(torch.from_numpy(g.uniform(0.5, 1.5, bn.bias.shape).astype(np.float32)))


def test_c8_layout_round_trip():
    from snvc_amd import ops
    x = torch.randn(2, 13, 3, 5, 7, device=dev())
    c8 = ops.to_c8(x)
    assert c8.shape == (2, 2, 3, 5, 7, 8) and c8.dtype == torch.float16
    assert torch.equal(ops.from_c8(c8, 13), h(x))
    assert torch.equal(c8[:, 1, ..., 5:], torch.zeros_like(c8[:, 1, ..., 5:]))       # channels 13..15 are zero
    # element (n, c, d, h, w) sits at [n, c // 8, d, h, w, c % 8]
    assert c8[1, 1, 2, 4, 6, 3].item() == x[1, 11, 2, 4, 6].half().item()
    # channel slices are views: converting into a slice of a larger buffer
    big = torch.zeros(2, 4, 3, 5, 7, 8, dtype=torch.float16, device=dev())
    ops.to_c8(x, out=big[:, 1:3])
    assert torch.equal(big[:, 1:3], c8) and big[:, 0].abs().sum() == 0 and big[:, 3].abs().sum() == 0


KINDS = {  # name: (k, stride, dil, transposed)
    "k1": (1, 1, 1, False), "k3": (3, 1, 1, False), "k3s2": (3, 2, 1, False), "k5": (5, 1, 1, False),
    "k5d2": (5, 1, 2, False), "k7": (7, 1, 1, False), "deconv": (3, 2, 1, True),
}


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("cin,cout,shape", [(64, 64, (5, 6, 37)), (16, 32, (4, 9, 33)), (128, 128, (2, 3, 40)),
                                            (64, 32, (6, 5, 36)), (32, 96, (3, 4, 34))])
def test_f16_layer_vs_torch(kind, cin, cout, shape):
    """Every layer kind of the fp16 family against torch's fp32 convolution of the half-rounded operands: BN affine,
    ReLU, residual before / after the activation, batch 2, tile-ragged sizes, 1-2 chunks, 1-2 output blocks.  32 output
    channels run the MI = 1 kernel forms, 64 / 128 the two-block forms, 96 a two-block form with a half-empty block."""
    from snvc_amd import ops
    from snvc_amd.models import submodule as S
    k, stride, dil, transposed = KINDS[kind]
    if stride == 2 and not transposed:
        shape = tuple(max(s, 3) for s in shape)
    g = np.random.default_rng(1000 * list(KINDS).index(kind) + cin + cout)
    pad = dil * (k - 1) // 2
    m = S._deconvbn_3d(cin, cout, False) if transposed else S.convbn_3d(cin, cout, k, stride, pad, dilation=dil)
    fan = cin * k ** 3
    with torch.no_grad():
        m[0].weight.copy_(torch.from_numpy((g.standard_normal(tuple(m[0].weight.shape)) * np.sqrt(2.0 / fan)).astype(np.float32)))
    seeded_bn(m[1], g)
    m.eval()
    x = torch.from_numpy(g.standard_normal((2, cin) + shape).astype(np.float32))
    with torch.no_grad():
        wq = h(m[0].weight)
        conv = (F.conv_transpose3d(h(x).double(), wq.double(), None, 2, 1, 1) if transposed
                else F.conv3d(h(x).double(), wq.double(), None, stride, pad, dil))
        ref = F.batch_norm(conv, m[1].running_mean.double(), m[1].running_var.double(), m[1].weight.double(),
                           m[1].bias.double(), False, 0.0, m[1].eps)
        res = torch.from_numpy(g.standard_normal(tuple(ref.shape)).astype(np.float32))
        m = m.to(dev())
        xc = ops.to_c8(x.to(dev()))
        rc = ops.to_c8(res.to(dev()))
        y = ops.from_c8(m.fused_f16(xc)).cpu()
        close_f16(y, ref, f"{kind} conv+bn")
        y = ops.from_c8(m.fused_f16(xc, relu=True, residual=rc)).cpu()
        close_f16(y, F.relu(ref + h(res).double()), f"{kind} relu(conv+res)")
        y = ops.from_c8(m.fused_f16(xc, relu=True, residual=rc, residual_after_act=True)).cpu()
        close_f16(y, F.relu(ref) + h(res).double(), f"{kind} relu(conv)+res")
        # output into a channel slice of a wider C8 buffer (the in-place torch.cat of vernier.py:433)
        big = torch.full((2, cout // 8 + 2) + tuple(ref.shape[2:]) + (8,), 7.0, dtype=torch.float16, device=dev())
        m.fused_f16(xc, relu=True, out=big[:, 1:1 + cout // 8])
        close_f16(ops.from_c8(big[:, 1:1 + cout // 8]).cpu(), F.relu(ref), f"{kind} sliced out")
        assert torch.all(big[:, 0] == 7.0) and torch.all(big[:, -1] == 7.0)


def test_f16_one_channel_head_vs_torch():
    """The occupancy head Conv3d(F, 1, 3) + Sigmoid (vernier.py:269-278): fp32 plane output."""
    from snvc_amd import ops
    from snvc_amd.models import submodule as S
    g = np.random.default_rng(11)
    for cin, shape in ((64, (5, 7, 41)), (32, (4, 4, 32))):
        conv = S.HipConv3d(cin, 1, 3, 1, 1, bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy((g.standard_normal(tuple(conv.weight.shape)) * 0.05).astype(np.float32)))
        x = torch.from_numpy(g.standard_normal((2, cin) + shape).astype(np.float32))
        ref = torch.sigmoid(F.conv3d(h(x).double(), h(conv.weight.detach()).double(), None, 1, 1))
        conv = conv.to(dev())
        with torch.no_grad():
            y = conv.fused_f16(ops.to_c8(x.to(dev())), sigmoid=True)
        assert y.dtype == torch.float32 and y.shape == ref.shape
        assert (y.cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("grid", [(4, 8, 12), (16, 16, 24)], ids=["direct", "lds_staged"])
def test_f16_gather_equals_rounded_fp32_gather(grid):
    from snvc_amd import ops
    g = np.random.default_rng(3)
    n, f, hf, wf = 2, 32, 16, 16
    v = grid[0] * grid[1] * grid[2]
    lf = torch.from_numpy(g.standard_normal((n, f, hf, wf)).astype(np.float32)).to(dev())
    rf = torch.from_numpy(g.standard_normal((n, f, hf, wf)).astype(np.float32)).to(dev())
    pts = g.uniform(-6, 70, (n, 2, v)).astype(np.float32)
    pts[0, 0, 0] = np.nan
    gl, gr = torch.from_numpy(pts).to(dev()), torch.from_numpy(pts[:, :, ::-1].copy()).to(dev())
    a = ops.voxel_gather_forward(lf, rf, gl, gr, (64, 64))                  # [N, 2F, V] fp32
    b = ops.voxel_gather_forward_f16(lf, rf, gl, gr, (64, 64))             # [N, 2F/8, V, 8] half
    exp = a.half().view(n, 2 * f // 8, 8, v).permute(0, 1, 3, 2)
    assert torch.equal(torch.isnan(b), torch.isnan(exp))
    assert torch.equal(torch.nan_to_num(b), torch.nan_to_num(exp))


def test_f16_elementwise_vs_torch():
    from snvc_amd import ops
    x = torch.randn(2, 16, 8, 5, 9, device=dev())
    occ = torch.rand(2, 1, 8, 5, 9, device=dev())
    xc = ops.to_c8(x)
    y = ops.from_c8(ops.mul_broadcast_c8(xc, occ))
    assert torch.equal(y, h(h(x) * occ))
    p = ops.avgpool_depth4_c8(xc)
    ref = F.avg_pool3d(h(x), (4, 1, 1), (4, 1, 1))
    assert p.shape == ref.shape and (p - ref).abs().max().item() < 1e-6


def _model(grid, F_, seed=2024):
    import bench
    from snvc_amd.models.vernier import VernierScale
    cfg = types.SimpleNamespace(vernier_type="BEV_type3", backbone="hrfeat", gn=False,
                                grid_resolution=[32, grid[1], 192], resolution=(256, 256),
                                x_range=(-1.0, 1.0), z_range=(-1.0, 1.0), num_parts=9)
    cfg.hrfeat = types.SimpleNamespace(output_channel=F_, name="identity")
    cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
    m = VernierScale(cfg)
    m.load_state_dict(bench.seeded_state(m, seed))
    return m.eval().to(dev())


@pytest.mark.parametrize("grid,F_", [((16, 16, 24), 32), ((16, 32, 48), 32), ((16, 32, 32), 64)],
                         ids=["hourglass", "hourglass16_F32", "hourglass16_F64"])
def test_f16_trunk_vs_fp32_trunk(grid, F_):
    """gather + 3D trunk in the fp16-storage mode against this library's fp32 path (same inputs, same parameters)."""
    m = _model(grid, F_)
    g = np.random.default_rng(17)
    v = grid[0] * grid[1] * grid[2]
    lf = torch.from_numpy(g.standard_normal((2, F_, 64, 64)).astype(np.float32)).to(dev())
    rf = torch.from_numpy(g.standard_normal((2, F_, 64, 64)).astype(np.float32)).to(dev())
    gl = torch.from_numpy(g.uniform(-8, 264, (2, 2, v)).astype(np.float32)).to(dev())
    gr = torch.from_numpy(g.uniform(-8, 264, (2, 2, v)).astype(np.float32)).to(dev())
    with torch.no_grad():
        bev32, occ32, _ = m.trunk_3d(m.construct_voxel(lf, rf, gl, gr))
        bev16, occ16, _ = m.trunk_3d_f16(m.construct_voxel_f16(lf, rf, gl, gr))
    assert bev16.dtype == torch.float32 and bev16.shape == bev32.shape and occ16.shape == occ32.shape
    rms = bev32.pow(2).mean().sqrt().item()
    e_bev = (bev16 - bev32).abs().max().item() / rms
    e_occ = (occ16 - occ32).abs().max().item()
    print(f"fp16 trunk vs fp32 trunk {grid} F={F_}: bev max|err|/rms = {e_bev:.2e}, occupancy max|err| = {e_occ:.2e}")
    assert e_bev <= 2e-2, e_bev
    assert e_occ <= 5e-3, e_occ


@pytest.mark.parametrize("name", ["G1", "G2"])
def test_f16_model_switch_vs_golden(name):
    """VernierScale.forward with ``model.precision = "f16"``: same dict, outputs near the REFERENCE's golden fp32
    outputs (tests/golden/reference_outputs.npz) within the fp16-storage tolerance."""
    import golden_cases as GC
    from test_gpu_parity import _cfg, rel_err, seeded
    from snvc_amd.models.vernier import VernierScale
    G = GC.load_golden()
    grid, gn, n, fh, fw, seed = GC.TRUNK_CASES[name]
    m = seeded(VernierScale(_cfg(grid, gn)), seed).to(dev())
    m.precision = "f16"
    lf, rf, gpl, gpr = GC.trunk_inputs(n, 32, fh, fw, grid, seed + 1)
    with torch.no_grad():
        out = m(lf.to(dev()), rf.to(dev()), gpl.to(dev()), gpr.to(dev()))
    assert set(out) == {"ncf", "occupancy", "coordinates"}
    for key, tol in (("occupancy", 5e-3), ("ncf", 2e-2), ("coordinates", 5e-3)):
        e = rel_err(out[key].cpu().numpy(), G[f"trunk/{name}/{key}"])
        print(f"f16 model vs reference golden {name}/{key}: {e:.2e}")
        assert e <= tol, (key, e)


def _spot_check_f16_layer(layer, x_c8, y_c8, pts, k, stride, dil, transposed):
    """Output voxels `pts` of relu(bn(conv(x))) (what ``fused_f16`` of a ConvBNReLU3d / the plain affine of a ConvBN3d gives)
    against torch-CPU float64 on the half-rounded weights and the input crop each voxel reads (zero-padded at the borders)."""
    from snvc_amd import ops
    conv, bn = (layer[0][0], layer[0][1]) if isinstance(layer[0], torch.nn.Sequential) else (layer[0], layer[1])
    relu = isinstance(layer[0], torch.nn.Sequential)
    wq = h(conv.weight.detach().cpu()).double()
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().cpu().double()
    bi = (bn.bias - bn.running_mean * bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().cpu().double()
    in_sp = tuple(x_c8.shape[2:5])
    for pt in pts:
        if transposed:          # contributors i of output o: 2i - 1 + k = o  ->  i in [o//2 - 1, o//2 + 1]
            lo, ext = [p // 2 - 1 for p in pt], 3
        else:
            pad_ = dil * (k - 1) // 2
            lo, ext = [p * stride - pad_ for p in pt], dil * (k - 1) + 1
        sl = [slice(max(l, 0), min(l + ext, s_)) for l, s_ in zip(lo, in_sp)]
        crop = ops.from_c8(x_c8[:, :, sl[0], sl[1], sl[2]].contiguous()).cpu().double()
        pad = []
        for l, s_ in reversed(list(zip(lo, in_sp))):
            pad += [max(-l, 0), max(l + ext - s_, 0)]
        crop = F.pad(crop, pad)
        if transposed:
            full = F.conv_transpose3d(crop, wq, None, 2, 1, 1)                         # 6^3 outputs for inputs lo .. lo+2
            raw = full[0, :, pt[0] - 2 * lo[0], pt[1] - 2 * lo[1], pt[2] - 2 * lo[2]]
        else:
            raw = F.conv3d(crop, wq, None, 1, 0, dil)[0, :, 0, 0, 0]
        ref = raw * sc + bi
        if relu:
            ref = torch.relu(ref)
        got = ops.from_c8(y_c8[:, :, pt[0]:pt[0] + 1, pt[1]:pt[1] + 1, pt[2]:pt[2] + 1].contiguous())[0, :, 0, 0, 0].cpu().double()
        tol = 1e-3 * ref.abs() + 2e-4 * max(ref.pow(2).mean().sqrt().item(), 1e-3)
        assert ((got - ref).abs() <= tol).all(), (k, stride, dil, transposed, pt, (got - ref).abs().max().item())


def test_f16_cfg5_full_size():
    """cfg5 at full size: grid (80,160,160), F = 64.  conv1 (7^3, 128 -> 64) against torch-CPU on input crops at
    spot voxels (tile edges, corners), the trunk against the fp32 path, everything finite."""
    from snvc_amd import ops
    grid, F_ = (80, 160, 160), 64
    m = _model(grid, F_)
    g = np.random.default_rng(23)
    v = grid[0] * grid[1] * grid[2]
    lf = torch.from_numpy(g.standard_normal((1, F_, 64, 64)).astype(np.float32)).to(dev())
    rf = torch.from_numpy(g.standard_normal((1, F_, 64, 64)).astype(np.float32)).to(dev())
    gl = torch.from_numpy(g.uniform(-8, 264, (1, 2, v)).astype(np.float32)).to(dev())
    gr = torch.from_numpy(g.uniform(-8, 264, (1, 2, v)).astype(np.float32)).to(dev())
    with torch.no_grad():
        vox = m.construct_voxel_f16(lf, rf, gl, gr)
        assert vox.shape == (1, 16, 80, 160, 160, 8)
        v1 = m.conv1.fused_f16(vox)
        conv, bn = m.conv1[0][0], m.conv1[0][1]
        wq = h(conv.weight.detach().cpu()).double()
        sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().cpu().double()
        bi = (bn.bias - bn.running_mean * bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().cpu().double()
        d, hh, w = grid
        for pt in [(0, 0, 0), (d - 1, hh - 1, w - 1), (3, 3, 31), (4, 4, 32), (40, 80, 95), (79, 0, 159), (17, 159, 128), (41, 77, 64)]:
            lo = [p - 3 for p in pt]
            sl = [slice(max(l, 0), min(l + 7, s)) for l, s in zip(lo, grid)]
            crop = ops.from_c8(vox[:, :, sl[0], sl[1], sl[2]].contiguous()).cpu().double()
            pad = []
            for l, s in reversed(list(zip(lo, grid))):
                pad += [max(-l, 0), max(l + 7 - s, 0)]
            ref = torch.relu(F.conv3d(F.pad(crop, pad), wq)[0, :, 0, 0, 0] * sc + bi)
            got = ops.from_c8(v1[:, :, pt[0]:pt[0] + 1, pt[1]:pt[1] + 1, pt[2]:pt[2] + 1].contiguous())[0, :, 0, 0, 0].cpu().double()
            assert ((got - ref).abs() <= 1e-3 * ref.abs() + 2e-4 * max(ref.pow(2).mean().sqrt().item(), 1e-3)).all(), pt
        # r3: the other layer kinds at the same full size -- k5 (conv2), dilated k5 (conv3, sub-grid classes), stride 2 and
        # transposed (the 16x hourglass's first and last layers) -- spot voxels against torch-CPU on input crops
        pts = [(0, 0, 0), (d - 1, hh - 1, w - 1), (3, 3, 31), (4, 4, 32), (40, 80, 95), (79, 0, 159), (17, 159, 128), (41, 77, 64)]
        _spot_check_f16_layer(m.conv2, v1, m.conv2.fused_f16(v1), pts, 5, 1, 1, False)
        _spot_check_f16_layer(m.conv3, v1, m.conv3.fused_f16(v1), pts, 5, 1, 2, False)
        s2 = m.hg_conv3d.conv1.fused_f16(v1)                                            # 64 -> 128 at half resolution
        _spot_check_f16_layer(m.hg_conv3d.conv1, v1, s2, [(p[0] // 2, p[1] // 2, p[2] // 2) for p in pts], 3, 2, 1, False)
        up = m.hg_conv3d.conv12.fused_f16(s2)                                           # 128 -> 64 back at full resolution
        _spot_check_f16_layer(m.hg_conv3d.conv12, s2, up, pts + [(1, 1, 1), (78, 158, 158), (5, 4, 33)], 3, 2, 1, True)
        del v1, s2, up
        bev16, occ16, _ = m.trunk_3d_f16(vox)
        del vox
        assert torch.isfinite(bev16).all() and torch.isfinite(occ16).all()
        bev32, occ32, _ = m.trunk_3d(m.construct_voxel(lf, rf, gl, gr))
        rms = bev32.pow(2).mean().sqrt().item()
        e_bev = (bev16 - bev32).abs().max().item() / rms
        e_occ = (occ16 - occ32).abs().max().item()
        print(f"cfg5 fp16 vs fp32: bev max|err|/rms = {e_bev:.2e}, occupancy max|err| = {e_occ:.2e}")
        assert e_bev <= 3e-2 and e_occ <= 1e-2


def test_f16_released_shape_full_size_forward():
    """The released local model's grid (32,128,192), F = 32, through VernierScale.forward in the fp16-storage mode
    (gather -> 3D trunk in C8 half -> fp32 BEV neck + heads) against the fp32 mode of the same model on the same inputs."""
    import types
    import bench
    from snvc_amd.models.vernier import VernierScale
    grid = (32, 128, 192)
    cfg = types.SimpleNamespace(vernier_type="BEV_type3", backbone="hrfeat", gn=False, grid_resolution=[32, 128, 192],
                                resolution=(256, 256), x_range=(-1.0, 1.0), z_range=(-1.0, 1.0), num_parts=9)
    cfg.hrfeat = types.SimpleNamespace(output_channel=32, name="identity")
    cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
    m = VernierScale(cfg)
    m.load_state_dict(bench.seeded_state(m, 5))
    m.eval().to(dev())
    g = np.random.default_rng(29)
    lf = torch.from_numpy(g.standard_normal((2, 32, 64, 64)).astype(np.float32)).to(dev())
    rf = torch.from_numpy(g.standard_normal((2, 32, 64, 64)).astype(np.float32)).to(dev())
    pl, pr = bench.projected_coordinates(2, grid, dev())
    with torch.no_grad():
        ref = m(lf, rf, pl, pr)
        m.precision = "f16"
        out = m(lf, rf, pl, pr)
    assert set(out) == set(ref) == {"ncf", "occupancy", "coordinates"}
    for k, tol in (("occupancy", 1e-2), ("ncf", 3e-2), ("coordinates", 1e-2)):
        assert out[k].shape == ref[k].shape and torch.isfinite(out[k]).all(), k
        e = (out[k] - ref[k]).abs().max().item() / max(ref[k].abs().max().item(), 1e-30)
        print(f"released shape, fp16-storage vs fp32 forward, {k}: max|err|/max = {e:.2e}")
        assert e <= tol, (k, e)


# ================================================================================================
# r4 split mode ("f16x3", conv3d_f16.hip F16Cfg::PL >= 2): the fp32 layers' contraction on three half-precision MFMAs per
# product.  These are fp32 layers: they are held to the fp32 path's own tolerances (TIGHT = 2e-5 of the range per layer)
# against torch's fp32 / fp64 convolution -- not to the fp16-storage mode's.
# ================================================================================================
def _x3_reference(x, w, scale, bias, stride, transposed, relu, residual=None, pre=True):
    xd, wd = x.double().cpu(), w.double().cpu()
    if transposed:
        y = F.conv_transpose3d(xd, wd, None, 2, 1, 1)
    else:
        y = F.conv3d(xd, wd, None, stride, 1)
    y = y * scale.double().cpu().view(1, -1, 1, 1, 1) + bias.double().cpu().view(1, -1, 1, 1, 1)
    if residual is not None and pre:
        y = y + residual.double().cpu()
    if relu:
        y = torch.relu(y)
    if residual is not None and not pre:
        y = y + residual.double().cpu()
    return y


@pytest.mark.parametrize("case", ["k1_64_32", "k5_32_32", "k5d2_32_32", "k7_64_32", "k7_128_64", "deconv_64_32", "k3_32_1"])
def test_split_mode_local_trunk_layer_kinds_vs_float64(case):
    """The other layer kinds of the local trunk in split mode (vernier.py:249-278: 1^3, 5^3, dilated 5^3, 7^3, the 32-channel
    transposed layer, the one-channel occupancy head with its Sigmoid) against float64 at the exact-fp32 tolerance."""
    from snvc_amd import ops
    from test_gpu_parity import TIGHT, check
    torch.manual_seed(len(case))
    cin, cout, k, dil, transposed, shape = {
        "k1_64_32": (64, 32, 1, 1, False, (5, 9, 40)), "k5_32_32": (32, 32, 5, 1, False, (7, 9, 37)),
        "k5d2_32_32": (32, 32, 5, 2, False, (9, 10, 36)), "k7_64_32": (64, 32, 7, 1, False, (8, 9, 35)),
        "k7_128_64": (128, 64, 7, 1, False, (5, 8, 33)), "deconv_64_32": (64, 32, 3, 1, True, (3, 5, 18)),
        "k3_32_1": (32, 1, 3, 1, False, (6, 7, 38))}[case]
    pad = 1 if transposed else dil * (k - 1) // 2
    x = torch.relu(torch.randn(2, cin, *shape, device=dev())) * 2.0 + 0.01 * torch.randn(2, cin, *shape, device=dev())
    w = torch.randn((cin, cout, 3, 3, 3) if transposed else (cout, cin, k, k, k), device=dev()) * np.sqrt(2.0 / (cin * k ** 3))
    scale, bias = torch.rand(cout, device=dev()) + 0.5, torch.randn(cout, device=dev()) * 0.3
    layer = ops.Conv3dLayerX3(w, k, 2 if transposed else 1, pad, dil, transposed)
    xd, wd = x.double().cpu(), w.double().cpu()
    raw = F.conv_transpose3d(xd, wd, None, 2, 1, 1) if transposed else F.conv3d(xd, wd, None, 1, pad, dil)
    aff = raw * scale.double().cpu().view(1, -1, 1, 1, 1) + bias.double().cpu().view(1, -1, 1, 1, 1)
    mul = ops.split_scale_for(x)                                    # the data-derived scale, as the trunk's first layers get it
    xs = ops.to_split(x, mul_dev=mul)
    if cout == 1:
        y = layer(xs, 0, scale, bias, flags=ops.EPI_SIGMOID, x_mul_dev=mul)
        assert y.dtype == torch.float32 and tuple(y.shape) == (2, 1) + shape
        check(y.cpu().numpy(), torch.sigmoid(aff).numpy(), TIGHT, f"{case}: occupancy head")
        return
    ref = torch.relu(aff)
    ys = layer(xs, 0, scale, bias, flags=ops.EPI_RELU, out_exp=3, x_mul_dev=mul)
    check(ops.from_split(ys, 3).cpu().numpy(), ref.numpy(), TIGHT, f"{case}: split -> split, data-scaled input")
    y32 = layer(ops.to_split(x, 2), 2, scale, bias, flags=ops.EPI_RELU, to_f32=True)
    check(y32.cpu().numpy(), ref.numpy(), TIGHT, f"{case}: split -> float32")
    # a residual stored with another exponent than the result (res_mul), after the activation: the trunk's conv2(v) + v
    res = torch.randn_like(y32)
    yr = layer(xs, 0, scale, bias, residual=ops.to_split(res, 5), res_exp=5, flags=ops.EPI_RELU | ops.EPI_ADD_POST, out_exp=2, x_mul_dev=mul)
    check(ops.from_split(yr, 2).cpu().numpy(), (ref + res.double().cpu()).numpy(), TIGHT, f"{case}: residual with its own exponent")


@pytest.mark.parametrize("case", ["k3_32_32", "k3_64_64", "k3_32_64", "k3s2_32_64", "k3s2_64_64", "deconv_64_64", "k3_32_32_odd",
                                  "k3s2_32_64_small", "k3s2_64_64_small", "deconv_64_64_small", "k3s2_64_64_small_odd", "deconv_64_64_small_odd",
                                  "k3s2_32_64_q16", "k3s2_64_64_q16", "k3s2_64_128_q16_odd", "k3s2_8_64_q16_tiny"])
def test_split_mode_layers_vs_float64(case):
    from snvc_amd import _lib as L_, ops
    from test_gpu_parity import TIGHT, check
    torch.manual_seed(len(case) * 7 + ord(case[-1]))
    # "_small": the half-height tile forms of the stride-2 / transposed layers (SNVC_ALGO_X3_SMALL, r5: what a launch with few
    # workgroups picks); the others force the full-height forms (algo 0), whatever the launch size
    small = "_small" in case
    cin, cout, stride, transposed, shape = {
        "k3_32_32": (32, 32, 1, False, (8, 12, 40)), "k3_64_64": (64, 64, 1, False, (8, 8, 36)), "k3_32_64": (32, 64, 1, False, (5, 9, 33)),
        "k3s2_32_64": (32, 64, 2, False, (8, 12, 72)), "k3s2_64_64": (64, 64, 2, False, (6, 10, 42)),
        "deconv_64_64": (64, 64, 2, True, (4, 6, 20)), "k3_32_32_odd": (32, 32, 1, False, (3, 5, 31)),
        "k3s2_32_64_small": (32, 64, 2, False, (8, 12, 72)), "k3s2_64_64_small": (64, 64, 2, False, (6, 10, 42)),
        "deconv_64_64_small": (64, 64, 2, True, (4, 6, 20)), "k3s2_64_64_small_odd": (64, 64, 2, False, (7, 9, 67)),
        "deconv_64_64_small_odd": (64, 64, 2, True, (3, 5, 35)),
        # r5: conv3d_x3s2q_kernel (16x16x32, both planes, three image slots): one chunk, eight chunks, two output blocks, ragged extents
        "k3s2_32_64_q16": (32, 64, 2, False, (8, 12, 72)), "k3s2_64_64_q16": (64, 64, 2, False, (6, 10, 42)),
        "k3s2_64_128_q16_odd": (64, 128, 2, False, (7, 9, 67)), "k3s2_8_64_q16_tiny": (8, 64, 2, False, (2, 2, 3))}[case]
    q16 = "_q16" in case
    forced = None if (stride == 1 and not transposed) else (L_.ALGO_X3_Q16 if q16 else L_.ALGO_X3_SMALL if small else 0)
    x = torch.relu(torch.randn(2, cin, *shape, device=dev())) * 2.0 + 0.01 * torch.randn(2, cin, *shape, device=dev())
    w = torch.randn((cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3), device=dev()) * np.sqrt(2.0 / (cin * 27))
    scale, bias = torch.rand(cout, device=dev()) + 0.5, torch.randn(cout, device=dev()) * 0.3
    layer = ops.Conv3dLayerX3(w, 3, stride, 1, 1, transposed, algo=forced)
    if q16:     # split -> split only (no residual, no fp32 output in this form: the caller's rule takes another one for those)
        old = ops.Conv3dLayerX3(w, 3, stride, 1, 1, transposed, algo=0)
        ref = _x3_reference(x, w, scale, bias, stride, transposed, True)
        for x_exp, out_exp in ((0, 0), (5, 3)):
            xs = ops.to_split(x, x_exp)
            flag = torch.zeros(1, dtype=torch.int32, device=dev())
            ys = layer(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out_exp=out_exp, overflow=flag)
            got = ops.from_split(ys, out_exp)
            check(got.cpu().numpy(), ref.numpy(), TIGHT, f"{case}: 16x16x32 stride-2 form vs float64, exponents {x_exp}/{out_exp}")
            yo = ops.from_split(old(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out_exp=out_exp, overflow=flag), out_exp)
            assert (got - yo).abs().max().item() <= 2e-6 * yo.abs().max().item(), "the two instruction shapes differ by fp32 summation order only"
            assert flag.item() == 0
        y0 = ops.from_split(layer(ops.to_split(x, 2), 2, out_exp=1), 1)                   # no affine, no activation (negative values kept)
        check(y0.cpu().numpy(), F.conv3d(x.double().cpu(), w.double().cpu(), None, 2, 1).numpy(), TIGHT, f"{case}: bare convolution")
        with pytest.raises(ops.Unsupported):
            layer(ops.to_split(x, 2), 2, scale, bias, flags=ops.EPI_RELU, to_f32=True)
        flag = torch.zeros(1, dtype=torch.int32, device=dev())
        layer(ops.to_split(x, 2), 2, scale * 4096.0, bias, flags=ops.EPI_RELU, out_exp=8, overflow=flag)   # clamped and flagged
        assert int(flag.item()) == 1
        a_ = layer(ops.to_split(x, 2), 2, scale, bias, flags=ops.EPI_RELU, out_exp=1)
        for _ in range(3):      # the refill / slot rotation is deterministic: bitwise repeatable
            assert torch.equal(layer(ops.to_split(x, 2), 2, scale, bias, flags=ops.EPI_RELU, out_exp=1), a_)
        return
    if forced is not None:      # what the launch-size rule picks by itself at this size: half-height tiles for a transposed layer only
        auto = ops.Conv3dLayerX3(w, 3, stride, 1, 1, transposed)
        auto(ops.to_split(x, 2), 2, scale, bias, flags=ops.EPI_RELU, to_f32=True)
        assert auto.algo == (L_.ALGO_X3_SMALL if transposed else 0)
        if not transposed:      # ... a stride-2 layer with a split output and no residual: the 16x16x32 form
            auto(ops.to_split(x, 2), 2, scale, bias, flags=ops.EPI_RELU, out_exp=1)
            assert auto.algo == L_.ALGO_X3_Q16
    if stride == 1 and not transposed:      # every kernel form of a stride-1 layer gives the same values (same MFMA order per output)
        from snvc_amd import _lib
        xs0 = ops.to_split(x, 2)
        forms = [ops.Conv3dLayerX3(w, algo=a)(xs0, 2, scale, bias, flags=ops.EPI_RELU, to_f32=True)
                 for a in (0, _lib.ALGO_X3_NARROW, _lib.ALGO_X3_SMALL, _lib.ALGO_X3_SERIAL)]
        ref0 = _x3_reference(x, w, scale, bias, 1, False, True)
        for k, f in enumerate(forms):
            check(f.cpu().numpy(), ref0.numpy(), TIGHT, f"{case}: kernel form {k}")
    for x_exp, out_exp in ((0, 0), (5, 3)):
        xs = ops.to_split(x, x_exp)
        assert torch.allclose(ops.from_split(xs, x_exp), x, rtol=0, atol=2e-6 * x.abs().max().item())       # 22 bits
        ref = _x3_reference(x, w, scale, bias, stride, transposed, True)
        y32 = layer(xs, x_exp, scale, bias, flags=ops.EPI_RELU, to_f32=True)
        check(y32.cpu().numpy(), ref.numpy(), TIGHT, f"{case}: split -> float32, exponents {x_exp}/{out_exp}")
        flag = torch.zeros(1, dtype=torch.int32, device=dev())
        ys = layer(xs, x_exp, scale, bias, flags=ops.EPI_RELU, out_exp=out_exp, overflow=flag)
        check(ops.from_split(ys, out_exp).cpu().numpy(), ref.numpy(), TIGHT, f"{case}: split -> split")
        assert flag.item() == 0
        # residual before / after the activation, split residual, both output forms
        res = torch.randn_like(y32)
        rs = ops.to_split(res, out_exp)
        for pre, fl in ((True, ops.EPI_ADD_PRE), (False, ops.EPI_ADD_POST)):
            ref_r = _x3_reference(x, w, scale, bias, stride, transposed, True, res, pre)
            yr = layer(xs, x_exp, scale, bias, residual=rs, flags=ops.EPI_RELU | fl, out_exp=out_exp)
            check(ops.from_split(yr, out_exp).cpu().numpy(), ref_r.numpy(), TIGHT, f"{case}: residual pre={pre}")
            yr32 = layer(xs, x_exp, scale, bias, residual=rs, flags=ops.EPI_RELU | fl, out_exp=out_exp, to_f32=True)
            check(yr32.cpu().numpy(), ref_r.numpy(), TIGHT, f"{case}: residual pre={pre}, float32 out")
    # no affine, no activation
    y0 = layer(ops.to_split(x, 2), 2, to_f32=True)
    ref0 = (F.conv_transpose3d(x.double().cpu(), w.double().cpu(), None, 2, 1, 1) if transposed else
            F.conv3d(x.double().cpu(), w.double().cpu(), None, stride, 1))
    check(y0.cpu().numpy(), ref0.numpy(), TIGHT, f"{case}: bare convolution")


def test_split_mode_side_head_and_overflow_flag():
    from snvc_amd import ops
    from test_gpu_parity import TIGHT, check
    torch.manual_seed(5)
    x = torch.relu(torch.randn(1, 32, 6, 8, 40, device=dev()))
    w = torch.randn(32, 32, 3, 3, 3, device=dev()) * 0.05
    scale, bias = torch.rand(32, device=dev()) + 0.5, torch.randn(32, device=dev()) * 0.2
    head = torch.randn(32, device=dev())
    layer = ops.Conv3dLayerX3(w)
    xs = ops.to_split(x, 4)
    ys, yh = layer(xs, 4, scale, bias, flags=ops.EPI_RELU, out_exp=6, head=head)
    ref = _x3_reference(x, w, scale, bias, 1, False, True)
    check(ops.from_split(ys, 6).cpu().numpy(), ref.numpy(), TIGHT, "side head: the layer itself")
    check(yh.cpu().numpy(), (ref * head.double().cpu().view(1, -1, 1, 1, 1)).sum(1, keepdim=True).numpy(), TIGHT, "side head: the projection")
    # an exponent that pushes the result beyond half's range: finite (clamped) and flagged
    flag = torch.zeros(1, dtype=torch.int32, device=dev())
    big = layer(xs, 4, scale, bias, flags=ops.EPI_RELU, out_exp=16, overflow=flag)
    assert flag.item() == 1 and torch.isfinite(big.float()).all()
    with pytest.raises(RuntimeError):
        ops.Conv3dLayerX3(torch.randn(32, 32, 5, 5, 5, device=dev()), 5, 2, 2, 1, False)       # a stride-2 k5 layer: not on the path


@pytest.mark.parametrize("case", ["k3_32_32", "k3_48_64", "k3_64_32_ragged", "k3_8_32", "k3_24_96", "k7_64_32", "k7_16_64", "k5d2_32_32", "k5d2_8_96",
                                  "k5_32_32", "k5_16_64"])
def test_split_mode_16x16x32_forms_vs_float64(case):
    """r4: the kernel forms on v_mfma_f32_16x16x32_f16 (SNVC_ALGO_X3_Q16; the default only from ~1000 workgroups on, forced here on
    small and ragged shapes): 3x3x3 (four taps of one channel group per MFMA, the image double-buffered and refilled under the last
    k-steps of a chunk -- one chunk only, an odd number of chunks, one / two / three output-channel blocks; side head), 7^3,
    dilated 5^3 and 5^3 (four taps per MFMA, planes serial, sub-grid classes, the plain 5^3 layer's quads over all 125 taps,
    residual before / after the activation with its own exponent) against float64 at the exact-fp32 tolerance, and against the 32x32x16 forms."""
    from snvc_amd import _lib, ops
    from test_gpu_parity import TIGHT, check
    torch.manual_seed(100 + len(case))
    cin, cout, k, dil, shape = {
        "k3_32_32": (32, 32, 3, 1, (8, 12, 40)), "k3_48_64": (48, 64, 3, 1, (5, 9, 33)), "k3_64_32_ragged": (64, 32, 3, 1, (10, 9, 70)),
        "k3_8_32": (8, 32, 3, 1, (6, 7, 34)), "k3_24_96": (24, 96, 3, 1, (7, 5, 65)),
        "k7_64_32": (64, 32, 7, 1, (8, 9, 35)), "k7_16_64": (16, 64, 7, 1, (5, 6, 64)), "k5d2_32_32": (32, 32, 5, 2, (9, 10, 36)),
        "k5d2_8_96": (8, 96, 5, 2, (7, 7, 50)), "k5_32_32": (32, 32, 5, 1, (9, 10, 36)), "k5_16_64": (16, 64, 5, 1, (6, 7, 50))}[case]
    pad = dil * (k - 1) // 2
    x = torch.relu(torch.randn(2, cin, *shape, device=dev())) * 2.0 + 0.01 * torch.randn(2, cin, *shape, device=dev())
    w = torch.randn(cout, cin, k, k, k, device=dev()) * np.sqrt(2.0 / (cin * k ** 3))
    scale, bias = torch.rand(cout, device=dev()) + 0.5, torch.randn(cout, device=dev()) * 0.3
    q16 = ops.Conv3dLayerX3(w, k, 1, pad, dil, algo=_lib.ALGO_X3_Q16)
    old = ops.Conv3dLayerX3(w, k, 1, pad, dil, algo=0)
    raw = F.conv3d(x.double().cpu(), w.double().cpu(), None, 1, pad, dil)
    aff = raw * scale.double().cpu().view(1, -1, 1, 1, 1) + bias.double().cpu().view(1, -1, 1, 1, 1)
    ref = torch.relu(aff)
    flag = torch.zeros(1, dtype=torch.int32, device=dev())
    xs = ops.to_split(x, 3)
    ys = q16(xs, 3, scale, bias, flags=ops.EPI_RELU, out_exp=2, overflow=flag)
    got = ops.from_split(ys, 2)
    check(got.cpu().numpy(), ref.numpy(), TIGHT, f"{case}: 16x16x32 form vs float64")
    yo = ops.from_split(old(xs, 3, scale, bias, flags=ops.EPI_RELU, out_exp=2, overflow=flag), 2)
    assert (got - yo).abs().max().item() <= 2e-6 * yo.abs().max().item(), "the two instruction shapes differ by fp32 summation order only"
    assert int(flag.item()) == 0
    if k == 3:
        if cout == 32:      # the side head: the classifier's projection of the layer's own result
            head = torch.randn(cout, device=dev())
            y2, hv = q16(xs, 3, scale, bias, flags=ops.EPI_RELU, out_exp=2, head=head, overflow=flag)
            assert torch.equal(y2, ys)
            check(hv.cpu().numpy(), (ref * head.double().cpu().view(1, -1, 1, 1, 1)).sum(1, keepdim=True).numpy(), TIGHT, f"{case}: side head")
        yf = q16(xs, 3, scale, bias, flags=ops.EPI_RELU, to_f32=True)      # r6: a float32 result in this form (the training step's layers)
        check(yf.cpu().numpy(), ref.numpy(), TIGHT, f"{case}: 16x16x32 form, float32 result")
        with pytest.raises(ops.Unsupported):      # no split residual in this form: the caller takes another one
            q16(xs, 3, scale, bias, residual=ys, flags=ops.EPI_RELU | ops.EPI_ADD_POST, out_exp=2)
    else:
        res = torch.randn_like(got)
        rs = ops.to_split(res, 5)
        yb = q16(xs, 3, scale, bias, residual=rs, res_exp=5, flags=ops.EPI_RELU | ops.EPI_ADD_POST, out_exp=2, overflow=flag)
        check(ops.from_split(yb, 2).cpu().numpy(), (ref + res.double().cpu()).numpy(), TIGHT, f"{case}: residual after the activation")
        ya = q16(xs, 3, scale, bias, residual=rs, res_exp=5, flags=ops.EPI_RELU | ops.EPI_ADD_PRE, out_exp=2, overflow=flag)
        check(ops.from_split(ya, 2).cpu().numpy(), torch.relu(aff + res.double().cpu()).numpy(), TIGHT, f"{case}: residual before the activation")
    # an exponent too large for half: clamped and flagged, like every split-mode form
    q16(xs, 3, scale * 4096.0, bias, flags=ops.EPI_RELU, out_exp=8, overflow=flag)
    assert int(flag.item()) == 1


def test_split_mode_16x16x32_k3_is_bitwise_repeatable():
    """The 3x3x3 form reads its LDS image with inline-asm ds_read_b128 behind hand-counted waits while the LDS-DMA refill of the
    next channel group is in flight (csrc/conv3d_f16.hip, conv3d_x3q_kernel): a wait that is one short, or a refill that lands in the
    buffer still being read, shows up as a run-to-run difference.  Many workgroups per CU, four chunks, both buffers in use, with and
    without the side head: every repetition is bit-identical to the first, and the first agrees with the 32x32x16 form."""
    from snvc_amd import _lib, ops
    torch.manual_seed(77)
    x = torch.relu(torch.randn(1, 32, 48, 40, 160, device=dev())) + 0.01 * torch.randn(1, 32, 48, 40, 160, device=dev())
    w = torch.randn(32, 32, 3, 3, 3, device=dev()) * 0.06
    scale, bias = torch.rand(32, device=dev()) + 0.5, torch.randn(32, device=dev()) * 0.3
    head = torch.randn(32, device=dev())
    q16 = ops.Conv3dLayerX3(w, algo=_lib.ALGO_X3_Q16)
    old = ops.Conv3dLayerX3(w, algo=0)
    xs = ops.to_split(x, 3)
    flag = torch.zeros(1, dtype=torch.int32, device=dev())
    first, first_head = q16(xs, 3, scale, bias, flags=ops.EPI_RELU, out_exp=2, head=head, overflow=flag)
    first, first_head = first.clone(), first_head.clone()
    for rep in range(12):
        if rep % 2:
            y = q16(xs, 3, scale, bias, flags=ops.EPI_RELU, out_exp=2, overflow=flag)
            assert torch.equal(y, first), f"repetition {rep} differs"
        else:
            y, yh = q16(xs, 3, scale, bias, flags=ops.EPI_RELU, out_exp=2, head=head, overflow=flag)
            assert torch.equal(y, first) and torch.equal(yh, first_head), f"repetition {rep} differs"
    yo = ops.from_split(old(xs, 3, scale, bias, flags=ops.EPI_RELU, out_exp=2, overflow=flag), 2)
    assert (ops.from_split(first, 2) - yo).abs().max().item() <= 2e-6 * yo.abs().max().item()
    assert int(flag.item()) == 0


@pytest.mark.parametrize("case", ["k7_64_32", "k7_16_64", "k5d2_32_64", "k5d2_24_32", "k5_32_64", "k5_8_32"])
def test_f16_storage_16x16x32_forms_vs_torch(case):
    """r4: the fp16-STORAGE 7^3 / dilated 5^3 / 5^3 layers take the 16x16x32 form (one or two 32-channel blocks per workgroup; the
    plain 5^3 layer with its quads over all 125 taps) by default; against torch on the half-rounded operands like every layer of
    the family, and against the 32x32x16 form."""
    from snvc_amd import ops
    torch.manual_seed(200 + len(case))
    cin, cout, k, dil, shape = {"k7_64_32": (64, 32, 7, 1, (6, 9, 37)), "k7_16_64": (16, 64, 7, 1, (5, 6, 64)),
                                "k5d2_32_64": (32, 64, 5, 2, (9, 10, 36)), "k5d2_24_32": (24, 32, 5, 2, (7, 7, 50)),
                                "k5_32_64": (32, 64, 5, 1, (9, 10, 36)), "k5_8_32": (8, 32, 5, 1, (6, 7, 50))}[case]
    pad = dil * (k - 1) // 2
    x = torch.randn(2, cin, *shape, device=dev())
    w = torch.randn(cout, cin, k, k, k, device=dev()) * np.sqrt(2.0 / (cin * k ** 3))
    scale, bias = torch.rand(cout, device=dev()) + 0.5, torch.randn(cout, device=dev()) * 0.3
    xc = ops.to_c8(x)
    lay = ops.Conv3dLayerF16(w, k, 1, pad, dil, False)
    assert lay.q16
    ops.X3_Q16[0] = False
    try:
        old = ops.Conv3dLayerF16(w, k, 1, pad, dil, False)
    finally:
        ops.X3_Q16[0] = True
    assert not old.q16
    res = torch.randn(2, cout, *shape, device=dev())
    rc = ops.to_c8(res)
    ref = F.conv3d(h(x).double().cpu(), h(w).double().cpu(), None, 1, pad, dil) * scale.double().cpu().view(1, -1, 1, 1, 1) \
        + bias.double().cpu().view(1, -1, 1, 1, 1)
    for flags, exp in ((ops.EPI_RELU, torch.relu(ref)), (ops.EPI_RELU | ops.EPI_ADD_PRE, torch.relu(ref + h(res).double().cpu())),
                       (ops.EPI_RELU | ops.EPI_ADD_POST, torch.relu(ref) + h(res).double().cpu())):
        kw = dict(residual=rc) if flags != ops.EPI_RELU else {}
        got = ops.from_c8(lay(xc, scale, bias, flags=flags, **kw), cout)
        close_f16(got.cpu(), exp.float(), f"{case} flags={flags}")
        was = ops.from_c8(old(xc, scale, bias, flags=flags, **kw), cout)
        close_f16(got.cpu(), was.cpu(), f"{case} flags={flags}: 16x16x32 vs 32x32x16")
