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
    """The benchmarked step at full size: factored first convolution vs the materialised concat volume."""
    from snvc_amd.models.stereo_volume import GlobalStack
    import bench
    m = GlobalStack(C)
    m.load_state_dict(bench.seeded_state(m))
    m.eval().to(dev())
    dl, dr, ds = (torch.from_numpy(a).to(dev()) for a in pair)
    with torch.no_grad():
        a = m.forward_pair(dl, dr, ds, 1, factored=True)
        b = m.forward_pair(dl, dr, ds, 1, factored=False)
    assert a.shape == (1, 1, D, H, W) and torch.isfinite(a).all()
    err = (a - b).abs().max().item() / b.abs().max().item()
    assert err < 1e-4, err
