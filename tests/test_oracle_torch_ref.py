"""oracle.torch_ref vs the golden vectors produced by the imported reference
(tests/golden/make_golden.py).  Runs on CPU; pins the checker the GPU parity tests use."""
import numpy as np
import pytest
import torch

import golden_cases as GC
from oracle import torch_ref as T

G = GC.load_golden()
# the generator demanded bit equality in this container; on another build of torch (other
# oneDNN kernels / thread counts) allow rounding-level drift
TOL = dict(rtol=1e-5, atol=1e-5)


def load(mod, seed):
    mod.load_state_dict(T.seeded_state_dict(mod, seed), strict=True)
    return mod.eval()


@pytest.mark.parametrize("name", list(GC.CONV_CASES))
def test_convbn_3d(name):
    cin, cout, k, s, p, dil, gn, shape, seed = GC.CONV_CASES[name]
    m = load(T.convbn_3d(cin, cout, k, s, p, dilation=dil, gn=gn), seed)
    with torch.no_grad():
        y = m(GC.randn((1, cin) + shape, seed + 1)).numpy()
    np.testing.assert_allclose(y, G[f"conv/{name}"], **TOL)


@pytest.mark.parametrize("name", list(GC.HOURGLASS_CASES))
def test_hourglass(name):
    c, gn, shape, seed = GC.HOURGLASS_CASES[name]
    m = load(T.hourglass(c, gn=gn), seed)
    x = GC.randn((1, c) + shape, seed + 1)
    with torch.no_grad():
        out, pre, post = m(x, None, None)
        np.testing.assert_allclose(out.numpy(), G[f"hourglass/{name}/out"], **TOL)
        np.testing.assert_allclose(pre.numpy(), G[f"hourglass/{name}/pre"], **TOL)
        np.testing.assert_allclose(post.numpy(), G[f"hourglass/{name}/post"], **TOL)
        sq = m(x, GC.randn(tuple(pre.shape), seed + 2), GC.randn(tuple(post.shape), seed + 3))[0]
        np.testing.assert_allclose(sq.numpy(), G[f"hourglass/{name}/sq_out"], **TOL)


@pytest.mark.parametrize("name", list(GC.HOURGLASS16_CASES))
def test_hourglass16(name):
    c, gn, shape, seed = GC.HOURGLASS16_CASES[name]
    m = load(T.hourglass_downsample_16(c, gn=gn), seed)
    with torch.no_grad():
        y = m(GC.randn((1, c) + shape, seed + 1))
    np.testing.assert_allclose(y[:, ::2, :, ::2, ::2].numpy(), G[f"hourglass16/{name}_sub"], **TOL)
    s = G[f"hourglass16/{name}_sum"]
    assert abs(y.double().sum().item() - s[0]) <= 1e-6 * s[1]


def test_disparityregression():
    x = GC.randn((2, 12, 5, 7), 901)
    depth = torch.from_numpy(np.linspace(2.0, 40.0, 12).astype(np.float32))
    np.testing.assert_allclose(T.disparityregression(x, depth).numpy(), G["disparityregression"], **TOL)


@pytest.mark.parametrize("name", list(GC.TRUNK_CASES))
def test_trunk(name):
    grid, gn, n, fh, fw, seed = GC.TRUNK_CASES[name]
    m = load(T.VernierTrunk(dim=32, grid=grid, gn=gn), seed)
    lf, rf, gpl, gpr = GC.trunk_inputs(n, 32, fh, fw, grid, seed + 1)
    keep = gpl.clone()
    with torch.no_grad():
        vox = T.sample_2d_feat(lf, rf, gpl, gpr, GC.RESOLUTION, grid)
        assert torch.equal(keep, gpl)  # our restatement leaves the caller's tensor alone
        np.testing.assert_allclose(vox[:, ::7, ::3, ::5, ::5].numpy(), G[f"trunk/{name}/voxel_sub"], **TOL)
        s = G[f"trunk/{name}/voxel_sum"]
        assert abs(vox.double().sum().item() - s[0]) <= 1e-6 * s[1]
        heat, occ, _, coords, _ = m.predict_3d_heatmaps(vox)
        bev, _, _ = m.trunk_3d(vox)
    np.testing.assert_allclose(heat.numpy(), G[f"trunk/{name}/ncf"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(occ.numpy(), G[f"trunk/{name}/occupancy"], **TOL)
    np.testing.assert_allclose(coords.numpy(), G[f"trunk/{name}/coordinates"], **TOL)
    np.testing.assert_allclose(bev[:, ::5].numpy(), G[f"trunk/{name}/bev_sub"], rtol=1e-4, atol=1e-4)
    # a12: index extraction (vernier.py:693) is bit-exact
    idx = np.argmax(heat.numpy().reshape(n, 9, -1), axis=2)
    assert np.array_equal(idx, G[f"trunk/{name}/argmax"])


@pytest.mark.parametrize("name", list(GC.TYPE2_CASES))
def test_trunk_type2(name):
    """vernier_type='BEV_type2' (vernier.py:191-248, :391-410): the restatement against the imported reference's outputs
    (tests/golden/make_golden_type2.py required bit equality when it made them)."""
    import os
    g2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vernier_type2.npz"))
    grid, gn, n, fh, fw, seed = GC.TYPE2_CASES[name]
    m = load(T.VernierTrunk(dim=32, grid=grid, gn=gn, vernier_type="BEV_type2"), seed)
    assert not hasattr(m, "coord_head")
    lf, rf, gpl, gpr = GC.trunk_inputs(n, 32, fh, fw, grid, seed + 1)
    with torch.no_grad():
        heat, occ, _, coords, _ = m.predict_3d_heatmaps(T.sample_2d_feat(lf, rf, gpl, gpr, GC.RESOLUTION, grid))
    assert coords is None
    np.testing.assert_allclose(heat.numpy(), g2[f"{name}/ncf"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(occ.numpy(), g2[f"{name}/occupancy"], **TOL)


def test_reference_gather_mutates_in_place_flag():
    assert bool(G["gather/inplace_normalised"][0])


@pytest.mark.parametrize("name", list(GC.GLOBAL_CASES))
def test_global_stack(name):
    c, shape, seed = GC.GLOBAL_CASES[name]
    m = load(T.GlobalStack(c), seed)
    with torch.no_grad():
        y = m(GC.randn((1, 2 * c) + shape, seed + 1))
    np.testing.assert_allclose(y.numpy(), G[f"global/{name}"], **TOL)
