"""oracle.numpy_ref.sample_2d_feat against torch's grid_sample (through oracle.torch_ref, itself pinned
bit-exact to the imported reference) and against the golden voxel samples."""
import numpy as np
import torch

import golden_cases as GC
from oracle import numpy_ref as NR
from oracle import torch_ref as T


def test_numpy_gather_matches_torch_grid_sample():
    r = np.random.default_rng(3)
    n, f, hf, wf, grid = 2, 5, 7, 9, (2, 3, 5)
    v = grid[0] * grid[1] * grid[2]
    lf = r.standard_normal((n, f, hf, wf)).astype(np.float32)
    rf = r.standard_normal((n, f, hf, wf)).astype(np.float32)
    res = (28, 36)
    pts = r.uniform(-10, 46, (n, 2, v)).astype(np.float32)
    pts[0, 0, :3] = [0.0, 36.0, -2.0]
    pts[0, 1, :3] = [0.0, 28.0, 14.0]
    pr = pts[:, :, ::-1].copy()
    exp = T.sample_2d_feat(torch.from_numpy(lf), torch.from_numpy(rf), torch.from_numpy(pts), torch.from_numpy(pr),
                           res, grid).reshape(n, 2 * f, v).numpy()
    got = NR.sample_2d_feat(lf, rf, pts, pr, res)
    np.testing.assert_allclose(got, exp, rtol=0, atol=2e-6)


def test_numpy_gather_matches_golden():
    G = GC.load_golden()
    grid, gn, n, fh, fw, seed = GC.TRUNK_CASES["G1"]
    lf, rf, gpl, gpr = GC.trunk_inputs(n, 32, fh, fw, grid, seed + 1)
    vox = NR.sample_2d_feat(lf.numpy(), rf.numpy(), gpl.numpy(), gpr.numpy(), GC.RESOLUTION).reshape(n, 64, *grid)
    np.testing.assert_allclose(vox[:, ::7, ::3, ::5, ::5], G["trunk/G1/voxel_sub"], rtol=0, atol=2e-6)


def test_grid_projection_oracle_matches_reference_golden():
    """oracle.numpy_ref.grid_projection was asserted bit-equal to the reference's _init_3d_grid /
    _to_cam / _generate_grid_proj by make_golden.py; here against the committed samples."""
    G = GC.load_golden()
    gp = GC.grid_proj_case()
    g = NR.init_3d_grid(gp["x_range"], gp["y_range"], gp["z_range"], gp["grid"])
    assert g.shape == (3, 16, 32, 48) and g[0, 0, :, 0].tolist() == np.linspace(-1.6, 1.6, 32).tolist()
    cl, cr, g3 = NR.grid_projection(gp["samples"], gp["P_left"], gp["P_right"], gp["trans_l"], gp["trans_r"], g)
    assert cl.dtype == np.float32 and cl.shape == (3, 2, 16 * 32 * 48)
    np.testing.assert_allclose(cl[:, :, ::37], G["gridproj/left_sub"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(cr[:, :, ::37], G["gridproj/right_sub"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(g3[:, ::97], G["gridproj/grid3d_sub"], rtol=1e-13, atol=1e-13)
    s = G["gridproj/sum"]
    assert abs(cl.astype(np.float64).sum() - s[0]) <= 1e-7 * s[2]
