"""Pins the C oracle of build_cost_volume with hand-derived known answers.

The reference ships no vectors for this op, so each test below
is derived by hand from BuildCostVolume_cuda.cu and cites the lines it exercises.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import native as O


def rng(seed=0):
    return np.random.default_rng(seed)


def test_zero_shift_is_copy():
    # (i) shift=0, ds=1: right half == right, left half == left for every d  (.cu:84-91)
    r = rng(1)
    L = r.standard_normal((2, 3, 5, 7)).astype(np.float32)
    R = r.standard_normal((2, 3, 5, 7)).astype(np.float32)
    out = O.cost_volume_forward(L, R, np.zeros((2, 4), np.float32), 1)
    assert out.shape == (2, 6, 4, 5, 7)
    for d in range(4):
        assert np.array_equal(out[:, :3, d], L)
        assert np.array_equal(out[:, 3:, d], R)


def test_half_pixel_row():
    # (ii) R=[10,20,30,40], shift=1.5 -> [0,0,15,25]: x=-1.5,-0.5 fail the gate (.cu:88)
    R = np.array([10, 20, 30, 40], np.float32).reshape(1, 1, 1, 4)
    out = O.cost_volume_forward(np.zeros_like(R), R, np.array([[1.5]], np.float32), 1)
    assert out[0, 1, 0, 0].tolist() == [0.0, 0.0, 15.0, 25.0]


def test_right_edge_clamp():
    # (iii) shift=0, w=W-1: passes x<=W-1, x_low clamps, returns R[W-1]  (.cu:41-46)
    R = np.arange(6, dtype=np.float32).reshape(1, 1, 1, 6) + 1
    out = O.cost_volume_forward(R, R, np.array([[0.0]], np.float32), 1)
    assert out[0, 1, 0, 0, -1] == 6.0


@pytest.mark.parametrize("k", [0, 1, 2, 5, 6, 9])
def test_integer_shift(k):
    # (iv) integer shift k: exact copy moved right by k with k leading zeros
    R = (np.arange(7, dtype=np.float32) + 1).reshape(1, 1, 1, 7)
    out = O.cost_volume_forward(R, R, np.array([[float(k)]], np.float32), 1)[0, 1, 0, 0]
    exp = np.zeros(7, np.float32)
    if k < 7:
        exp[k:] = R[0, 0, 0, : 7 - k]
    assert np.array_equal(out, exp)


def test_downsample_two():
    # (v) ds=2: left[..., 2h, 2w]; x = 2w - shift sampled on row 2h  (.cu:76-91)
    r = rng(3)
    L = r.standard_normal((1, 2, 4, 8)).astype(np.float32)
    R = r.standard_normal((1, 2, 4, 8)).astype(np.float32)
    s = np.array([[0.0, 1.0, 2.5]], np.float32)
    out = O.cost_volume_forward(L, R, s, 2)
    assert out.shape == (1, 4, 3, 2, 4)
    assert np.array_equal(out[0, :2, 1], L[0, :, ::2, ::2])
    # d=1 (shift 1): x = 2w-1 -> w=0 gated out, others R[2h, 2w-1]
    exp = np.zeros((2, 2, 4), np.float32)
    exp[:, :, 1:] = R[0, :, ::2, 1::2][:, :, :3]
    assert np.array_equal(out[0, 2:, 1], exp)
    # d=2 (shift 2.5): x = 2w-2.5 -> w>=2: 0.5*R[2w-3]+0.5*R[2w-2]
    exp = np.zeros((2, 2, 4), np.float32)
    for w in (2, 3):
        exp[:, :, w] = 0.5 * R[0, :, ::2, 2 * w - 3] + 0.5 * R[0, :, ::2, 2 * w - 2]
    assert np.allclose(out[0, 2:, 2], exp, rtol=0, atol=1e-6)


def test_requires_divisible():
    with pytest.raises(RuntimeError):
        O.cost_volume_forward(np.zeros((1, 1, 3, 4), np.float32), np.zeros((1, 1, 3, 4), np.float32),
                              np.zeros((1, 1), np.float32), 2)


def test_shape_errors():
    z = np.zeros((1, 1, 2, 2), np.float32)
    with pytest.raises(RuntimeError, match="match their size"):
        O.cost_volume_forward(z, np.zeros((1, 1, 2, 3), np.float32), np.zeros((1, 1), np.float32), 1)
    with pytest.raises(RuntimeError, match="same batch"):
        O.cost_volume_forward(z, z, np.zeros((2, 1), np.float32), 1)


def test_empty():
    z = np.zeros((0, 3, 4, 4), np.float32)
    assert O.cost_volume_forward(z, z, np.zeros((0, 5), np.float32), 1).shape == (0, 6, 5, 4, 4)


def _grid_sample_expr(R, shift):
    """(vii) independent expression: masked 1-row grid_sample(align_corners=True)."""
    N, C, H, W = R.shape
    D = shift.shape[1]
    Rt = torch.from_numpy(R).double()
    out = torch.zeros(N, C, D, H, W, dtype=torch.float64)
    for d in range(D):
        x = torch.arange(W, dtype=torch.float64)[None, :] - torch.from_numpy(shift[:, d]).double()[:, None]  # [N,W]
        mask = (x >= 0) & (x <= W - 1)
        gx = x / (W - 1) * 2 - 1 if W > 1 else torch.zeros_like(x)
        gy = (torch.arange(H, dtype=torch.float64) / max(H - 1, 1) * 2 - 1) if H > 1 else torch.zeros(H, dtype=torch.float64)
        grid = torch.stack([gx[:, None, :].expand(N, H, W), gy[None, :, None].expand(N, H, W)], -1)
        samp = F.grid_sample(Rt, grid, mode="bilinear", padding_mode="border", align_corners=True)
        out[:, :, d] = samp * mask[:, None, None, :]
    return out.numpy()


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-6), (np.float64, 1e-12)])
def test_against_grid_sample_expression(dtype, tol):
    r = rng(7)
    N, C, H, W, D = 2, 3, 4, 19, 9
    L = r.standard_normal((N, C, H, W)).astype(dtype)
    R = r.standard_normal((N, C, H, W)).astype(dtype)
    shift = (r.random((N, D)) * 22).astype(dtype)
    shift[0, 0] = 0.0
    shift[1, 1] = 3.0
    out = O.cost_volume_forward(L, R, shift, 1)
    exp = _grid_sample_expr(R.astype(np.float64), shift.astype(np.float64))
    assert np.allclose(out[:, C:], exp, rtol=0, atol=tol * 10)
    assert np.array_equal(out[:, :C], np.broadcast_to(L[:, :, None], (N, C, D, H, W)))


def test_backward_left_is_sum_over_d_and_adjoint():
    # (vi) gL == sum_d g_L ; <fwd(R), g> == <R, bwd(g)> in fp64
    r = rng(11)
    N, C, H, W, D, ds = 2, 2, 3, 11, 6, 1
    L = r.standard_normal((N, C, H, W))
    R = r.standard_normal((N, C, H, W))
    shift = r.random((N, D)) * 12
    shift[0, 0] = 0.0
    shift[0, 1] = 2.0
    g = r.standard_normal((N, 2 * C, D, H, W))
    out = O.cost_volume_forward(L, R, shift, ds)
    gL, gR = O.cost_volume_backward(g, shift, ds)
    assert np.allclose(gL, g[:, :C].sum(2), rtol=0, atol=1e-12)
    lhs = (out[:, C:] * g[:, C:]).sum()
    rhs = (R * gR).sum()
    assert abs(lhs - rhs) < 1e-10
    lhs = (out[:, :C] * g[:, :C]).sum()
    assert abs(lhs - (L * gL).sum()) < 1e-10


def test_backward_downsample_lattice():
    r = rng(12)
    N, C, H, W, D, ds = 1, 2, 2, 5, 3, 2
    g = r.standard_normal((N, 2 * C, D, H, W))
    shift = np.array([[0.0, 1.0, 3.5]])
    gL, gR = O.cost_volume_backward(g, shift, ds)
    assert gL.shape == (N, C, H * ds, W * ds)
    off = np.ones((H * ds, W * ds), bool)
    off[::ds, ::ds] = False
    assert np.all(gL[:, :, off] == 0)
    assert np.allclose(gL[:, :, ::ds, ::ds], g[:, :C].sum(2))
    assert np.all(gR[:, :, 1::2] == 0)  # only rows h*ds receive gradient (ly = 0)
    # adjoint identity on the ds lattice
    R = r.standard_normal((N, C, H * ds, W * ds))
    out = O.cost_volume_forward(R, R, shift, ds)
    assert abs((out[:, C:] * g[:, C:]).sum() - (R * gR).sum()) < 1e-10


def test_backward_tiny_weight_gate():
    # tap 2 skipped when lx < 1e-10 (.cu:197): integer shift puts everything on tap 1
    g = np.ones((1, 2, 1, 1, 4))
    gL, gR = O.cost_volume_backward(g, np.array([[1.0]]), 1)
    assert gR[0, 0, 0].tolist() == [1.0, 1.0, 1.0, 0.0]
