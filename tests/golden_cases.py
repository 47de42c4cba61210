"""Case table shared by tests/golden/make_golden.py (generator, needs the reference) and the
parity tests (consumers, need only tests/golden/reference_outputs.npz).

Inputs and parameters are never stored: they are re-drawn from ``numpy.random.default_rng``
with the seeds below (numpy's Generator stream is stable across versions).
"""
import os

import numpy as np
import torch

GOLDEN_NPZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz")

RESOLUTION = (64, 64)  # cfg.resolution of the synthetic square RoI (last bullet)

# name: (cin, cout, k, stride, pad, dilation, gn, spatial (D,H,W), seed)
CONV_CASES = {
    "k1_64to32": (64, 32, 1, 1, 0, 1, False, (4, 6, 40), 100),
    "k3_64to32": (64, 32, 3, 1, 1, 1, False, (4, 6, 40), 110),
    "k3_32to32_gn": (32, 32, 3, 1, 1, 1, True, (4, 6, 40), 120),
    "k3s2_32to64": (32, 64, 3, 2, 1, 1, False, (4, 8, 40), 130),
    "k3s2_64to64_gn": (64, 64, 3, 2, 1, 1, True, (4, 8, 24), 140),
    "k5_32to32": (32, 32, 5, 1, 2, 1, False, (5, 6, 36), 150),
    "k5d2_32to32": (32, 32, 5, 1, 4, 2, False, (9, 9, 36), 160),
    "k7_64to32": (64, 32, 7, 1, 3, 1, False, (7, 7, 34), 170),
    "k3_odd_w": (32, 32, 3, 1, 1, 1, False, (3, 5, 13), 180),
}

# name: (C, gn, spatial, seed)
HOURGLASS_CASES = {
    "c32": (32, False, (8, 8, 36), 200),
    "c32_gn": (32, True, (4, 8, 16), 210),
}
HOURGLASS16_CASES = {
    "c32": (32, False, (16, 16, 48), 300),
    "c32_gn": (32, True, (16, 16, 32), 310),
}

# name: (grid (nh,nw,nl), gn, N, feat H, feat W, seed)
TRUNK_CASES = {
    "G1": ((16, 16, 24), False, 2, 16, 16, 400),      # plain hourglass + hourglass2d (nw <= 16)
    "G1_gn": ((16, 16, 24), True, 1, 16, 16, 410),
    "G2": ((16, 32, 48), False, 1, 16, 16, 420),      # hourglass_downsample_16
    "G2_gn": ((16, 32, 48), True, 1, 16, 16, 430),
}

# vernier_type='BEV_type2' (reference vernier.py:191-248, :391-410): the same 3D trunk without the coordinate head; nw > 16 only (the
# reference's own forward fails on the plain hourglass there)
TYPE2_CASES = {
    "T2": ((32, 32, 48), False, 1, 16, 16, 440),
    "T2_gn": ((32, 32, 48), True, 1, 16, 16, 450),
}

# name: (C, spatial (D,H,W), seed)
GLOBAL_CASES = {
    "c32_small": (32, (8, 12, 40), 500),
}


def randn(shape, seed, dtype=np.float32):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(dtype))


def trunk_inputs(n, f, fh, fw, grid, seed):
    """left/right feature maps [n,f,fh,fw] and RoI-pixel projections [n,2,V]; about 6 % of the
    projections fall outside the crop to exercise zero padding."""
    rng = np.random.default_rng(seed)
    nh, nw, nl = grid
    v = nh * nw * nl
    lf = torch.from_numpy(rng.standard_normal((n, f, fh, fw)).astype(np.float32))
    rf = torch.from_numpy(rng.standard_normal((n, f, fh, fw)).astype(np.float32))
    lo, hi = -0.03 * RESOLUTION[0], 1.03 * RESOLUTION[0]
    gpl = torch.from_numpy(rng.uniform(lo, hi, (n, 2, v)).astype(np.float32))
    gpr = torch.from_numpy(rng.uniform(lo, hi, (n, 2, v)).astype(np.float32))
    return lf, rf, gpl, gpr


def load_golden():
    return np.load(GOLDEN_NPZ)


def grid_proj_case():
    """a11: synthetic KITTI-like calibration + three box proposals + RoI-crop affines."""
    P2 = np.array([7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01,
                   0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01,
                   0.0, 0.0, 1.0, 2.745884e-03]).reshape(3, 4)
    P3 = P2.copy()
    P3[0, 3] = -3.395242e+02
    P3[1, 3] = 2.199936e+00
    samples = np.array([[1.52, 1.63, 3.88, 2.10, 1.65, 14.2, -1.52],
                        [1.60, 1.70, 4.10, -6.30, 1.80, 28.7, 0.31],
                        [1.45, 1.58, 3.60, 0.40, 1.55, 8.9, 1.62]])
    r = np.random.default_rng(777)
    trans_l = np.zeros((3, 2, 3))
    trans_r = np.zeros((3, 2, 3))
    for t in (trans_l, trans_r):
        t[:, 0, 0] = r.uniform(1.5, 3.0, 3)
        t[:, 1, 1] = r.uniform(1.5, 3.0, 3)
        t[:, 0, 2] = r.uniform(-1500, -200, 3)
        t[:, 1, 2] = r.uniform(-500, -100, 3)
        t[:, 0, 1] = r.uniform(-0.01, 0.01, 3)
    return dict(P_left=P2, P_right=P3, samples=samples, trans_l=trans_l, trans_r=trans_r,
                x_range=(-1.6, 1.6), y_range=(-0.8, 0.8), z_range=(-2.4, 2.4), grid=(16, 32, 48))
