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
