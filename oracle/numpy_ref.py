"""numpy restatements of small index / gather routines -- TEST INFRASTRUCTURE (see oracle/__init__.py).

sample_2d_feat: VernierScale._sample_2d_feat (snvc/models/vernier.py:323-349) written out
operation by operation in float32, following ATen's CPU grid_sampler_2d for
(bilinear, zeros, align_corners=False) -- aten/src/ATen/native/cpu/GridSamplerKernel.cpp, pinned
upstream at pytorch 1.9.0 (spec-file.txt:20); the algorithm is unchanged in 2.10:
    unnormalise:  ix = (x + 1) * (W / 2) - 0.5
    weights    :  w = ix - floor(ix), e = 1 - w, n = iy - floor(iy), s = 1 - n
                  nw = s*e, ne = s*w, sw = n*e, se = n*w
    value      :  nw_val*nw + ne_val*ne + sw_val*sw + se_val*se, out-of-range taps read as 0
It is checked against torch.nn.functional.grid_sample in tests/test_oracle_numpy_ref.py and is the
bit-exact target of the HIP gather kernel (whose arithmetic is written the same way, contraction off).
"""
import numpy as np

f32 = np.float32


def _taps(px, py, res_x, res_y, hf, wf):
    gx = (px / f32(res_x) * f32(2) - f32(1)).astype(f32)
    gy = (py / f32(res_y) * f32(2) - f32(1)).astype(f32)
    x = ((gx + f32(1)) * (f32(wf) / f32(2)) - f32(0.5)).astype(f32)
    y = ((gy + f32(1)) * (f32(hf) / f32(2)) - f32(0.5)).astype(f32)
    xf, yf = np.floor(x), np.floor(y)
    w = (x - xf).astype(f32)
    e = (f32(1) - w).astype(f32)
    n = (y - yf).astype(f32)
    s = (f32(1) - n).astype(f32)
    wts = [(s * e).astype(f32), (s * w).astype(f32), (n * e).astype(f32), (n * w).astype(f32)]
    ok = np.isfinite(xf) & np.isfinite(yf) & (xf >= -2) & (xf <= wf + 1) & (yf >= -2) & (yf <= hf + 1)
    x0 = np.where(ok, xf, -2).astype(np.int64)
    y0 = np.where(ok, yf, -2).astype(np.int64)
    offs = []
    for dy, dx in ((0, 0), (0, 1), (1, 0), (1, 1)):
        xx, yy = x0 + dx, y0 + dy
        valid = (xx >= 0) & (xx < wf) & (yy >= 0) & (yy < hf)
        offs.append(np.where(valid, yy * wf + xx, -1))
    return offs, wts


def sample_2d_feat(left, right, l_pts, r_pts, resolution):
    """left,right [N,F,Hf,Wf] f32; l_pts,r_pts [N,2,V] -> [N,2F,V] f32."""
    n, f, hf, wf = left.shape
    v = l_pts.shape[2]
    out = np.empty((n, 2 * f, v), dtype=f32)
    for side, (feat, pts) in enumerate(((left, l_pts), (right, r_pts))):
        for b in range(n):
            offs, wts = _taps(pts[b, 0].astype(f32), pts[b, 1].astype(f32), resolution[1], resolution[0], hf, wf)
            planes = feat[b].reshape(f, hf * wf).astype(f32)
            acc = None
            for o, wt in zip(offs, wts):
                val = np.where(o[None, :] >= 0, planes[:, np.maximum(o, 0)], f32(0)).astype(f32)
                term = (val * wt[None, :]).astype(f32)
                acc = term if acc is None else (acc + term).astype(f32)
            out[b, side * f:(side + 1) * f] = acc
    return out
