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


# --------------------------------------------------------------------------------------------
# a11: the host-side producer of the path's coordinates
# --------------------------------------------------------------------------------------------
def init_3d_grid(x_range, y_range, z_range, grid_resolution):
    """refinementDataset._init_3d_grid (snvc/dataset/KITTIRefinement_dataset.py:267-282):
    grid_3d [3, nh, nw, nl] float64, voxel order (ih*nw + iw)*nl + il."""
    nh, nw, nl = grid_resolution
    x_pts = np.linspace(x_range[0], x_range[1], nw)
    y_pts = np.linspace(y_range[0], y_range[1], nh)
    z_pts = np.linspace(z_range[0], z_range[1], nl)
    gx, gy, gz = np.meshgrid(x_pts, y_pts, z_pts, indexing="xy")
    return np.concatenate([gx[None], gy[None], gz[None]])


def grid_projection(samples, P_left, P_right, trans_l, trans_r, grid_3d):
    """refinementDataset._to_cam + _generate_grid_proj (KITTIRefinement_dataset.py:828-868) with
    Calibration.project_rect_to_image (dataset/kitti_util.py:282-293) and affine_transform
    (utils/img_proc.py:71-74).  samples [N,7] = (h,w,l,x,y,z,ry); P_* [3,4]; trans_* [N,2,3].
    Returns (coord_l [N,2,V] f32, coord_r [N,2,V] f32, grid_cam [N,V,3] f64)."""
    pts = grid_3d.reshape(3, -1)
    cl, cr, g3 = [], [], []
    for i, s in enumerate(samples):
        ry = s[6] + 0.5 * np.pi
        rot = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
        x, y, z = s[3:6]
        cam = (rot @ pts + np.array([[x], [y - s[0] * 0.5], [z]])).T          # [V,3]
        g3.append(cam[None])
        hom = np.hstack([cam, np.ones((len(cam), 1))])
        for P, tr, dst in ((P_left, trans_l[i], cl), (P_right, trans_r[i], cr)):
            p2 = hom @ P.T
            uv = p2[:, :2] / p2[:, 2:3]
            h2 = np.concatenate([uv, np.ones((len(uv), 1))], axis=1)
            dst.append((tr @ h2.T).astype(np.float32)[None])
    return np.concatenate(cl), np.concatenate(cr), np.concatenate(g3)
