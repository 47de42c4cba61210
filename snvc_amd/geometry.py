"""Device-side producer of the path's coordinates.

Drop-in for the three ``refinementDataset`` methods that turn box proposals into the
``grid_proj_left`` / ``grid_proj_right`` tensors VernierScale consumes
(snvc/dataset/KITTIRefinement_dataset.py:267-282 ``_init_3d_grid``, :828-846 ``_to_cam``,
:848-868 ``_generate_grid_proj``).  The reference computes them with numpy float64 on the host
(786 k points x 2 cameras per instance) and ships 2 x 6.3 MB per instance to the GPU; here they are
generated where they are used, from 31 doubles per instance.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check


class GridProjector:
    """``cfg`` needs ``x_range``, ``y_range``, ``z_range`` and ``grid_resolution`` = (nh, nw, nl),
    the attributes ``_init_3d_grid`` reads (KITTIRefinement_dataset.py:271-276)."""

    def __init__(self, cfg):
        self.ranges = np.array([cfg.x_range[0], cfg.x_range[1], cfg.y_range[0], cfg.y_range[1],
                                cfg.z_range[0], cfg.z_range[1]], dtype=np.float64)
        self.nh, self.nw, self.nl = (int(v) for v in cfg.grid_resolution)

    @property
    def num_points(self):
        return self.nh * self.nw * self.nl

    def generate(self, samples, P_left, P_right, trans_l, trans_r, device, with_grid_3d=False):
        """samples [N,7] (h,w,l,x,y,z,ry); P_left/P_right [3,4] (calib_left.P / calib_right.P);
        trans_l/trans_r [N,2,3] (meta_roi['trans_l'/'trans_r']).  numpy or torch inputs.
        Returns (coord_l, coord_r[, grid_3d]) like ``_generate_grid_proj``: float32 [N,2,V] on
        ``device`` (and float64 [N,V,3])."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("GridProjector.generate needs a GPU device: Not implemented on the CPU")

        def dev64(a, shape):
            t = torch.as_tensor(np.asarray(a, dtype=np.float64) if not torch.is_tensor(a) else a, dtype=torch.float64)
            t = t.reshape(shape).contiguous()
            return t.to(device)

        n = len(samples)
        s = dev64(samples, (n, 7))
        pl, pr = dev64(P_left, (3, 4)), dev64(P_right, (3, 4))
        tl, tr = dev64(trans_l, (n, 2, 3)), dev64(trans_r, (n, 2, 3))
        v = self.num_points
        out_l = torch.empty((n, 2, v), dtype=torch.float32, device=device)
        out_r = torch.empty_like(out_l)
        g3 = torch.empty((n, v, 3), dtype=torch.float64, device=device) if with_grid_3d else None
        if n:
            p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
            with torch.cuda.device(device):
                check(_lib.lib().snvc_grid_projection(
                    p(s), p(pl), p(pr), p(tl), p(tr), self.ranges.ctypes.data_as(ctypes.c_void_p), self.nh, self.nw,
                    self.nl, p(out_l), p(out_r), p(g3), n,
                    ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "snvc_grid_projection")
        return (out_l, out_r, g3) if with_grid_3d else (out_l, out_r)
