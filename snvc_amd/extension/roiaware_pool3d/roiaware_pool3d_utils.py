"""Host-side face of ``snvc.extension.roiaware_pool3d`` on the HIP kernels.

Written against the reference module's INTERFACE (snvc/extension/roiaware_pool3d/
roiaware_pool3d_utils.py: the public names, their argument order and what each returns), not its
bodies.  The reference file cannot even be imported as shipped (``snvc.utils.common_utils`` does
not exist, :6); the numpy<->torch convenience it wanted from there is ``_as_tensor`` below.

    points_in_boxes_cpu(points, boxes)        -> (num_boxes, num_points) int 0/1 flags        (:10-26)
    points_in_boxes_cpu_idmap(points, boxes)  -> (num_points,) largest containing box, -1     (:29-44)
    points_in_boxes_gpu(points, boxes)        -> (B, M) first containing box, -1              (:68-81)
    RoIAwarePool3d(out_size, max_pts)(rois, pts, pts_feature, pool_method)                    (:84-93)
    RoIAwarePool3dFunction.apply(rois, pts, pts_feature, out_size, max_pts, pool_method)      (:96-147)

``depth_map_in_boxes_cpu`` (:47-65) needs a KITTI ``calib`` object from the dataset layer and is
outside the path.  ``roiaware_pool3d_cuda`` carries the four names of the pybind module
(roiaware_pool3d.cpp:172-177), bound to the C-ABI wrappers in ``snvc_amd.ops``.
"""
import types

import numpy as np
import torch
import torch.nn as nn

from ... import ops

roiaware_pool3d_cuda = types.SimpleNamespace(
    forward=ops.roiaware_pool3d_forward,
    backward=ops.roiaware_pool3d_backward,
    points_in_boxes_gpu=ops.points_in_boxes_gpu,
    points_in_boxes_cpu=ops.points_in_boxes_cpu,
)

_POOL_METHODS = ("max", "avg")      # position = the integer the native op takes


def _as_tensor(a):
    """(float tensor, came_from_numpy)"""
    if isinstance(a, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)), True
    return a, False


def _back(t, numpy_out):
    return t.numpy() if numpy_out else t


def points_in_boxes_cpu(points, boxes):
    """points (num_points, 3), boxes (N, 7) = [x, y, z, dx, dy, dz, heading] -> (N, num_points)
    int32 flags; a host function in the reference as well (roiaware_pool3d.cpp:137-168)."""
    if points.shape[1] != 3 or boxes.shape[1] != 7:
        raise AssertionError("points must be (num_points, 3) and boxes (N, 7)")
    pts, from_np_p = _as_tensor(points)
    bxs, from_np_b = _as_tensor(boxes)
    flags = torch.zeros((bxs.shape[0], pts.shape[0]), dtype=torch.int32)
    if flags.numel():
        roiaware_pool3d_cuda.points_in_boxes_cpu(bxs.float().contiguous(), pts.float().contiguous(), flags)
    return _back(flags, from_np_p or from_np_b)


def points_in_boxes_cpu_idmap(points, boxes):
    """Per point: the highest index among the boxes that contain it, -1 for background."""
    pts, from_np_p = _as_tensor(points)
    bxs, from_np_b = _as_tensor(boxes)
    n_box, n_pts = len(bxs), len(pts)
    if n_box == 0 or n_pts == 0:
        ids = torch.full((n_pts,), -1, dtype=torch.int32)
    else:
        inside = points_in_boxes_cpu(pts, bxs) > 0                               # (n_box, n_pts)
        box_no = torch.arange(n_box, dtype=torch.int32).unsqueeze(1).expand(n_box, n_pts)
        ids = torch.where(inside, box_no, torch.full_like(box_no, -1)).amax(dim=0)
    return _back(ids, from_np_p or from_np_b)


def points_in_boxes_gpu(points, boxes):
    """points (B, M, 3), boxes (B, T, 7), both on the GPU -> (B, M) int32: index of the first box
    (ascending) containing the point, -1 = none (roiaware_pool3d_kernel.cu:313-336)."""
    if points.dim() != 3 or boxes.dim() != 3 or points.shape[0] != boxes.shape[0]:
        raise AssertionError("points (B, M, 3) and boxes (B, T, 7) must share the batch dimension")
    if points.shape[2] != 3 or boxes.shape[2] != 7:
        raise AssertionError("points (B, M, 3), boxes (B, T, 7)")
    owner = torch.full(points.shape[:2], -1, dtype=torch.int32, device=points.device)
    roiaware_pool3d_cuda.points_in_boxes_gpu(boxes.contiguous(), points.contiguous(), owner)
    return owner


def _out_dims(out_size):
    if isinstance(out_size, int):
        return out_size, out_size, out_size
    dims = tuple(out_size)
    if len(dims) != 3 or not all(isinstance(d, int) for d in dims):
        raise AssertionError("out_size must be an int or three ints")
    return dims


class RoIAwarePool3dFunction(torch.autograd.Function):
    """rois (N, 7), pts (npoints, 3), pts_feature (npoints, C) -> pooled (N, ox, oy, oz, C).
    Only ``pts_feature`` receives a gradient."""

    @staticmethod
    def forward(ctx, rois, pts, pts_feature, out_size, max_pts_each_voxel, pool_method):
        if rois.shape[1] != 7 or pts.shape[1] != 3:
            raise AssertionError("rois (N, 7), pts (npoints, 3)")
        ox, oy, oz = _out_dims(out_size)
        method = _POOL_METHODS.index(pool_method)
        n_rois, n_pts, n_chan = rois.shape[0], pts.shape[0], pts_feature.shape[-1]
        vox = (n_rois, ox, oy, oz)
        # the native op fills pre-zeroed outputs (its contract, roiaware_pool3d.cpp:29-66)
        pooled = torch.zeros(vox + (n_chan,), dtype=pts_feature.dtype, device=pts_feature.device)
        winner = torch.zeros(vox + (n_chan,), dtype=torch.int32, device=pts_feature.device)
        members = torch.zeros(vox + (max_pts_each_voxel,), dtype=torch.int32, device=pts_feature.device)
        roiaware_pool3d_cuda.forward(rois.contiguous(), pts.contiguous(), pts_feature.contiguous(), winner, members,
                                     pooled, method)
        # kept under the reference's attribute name: downstream code inspects it
        ctx.roiaware_pool3d_for_backward = (members, winner, method, n_pts, n_chan)
        return pooled

    @staticmethod
    def backward(ctx, grad_out):
        members, winner, method, n_pts, n_chan = ctx.roiaware_pool3d_for_backward
        grad_feat = torch.zeros((n_pts, n_chan), dtype=grad_out.dtype, device=grad_out.device)
        roiaware_pool3d_cuda.backward(members, winner, grad_out.contiguous(), grad_feat, method)
        return None, None, grad_feat, None, None, None


class RoIAwarePool3d(nn.Module):
    def __init__(self, out_size, max_pts_each_voxel=128):
        super().__init__()
        self.out_size = out_size
        self.max_pts_each_voxel = max_pts_each_voxel

    def forward(self, rois, pts, pts_feature, pool_method="max"):
        if pool_method not in _POOL_METHODS:
            raise AssertionError(f"pool_method must be one of {_POOL_METHODS}")
        return RoIAwarePool3dFunction.apply(rois, pts, pts_feature, self.out_size, self.max_pts_each_voxel, pool_method)
