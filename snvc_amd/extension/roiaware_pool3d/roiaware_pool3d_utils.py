"""Drop-in for ``snvc.extension.roiaware_pool3d.roiaware_pool3d_utils`` (reference :1-151).

The reference file cannot be imported as shipped (it imports a non-existent
``snvc.utils.common_utils``, :6); the two helpers it needs from there are restated locally.
``roiaware_pool3d_cuda`` keeps the four names of the pybind module (roiaware_pool3d.cpp:172-177).
"""
import types

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from ... import ops

roiaware_pool3d_cuda = types.SimpleNamespace(
    forward=ops.roiaware_pool3d_forward,
    backward=ops.roiaware_pool3d_backward,
    points_in_boxes_gpu=ops.points_in_boxes_gpu,
    points_in_boxes_cpu=ops.points_in_boxes_cpu,
)


def _check_numpy_to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def points_in_boxes_cpu(points, boxes):
    """points (num_points, 3), boxes (N, 7) [x,y,z,dx,dy,dz,heading] -> (N, num_points) 0/1 flags."""
    assert boxes.shape[1] == 7
    assert points.shape[1] == 3
    points, is_numpy = _check_numpy_to_torch(points)
    boxes, is_numpy = _check_numpy_to_torch(boxes)
    point_indices = points.new_zeros((boxes.shape[0], points.shape[0]), dtype=torch.int)
    roiaware_pool3d_cuda.points_in_boxes_cpu(boxes.float().contiguous(), points.float().contiguous(), point_indices)
    return point_indices.numpy() if is_numpy else point_indices


def points_in_boxes_cpu_idmap(points, boxes):
    points, is_numpy = _check_numpy_to_torch(points)
    boxes, is_numpy = _check_numpy_to_torch(boxes)
    if len(boxes) > 0 and len(points) > 0:
        point_indices = points_in_boxes_cpu(points, boxes)
        point_indices[point_indices == 0] = -1
        for i in range(boxes.shape[0]):
            point_indices[i, point_indices[i] == 1] = i
        point_indices = point_indices.max(0).values
    else:
        point_indices = torch.full((len(points),), -1, dtype=torch.int32)
    return point_indices.numpy() if is_numpy else point_indices


def points_in_boxes_gpu(points, boxes):
    """points (B, M, 3), boxes (B, T, 7) -> (B, M) index of the first containing box, -1 = none."""
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    batch_size, num_points, _ = points.shape
    box_idxs_of_pts = points.new_zeros((batch_size, num_points), dtype=torch.int).fill_(-1)
    roiaware_pool3d_cuda.points_in_boxes_gpu(boxes.contiguous(), points.contiguous(), box_idxs_of_pts)
    return box_idxs_of_pts


class RoIAwarePool3d(nn.Module):
    def __init__(self, out_size, max_pts_each_voxel=128):
        super().__init__()
        self.out_size = out_size
        self.max_pts_each_voxel = max_pts_each_voxel

    def forward(self, rois, pts, pts_feature, pool_method='max'):
        assert pool_method in ['max', 'avg']
        return RoIAwarePool3dFunction.apply(rois, pts, pts_feature, self.out_size, self.max_pts_each_voxel, pool_method)


class RoIAwarePool3dFunction(Function):
    @staticmethod
    def forward(ctx, rois, pts, pts_feature, out_size, max_pts_each_voxel, pool_method):
        """rois (N,7), pts (npoints,3), pts_feature (npoints,C) -> pooled (N,ox,oy,oz,C)."""
        assert rois.shape[1] == 7 and pts.shape[1] == 3
        if isinstance(out_size, int):
            out_x = out_y = out_z = out_size
        else:
            assert len(out_size) == 3
            for k in range(3):
                assert isinstance(out_size[k], int)
            out_x, out_y, out_z = out_size
        num_rois = rois.shape[0]
        num_channels = pts_feature.shape[-1]
        num_pts = pts.shape[0]
        pooled_features = pts_feature.new_zeros((num_rois, out_x, out_y, out_z, num_channels))
        argmax = pts_feature.new_zeros((num_rois, out_x, out_y, out_z, num_channels), dtype=torch.int)
        pts_idx_of_voxels = pts_feature.new_zeros((num_rois, out_x, out_y, out_z, max_pts_each_voxel), dtype=torch.int)
        pool_method = {'max': 0, 'avg': 1}[pool_method]
        roiaware_pool3d_cuda.forward(rois.contiguous(), pts.contiguous(), pts_feature.contiguous(), argmax,
                                     pts_idx_of_voxels, pooled_features, pool_method)
        ctx.roiaware_pool3d_for_backward = (pts_idx_of_voxels, argmax, pool_method, num_pts, num_channels)
        return pooled_features

    @staticmethod
    def backward(ctx, grad_out):
        pts_idx_of_voxels, argmax, pool_method, num_pts, num_channels = ctx.roiaware_pool3d_for_backward
        grad_in = grad_out.new_zeros((num_pts, num_channels))
        roiaware_pool3d_cuda.backward(pts_idx_of_voxels, argmax, grad_out.contiguous(), grad_in, pool_method)
        return None, None, grad_in, None, None, None
