"""ctypes bindings of the C oracle (``oracle/_build/liboracle.so``).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  All functions take and
return numpy arrays; nothing here touches a GPU.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatements with gcc (seconds).  Returns the .so path."""
    srcs = [os.path.join(_HERE, f) for f in ("cost_volume_ref.c", "roiaware_pool3d_ref.c")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


_I64 = ctypes.c_int64
_SUFFIX = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}


def cost_volume_forward(left, right, shift, downsample: int):
    """``build_cost_volume_forward`` semantics (BuildCostVolume_cuda.cu:208-256)."""
    left = np.ascontiguousarray(left)
    right = np.ascontiguousarray(right, dtype=left.dtype)
    shift = np.ascontiguousarray(shift, dtype=left.dtype)
    if left.shape != right.shape:
        raise RuntimeError("Left image and right image should match their size.")
    if left.shape[0] != shift.shape[0]:
        raise RuntimeError("Image and shift should of same batch.")
    n, c, hi, wi = left.shape
    d = shift.shape[1]
    out = np.empty((n, 2 * c, d, hi // downsample, wi // downsample), dtype=left.dtype)
    if out.size == 0:
        return out
    fn = getattr(lib(), "oracle_cost_volume_forward_" + _SUFFIX[left.dtype])
    rc = fn(_p(left), _p(right), _p(shift), _p(out),
            _I64(n), _I64(c), _I64(hi), _I64(wi), _I64(d), _I64(downsample))
    if rc != 0:
        raise RuntimeError("oracle_cost_volume_forward: unusable shapes (H, W must be multiples of downsample)")
    return out


def cost_volume_backward(grad, shift, downsample: int):
    """``build_cost_volume_backward`` semantics (BuildCostVolume_cuda.cu:259-303)."""
    grad = np.ascontiguousarray(grad)
    shift = np.ascontiguousarray(shift, dtype=grad.dtype)
    n, c2, d_, h, w = grad.shape
    c = c2 // 2
    d = shift.shape[1]
    if grad.size != n * c * 2 * d * h * w:
        raise RuntimeError("grad shape is wrong")
    gl = np.zeros((n, c, h * downsample, w * downsample), dtype=grad.dtype)
    gr = np.zeros_like(gl)
    if grad.size == 0:
        return gl, gr
    fn = getattr(lib(), "oracle_cost_volume_backward_" + _SUFFIX[grad.dtype])
    rc = fn(_p(grad), _p(shift), _p(gl), _p(gr),
            _I64(n), _I64(c), _I64(h), _I64(w), _I64(d), _I64(downsample))
    if rc != 0:
        raise RuntimeError("oracle_cost_volume_backward failed")
    return gl, gr


def roiaware_mask(rois, pts, out_size):
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    ox, oy, oz = out_size
    mask = np.empty((rois.shape[0], pts.shape[0]), dtype=np.int32)
    lib().oracle_roiaware_mask(_p(rois), _p(pts), _p(mask), rois.shape[0], pts.shape[0], ox, oy, oz)
    return mask


def roiaware_pool3d_forward(rois, pts, feat, out_size, max_pts_each_voxel=128, pool_method="max"):
    """Returns (pooled [B,ox,oy,oz,C] f32, argmax int32, pts_idx_of_voxels int32)."""
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    ox, oy, oz = out_size
    assert ox < 256 and oy < 256 and oz < 256  # roiaware_pool3d.cpp:53
    b, p, c = rois.shape[0], pts.shape[0], feat.shape[1]
    pooled = np.zeros((b, ox, oy, oz, c), dtype=np.float32)
    argmax = np.zeros((b, ox, oy, oz, c), dtype=np.int32)
    lists = np.zeros((b, ox, oy, oz, max_pts_each_voxel), dtype=np.int32)
    scratch = np.empty((b, p), dtype=np.int32)
    lib().oracle_roiaware_pool3d_forward(
        _p(rois), _p(pts), _p(feat), _p(argmax), _p(lists), _p(pooled),
        b, p, c, max_pts_each_voxel, ox, oy, oz, {"max": 0, "avg": 1}[pool_method], _p(scratch))
    return pooled, argmax, lists


def roiaware_pool3d_backward(lists, argmax, grad_out, num_pts, pool_method="max"):
    lists = np.ascontiguousarray(lists, dtype=np.int32)
    argmax = np.ascontiguousarray(argmax, dtype=np.int32)
    grad_out = np.ascontiguousarray(grad_out, dtype=np.float32)
    b, ox, oy, oz, max_pts = lists.shape
    c = grad_out.shape[-1]
    grad_in = np.zeros((num_pts, c), dtype=np.float32)
    lib().oracle_roiaware_pool3d_backward(
        _p(lists), _p(argmax), _p(grad_out), _p(grad_in),
        b, c, max_pts, ox, oy, oz, {"max": 0, "avg": 1}[pool_method])
    return grad_in


def points_in_boxes_gpu(points, boxes):
    """points [Bs,M,3], boxes [Bs,T,7] -> [Bs,M] int32, -1 = background."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    bs, m, _ = points.shape
    out = np.full((bs, m), -1, dtype=np.int32)
    lib().oracle_points_in_boxes_gpu(_p(boxes), _p(points), _p(out), bs, boxes.shape[1], m)
    return out


def points_in_boxes_cpu(points, boxes):
    """points [M,3], boxes [T,7] -> [T,M] int32 flags (margin 1e-2)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    out = np.zeros((boxes.shape[0], points.shape[0]), dtype=np.int32)
    lib().oracle_points_in_boxes_cpu(_p(boxes), _p(points), _p(out), boxes.shape[0], points.shape[0])
    return out
