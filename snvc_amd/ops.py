"""Tensor-level wrappers over the C ABI (``include/snvc_hip.h``).

PyTorch is plumbing here: it owns device memory and the HIP stream.  Every function
  * requires its tensors on a GPU (``RuntimeError("... Not implemented on the CPU")`` otherwise,
    the message the reference's dispatcher gives, BuildCostVolume.cpp:26,41),
  * makes inputs contiguous like the reference launchers do (BuildCostVolume_cuda.cu:243-245),
  * allocates outputs with the input's dtype/device (at::empty / at::zeros there),
  * launches on ``torch.cuda.current_stream()`` without synchronising.
"""
import contextlib
import ctypes
import math
import threading
import time
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (EPI_ADD_POST, EPI_ADD_PRE, EPI_RELU, EPI_SIGMOID, EPI_STREAM_OUT, F32, F64, Conv3dDesc,
                   Unsupported, check)

__all__ = [
    "cost_volume_forward", "cost_volume_forward_right", "cost_volume_backward", "cost_volume_backward_right", "depth_class_sums", "voxel_gather_forward", "voxel_gather_backward",
    "Conv3dLayer", "conv_variant", "conv3d_wgrad", "Conv3dLayerF16", "to_c8", "from_c8", "voxel_gather_forward_f16",
    "mul_broadcast_c8", "avgpool_depth4_c8", "volume_resample", "rect_to_psv_grid", "act_backward_reduce", "act_backward_apply", "bn_backward_coefs", "norm_stats", "affine_act", "mul_broadcast", "avgpool_depth4", "zero_stuff2x",
    "disparity_regression", "argmax_rows", "roiaware_pool3d_forward", "roiaware_pool3d_backward",
    "points_in_boxes_gpu", "points_in_boxes_cpu",
    "EPI_RELU", "EPI_ADD_PRE", "EPI_ADD_POST", "EPI_SIGMOID", "EPI_STREAM_OUT",
]


_variant_bits = 0


@contextlib.contextmanager
def conv_variant(algo_bits: int):
    """Within the block every conv3d / wgrad launch of this PROCESS carries ``algo_bits`` in
    ``snvc_conv3d_desc.algo`` (the SNVC_ALGO_* kernel-form selectors of include/snvc_hip.h):
    how the parity tests and the tuning scripts reach every instantiated kernel form.  Process-wide, not per thread:
    autograd runs ``backward`` on its own worker thread, and a ``loss.backward()`` inside the block must see the bits."""
    global _variant_bits
    prev = _variant_bits
    _variant_bits = int(algo_bits)
    try:
        yield
    finally:
        _variant_bits = prev


def _algo(exact: bool = False) -> int:
    return (_variant_bits | _lib.ALGO_DIRECT) if exact else _variant_bits


def _gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: Not implemented on the CPU")


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dtype_tag(t: torch.Tensor, what: str) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise RuntimeError(f'"{what}" not implemented for {t.dtype}')  # AT_DISPATCH_FLOATING_TYPES


# ------------------------------------------------------------------------------ cost volume
def cost_volume_forward(left, right, shift, downsample: int):
    """build_cost_volume_forward (BuildCostVolume_cuda.cu:208-256)."""
    _gpu(left, "left"); _gpu(right, "right"); _gpu(shift, "shift")
    if left.dim() != 4 or tuple(left.shape) != tuple(right.shape):
        raise RuntimeError("Left image and right image should match their size.")
    if shift.dim() != 2 or left.size(0) != shift.size(0):
        raise RuntimeError("Image and shift should of same batch.")
    tag = _dtype_tag(left, "BuildCostVolume_forward")
    if right.dtype != left.dtype or shift.dtype != left.dtype:
        raise RuntimeError("left, right and shift must share one dtype")
    downsample = int(downsample)
    if downsample < 1:
        raise RuntimeError("downsample must be >= 1")
    n, c, hi, wi = left.shape
    d = shift.size(1)
    if hi % downsample or wi % downsample:
        raise RuntimeError("feature height and width must be multiples of downsample")
    out = torch.empty((n, 2 * c, d, hi // downsample, wi // downsample), dtype=left.dtype, device=left.device)
    if out.numel() == 0:
        return out
    left, right, shift = left.contiguous(), right.contiguous(), shift.contiguous()
    with torch.cuda.device(left.device):
        check(_lib.lib().snvc_cost_volume_forward(_ptr(left), _ptr(right), _ptr(shift), _ptr(out), n, c, hi, wi, d,
                                                  downsample, tag, _stream(left)), "build_cost_volume_forward")
    return out


def cost_volume_forward_right(right, shift, out=None):
    """Right (warped) half of build_cost_volume at downsample 1: [N,C,D,H,W] == full[:, C:]."""
    _gpu(right, "right"); _gpu(shift, "shift")
    if right.dtype != torch.float32 or shift.dtype != torch.float32:
        raise RuntimeError("cost_volume_forward_right is fp32 only")
    n, c, h, w = right.shape
    d = shift.size(1)
    if out is None:
        out = torch.empty((n, c, d, h, w), dtype=torch.float32, device=right.device)
    elif tuple(out.shape) != (n, c, d, h, w) or out.dtype != torch.float32 or not out.is_contiguous():
        raise RuntimeError("cost_volume_forward_right `out` must be a contiguous float32 [N,C,D,H,W] tensor")
    if out.numel() == 0:
        return out
    right, shift = right.contiguous(), shift.contiguous()
    with torch.cuda.device(right.device):
        check(_lib.lib().snvc_cost_volume_forward_right(_ptr(right), _ptr(shift), _ptr(out), n, c, h, w, d, 1,
                                                        _stream(right)), "snvc_cost_volume_forward_right")
    return out


def cost_volume_backward(grad, shift, downsample: int):
    """build_cost_volume_backward (BuildCostVolume_cuda.cu:259-303) -> (grad_left, grad_right)."""
    _gpu(grad, "grad"); _gpu(shift, "shift")
    tag = _dtype_tag(grad, "BuildCostVolume_backward")
    if shift.dtype != grad.dtype:
        raise RuntimeError("grad and shift must share one dtype")
    downsample = int(downsample)
    n, c2, _, h, w = grad.shape
    c, d = c2 // 2, shift.size(1)
    if grad.numel() != n * c * 2 * d * h * w:
        raise RuntimeError("grad shape is wrong")
    gl = torch.empty((n, c, h * downsample, w * downsample), dtype=grad.dtype, device=grad.device)
    gr = torch.empty_like(gl)
    if gl.numel() == 0:
        return gl, gr
    grad, shift = grad.contiguous(), shift.contiguous()
    with torch.cuda.device(grad.device):
        check(_lib.lib().snvc_cost_volume_backward(_ptr(grad), _ptr(shift), _ptr(gl), _ptr(gr), n, c, h, w, d,
                                                   downsample, tag, _stream(grad)), "build_cost_volume_backward")
    return gl, gr


def cost_volume_backward_right(grad_r, shift):
    """Adjoint of cost_volume_forward_right: grad_r [N,C,D,H,W] -> grad_right [N,C,H,W] (fp32, downsample 1)."""
    _gpu(grad_r, "grad"); _gpu(shift, "shift")
    if grad_r.dtype != torch.float32 or shift.dtype != torch.float32 or grad_r.dim() != 5:
        raise RuntimeError("cost_volume_backward_right needs float32 [N,C,D,H,W]")
    grad_r, shift = grad_r.contiguous(), shift.contiguous()
    n, c, d, h, w = grad_r.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=grad_r.device)
    with torch.cuda.device(grad_r.device):
        check(_lib.lib().snvc_cost_volume_backward_right(_ptr(grad_r), _ptr(shift), _ptr(out), n, c, h, w, d, _stream(grad_r)),
              "snvc_cost_volume_backward_right")
    return out


def depth_class_sums(g):
    """g [N,C,D,H,W] -> [N,C,3,H,W]: (g[:, :, 0], g[:, :, 1:-1].sum(2), g[:, :, -1]) -- the adjoint of the depth-class
    planes of the factored first convolution."""
    _gpu(g, "g")
    g = g.contiguous()
    n, c, d, h, w = g.shape
    out = torch.empty((n, c, 3, h, w), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        check(_lib.lib().snvc_depth_class_sums(_ptr(g), _ptr(out), n * c, d, h * w, _stream(g)), "snvc_depth_class_sums")
    return out


# ------------------------------------------------------------------------------ voxel gather
def _check_gather(left, right, l_pts, r_pts):
    for t, nm in ((left, "left"), (right, "right"), (l_pts, "grid_proj_left"), (r_pts, "grid_proj_right")):
        _gpu(t, nm)
        if t.dtype != torch.float32:
            raise RuntimeError(f"{nm} must be float32")
    if left.dim() != 4 or tuple(left.shape) != tuple(right.shape):
        raise RuntimeError("left and right feature maps must have the same [N,F,H,W] shape")
    if l_pts.dim() != 3 or l_pts.size(1) != 2 or tuple(l_pts.shape) != tuple(r_pts.shape) or l_pts.size(0) != left.size(0):
        raise RuntimeError("grid projections must be [N,2,V] for the same batch as the features")


def voxel_gather_forward(left, right, l_pts, r_pts, resolution) -> torch.Tensor:
    """_sample_2d_feat(aggregate='concat') (vernier.py:323-349) -> [N, 2F, V]."""
    _check_gather(left, right, l_pts, r_pts)
    n, f, hf, wf = left.shape
    v = l_pts.size(2)
    out = torch.empty((n, 2 * f, v), dtype=torch.float32, device=left.device)
    if out.numel() == 0:
        return out
    left, right, l_pts, r_pts = left.contiguous(), right.contiguous(), l_pts.contiguous(), r_pts.contiguous()
    with torch.cuda.device(left.device):
        # channels-last copy of both feature maps (<= 4 MB on the path): 4x fewer gather instructions, same bits
        ws = torch.empty(_lib.lib().snvc_voxel_gather_workspace_floats(n, f, hf, wf), dtype=torch.float32,
                         device=left.device) if f % 4 == 0 else None
        check(_lib.lib().snvc_voxel_gather_forward_ws(_ptr(left), _ptr(right), _ptr(l_pts), _ptr(r_pts), _ptr(out),
                                                      _ptr(ws), n, f, hf, wf, v, float(resolution[1]),
                                                      float(resolution[0]), _stream(left)), "voxel_gather_forward")
    return out


def voxel_atten_scale_(vox):
    """In place: vox [N,2F,V] *= clamp(cosine_similarity(left half, right half, dim=1), 0) (vernier.py:341-344)."""
    _gpu(vox, "vox")
    if vox.dtype != torch.float32 or vox.dim() != 3 or vox.size(1) % 2 or not vox.is_contiguous():
        raise RuntimeError("voxel_atten_scale_ needs a contiguous float32 [N,2F,V] tensor")
    n, c2, v = vox.shape
    if vox.numel():
        with torch.cuda.device(vox.device):
            check(_lib.lib().snvc_voxel_atten_scale(_ptr(vox), n, c2 // 2, v, _stream(vox)), "snvc_voxel_atten_scale")
    return vox


def voxel_gather_backward(grad_out, l_pts, r_pts, feat_shape, resolution, deterministic: bool = True):
    """Adjoint of voxel_gather_forward w.r.t. the two feature maps.  deterministic=True (default): the sorted,
    atomics-free form (snvc_voxel_gather_backward_det, bit-reproducible); False: the float-atomics form."""
    n, f, hf, wf = feat_shape
    _gpu(grad_out, "grad_out")
    v = l_pts.size(2)
    gl = torch.empty((n, f, hf, wf), dtype=torch.float32, device=grad_out.device)
    gr = torch.empty_like(gl)
    grad_out, l_pts, r_pts = grad_out.contiguous(), l_pts.contiguous(), r_pts.contiguous()
    with torch.cuda.device(grad_out.device):
        if deterministic:
            nbytes = _lib.lib().snvc_voxel_gather_backward_workspace_bytes(n, f, hf, wf, v)
            if nbytes < 0:
                check(1, "snvc_voxel_gather_backward_workspace_bytes")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=grad_out.device)
            check(_lib.lib().snvc_voxel_gather_backward_det(_ptr(grad_out), _ptr(l_pts), _ptr(r_pts), _ptr(gl), _ptr(gr), _ptr(ws),
                                                            n, f, hf, wf, v, float(resolution[1]), float(resolution[0]),
                                                            _stream(grad_out)), "voxel_gather_backward_det")
        else:
            check(_lib.lib().snvc_voxel_gather_backward(_ptr(grad_out), _ptr(l_pts), _ptr(r_pts), _ptr(gl), _ptr(gr), n, f,
                                                        hf, wf, v, float(resolution[1]), float(resolution[0]),
                                                        _stream(grad_out)), "voxel_gather_backward")
    return gl, gr


# ------------------------------------------------------------------------------ conv3d
def _dense_inner(t: torch.Tensor) -> bool:
    """True if dims 1.. are laid out densely (dim 0 may have any stride): a channel slice of a
    contiguous [N, Ctot, D, H, W] buffer qualifies."""
    exp = 1
    for size, stride in zip(reversed(t.shape[1:]), reversed(t.stride()[1:])):
        if size != 1 and stride != exp:
            return False
        exp *= size
    return True


def _batch_stride(t: torch.Tensor) -> int:
    return t.stride(0) if t.size(0) > 1 else math.prod(t.shape[1:])


class Conv3dLayer:
    """One Conv3d / ConvTranspose3d layer prepared for the HIP kernel: geometry, packed weights.

    Geometry follows nn.Conv3d(cin, cout, k, stride, pad, dilation, bias=False) as built by
    convbn_3d (submodule.py:41-48) or nn.ConvTranspose3d(cin, cout, 3, padding=1,
    output_padding=1, stride=2, bias=False) (submodule.py:127-134,198-205).
    """

    def __init__(self, weight: torch.Tensor, ksize: int, stride: int, pad: int, dilation: int, transposed: bool,
                 planar: bool = False, ksize_h: int = 0):
        """planar=True: a depth-1 layer (desc.ksize_d = 1) -- an nn.Conv2d(k, stride, padding=(k-1)/2) of the 2D BEV
        neck run on [N,C,1,H,W] views; ``weight`` is then [Cout,Cin,1,k,k] (or the Conv2d's own [Cout,Cin,k,k]).
        ksize_h (planar only): kernel extent along H when it differs from ``ksize`` (desc.ksize_h; the 3 x 7 layer of the
        sheared first convolution: weight [Cout,Cin,3,7], padding (1,3))."""
        _gpu(weight, "weight")
        if weight.dtype != torch.float32:
            raise RuntimeError("conv3d weights must be float32")
        self.transposed = bool(transposed)
        self.planar = bool(planar)
        if planar and weight.dim() == 4:
            weight = weight.unsqueeze(2)
        if planar and transposed:
            # nn.ConvTranspose2d(k3,s2,p1,op1) [Cin,Cout,3,3]: the kd = 1 plane of a 3x3x3 transposed kernel; the depth-1
            # form of the parity-class kernel launches only the two classes whose taps lie on that plane
            if tuple(weight.shape[2:]) != (1, 3, 3) or (ksize, stride, pad, dilation) != (3, 2, 1, 1):
                raise RuntimeError("a depth-1 transposed layer is ConvTranspose2d(k3,s2,p1,op1): weight [Cin,Cout,3,3]")
            w3 = torch.zeros(weight.shape[:2] + (3, 3, 3), dtype=torch.float32, device=weight.device)
            w3[:, :, 1] = weight[:, :, 0]
            weight = w3
        if transposed:
            self.cin, self.cout = weight.shape[0], weight.shape[1]
        else:
            self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.ksize_h = int(ksize_h) if planar else 0
        if tuple(weight.shape[2:]) != ((1, self.ksize_h or ksize, ksize) if (planar and not transposed) else (ksize,) * 3):
            raise RuntimeError("only cubic kernels (or depth-1 k x k / ksize_h x k ones with planar=True) are on the path")
        self.ksize, self.stride, self.pad, self.dilation = int(ksize), int(stride), int(pad), int(dilation)
        self.device = weight.device
        probe = self._desc(1, (1, 16, 32) if planar else (16, 16, 32), 0)
        count = _lib.lib().snvc_conv3d_packed_weight_count(ctypes.byref(probe))
        if count < 0:
            check(1, "snvc_conv3d_packed_weight_count")
        self.packed = torch.empty(count, dtype=torch.float32, device=weight.device)
        with torch.cuda.device(weight.device):
            check(_lib.lib().snvc_conv3d_pack_weights(ctypes.byref(probe), _ptr(weight.detach().contiguous()),
                                                      _ptr(self.packed), _stream(weight)), "snvc_conv3d_pack_weights")

    def out_spatial(self, in_spatial):
        if self.transposed and getattr(self, "planar", False):
            return (in_spatial[0], 2 * in_spatial[1], 2 * in_spatial[2])
        if self.transposed:
            return tuple(2 * s for s in in_spatial)
        eff = self.dilation * (self.ksize - 1) + 1
        if getattr(self, "planar", False):      # the stride and the padding apply to H and W only
            kh = getattr(self, "ksize_h", 0) or self.ksize
            return (in_spatial[0], (in_spatial[1] + 2 * ((kh - 1) // 2) - kh) // self.stride + 1,
                    (in_spatial[2] + 2 * self.pad - eff) // self.stride + 1)
        return tuple((s + 2 * self.pad - eff) // self.stride + 1 for s in in_spatial)

    def _desc(self, n, in_spatial, flags, x_bs=0, y_bs=0, r_bs=0) -> Conv3dDesc:
        d = Conv3dDesc()
        d.N, d.Cin = n, self.cin
        d.Din, d.Hin, d.Win = in_spatial
        d.Cout = self.cout
        d.Dout, d.Hout, d.Wout = self.out_spatial(in_spatial)
        d.ksize, d.stride, d.dilation, d.pad = self.ksize, self.stride, self.dilation, self.pad
        d.transposed = 1 if self.transposed else 0
        d.flags = flags
        d.ksize_d = 1 if getattr(self, "planar", False) else 0
        d.ksize_h = getattr(self, "ksize_h", 0)
        d.x_batch_stride, d.y_batch_stride, d.res_batch_stride = x_bs, y_bs, r_bs
        return d

    def __call__(self, x, scale=None, bias=None, residual=None, flags=0, out=None, depth_planes=None, exact=False,
                 side_head=None):
        """y = epilogue(conv(x)); x / out / residual may be channel slices of larger buffers.
        depth_planes [N,Cout,3,H,W]: see snvc_conv3d_forward_ex.  exact=True: SNVC_ALGO_DIRECT (no Winograd).
        side_head: a [Cout] (or [1,Cout,1,1,1]) weight -> returns ``(y, y_head)`` with ``y_head = sum_c side_head[c]*y[:, c]``
        written by the same launch (snvc_conv3d_forward_side_head), or ``(y, None)`` when the layer does not qualify."""
        _gpu(x, "x")
        if x.dtype != torch.float32 or x.dim() != 5 or x.size(1) != self.cin:
            raise RuntimeError(f"conv3d input must be float32 [N,{self.cin},D,H,W], got {tuple(x.shape)} {x.dtype}")
        if not _dense_inner(x):
            x = x.contiguous()
        n = x.size(0)
        in_sp = tuple(x.shape[2:])
        out_shape = (n, self.cout) + self.out_spatial(in_sp)
        if min(out_shape[2:]) < 1:
            raise RuntimeError("conv3d output would be empty")
        if out is None:
            out = torch.empty(out_shape, dtype=torch.float32, device=x.device)
        else:
            if tuple(out.shape) != out_shape or out.dtype != torch.float32 or not _dense_inner(out):
                raise RuntimeError("conv3d `out` must be a float32 channel-dense view of the output shape")
        if residual is not None:
            if tuple(residual.shape) != out_shape:
                raise RuntimeError("residual must have the output's shape")
            if not _dense_inner(residual):
                residual = residual.contiguous()
        if n == 0:
            return out
        d = self._desc(n, in_sp, flags, _batch_stride(x), _batch_stride(out),
                       _batch_stride(residual) if residual is not None else 0)
        d.algo = _algo(exact)
        if depth_planes is not None:
            if tuple(depth_planes.shape) != (n, self.cout, 3) + out_shape[3:] or not depth_planes.is_contiguous():
                raise RuntimeError("depth_planes must be a contiguous [N,Cout,3,H,W] tensor")
        with torch.cuda.device(x.device):
            if side_head is not None:
                hw = side_head.detach().reshape(-1).contiguous()
                if depth_planes is None and hw.numel() == self.cout and hw.dtype == torch.float32:
                    y_head = torch.empty((n, 1) + out_shape[2:], dtype=torch.float32, device=x.device)
                    rc = _lib.lib().snvc_conv3d_forward_side_head(ctypes.byref(d), _ptr(x), _ptr(self.packed), _ptr(scale),
                                                                 _ptr(bias), _ptr(residual), _ptr(out), _ptr(hw), _ptr(y_head),
                                                                 _stream(x))
                    if rc == 0:
                        return out, y_head
                    if rc != 2:        # anything but SNVC_ERR_UNSUPPORTED is an error
                        check(rc, "snvc_conv3d_forward_side_head")
            check(_lib.lib().snvc_conv3d_forward_ex(ctypes.byref(d), _ptr(x), _ptr(self.packed), _ptr(scale), _ptr(bias),
                                                    _ptr(residual), _ptr(depth_planes), _ptr(out), _stream(x)),
                  "snvc_conv3d_forward")
        return (out, None) if side_head is not None else out

    def forward_stats(self, x, gamma, beta, eps: float):
        """raw = conv(x) together with the batch statistics of raw, taken in the convolution's own epilogue
        (snvc_conv3d_forward_stats): returns (raw, scale, shift, mean, var) with the [1, C] shapes of ``norm_stats``, or None
        when the layer does not take a kernel form that carries the statistics epilogue (nothing was launched)."""
        _gpu(x, "x")
        if (self.planar or self.ksize != 3 or self.dilation != 1 or self.stride not in (1, 2) or (self.transposed and self.stride != 2) or self.cout % 32
                or x.dtype != torch.float32 or x.dim() != 5 or x.size(1) != self.cin or x.size(0) == 0 or _algo() != 0):
            return None
        if not _dense_inner(x):
            x = x.contiguous()
        n, in_sp = x.size(0), tuple(x.shape[2:])
        out_shape = (n, self.cout) + self.out_spatial(in_sp)
        if min(out_shape[2:]) < 1:
            return None
        out = torch.empty(out_shape, dtype=torch.float32, device=x.device)
        d = self._desc(n, in_sp, 0, _batch_stride(x), _batch_stride(out), 0)
        d.algo = _algo()
        nbytes = _lib.lib().snvc_conv3d_stats_workspace_bytes(ctypes.byref(d))
        if nbytes < 0:
            return None
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        scale = torch.empty((1, self.cout), dtype=torch.float32, device=x.device)
        shift, mean, var = torch.empty_like(scale), torch.empty_like(scale), torch.empty_like(scale)
        with torch.cuda.device(x.device):
            rc = _lib.lib().snvc_conv3d_forward_stats(ctypes.byref(d), _ptr(x), _ptr(self.packed), _ptr(out), _ptr(gamma), _ptr(beta),
                                                      _ptr(scale), _ptr(shift), _ptr(mean), _ptr(var), _ptr(ws), float(eps), _stream(x))
        if rc == 2:            # SNVC_ERR_UNSUPPORTED: the caller runs the convolution and the statistics pass separately
            return None
        check(rc, "snvc_conv3d_forward_stats")
        return out, scale, shift, mean, var


WARPED_EXPAND_R3 = 256           # include/snvc_hip.h SNVC_WARPED_EXPAND_R3
WARPED_EXPAND_FORM = [0]         # tests / A-B timing: [WARPED_EXPAND_R3] selects the r3 kernel form


def warped_expand(p, q, e, planes, shift, scale, bias, out, flags: int = 0):
    """First convolution over the warped half for ANY shift array, warp after convolution (snvc_warped_expand):
    p, q [N,3C,H,W] (kd-major stacks of the depth-1 convolutions of the right feature: all taps | the kw = +1 taps alone),
    e [N,9C,H,4] ((kd, kw)-major, the first image column), planes [N,C,3,H,W] or None, shift [N,D] float32,
    out [N,C,D,H,W] written in place."""
    _gpu(p, "p"); _gpu(q, "q"); _gpu(e, "e"); _gpu(out, "out"); _gpu(shift, "shift")
    n, c, d, h, w = out.shape
    for t, ch, ww in ((p, 3 * c, w), (q, 3 * c, w), (e, 9 * c, 4)):
        if t.dtype != torch.float32 or tuple(t.shape) != (n, ch, h, ww) or not t.is_contiguous():
            raise RuntimeError("warped_expand needs contiguous float32 p / q [N,3C,H,W] and e [N,9C,H,4]")
    if out.dtype != torch.float32 or not out.is_contiguous():
        raise RuntimeError("warped_expand needs a contiguous float32 out [N,C,D,H,W]")
    if shift.dtype != torch.float32 or tuple(shift.shape) != (n, d):
        raise RuntimeError("warped_expand needs a float32 shift [N,D]")
    if planes is not None and (tuple(planes.shape) != (n, c, 3, h, w) or not planes.is_contiguous()):
        raise RuntimeError("planes must be a contiguous [N,C,3,H,W] tensor")
    shift = shift.contiguous()
    with torch.cuda.device(out.device):
        check(_lib.lib().snvc_warped_expand(_ptr(p), _ptr(q), _ptr(e), _ptr(planes), _ptr(shift), _ptr(scale), _ptr(bias), _ptr(out),
                                            n, c, d, h, w, int(flags) | WARPED_EXPAND_FORM[0], _stream(out)), "snvc_warped_expand")
    return out


def warped_expand_backward(dy, shift, want_planes: bool = True):
    """Adjoint of ``warped_expand`` in one pass over the output gradient dy [N,C,D,H,W] (snvc_warped_expand_backward):
    returns (a [N,3,3,C,H,W], dplanes [N,C,3,H,W] or None) -- a[:, kd, kw] = dy taken back through the warp of plane d+kd-1 and
    the (kd, kw) tap, from which the right feature's and the weights' gradients are depth-1 convolution work; dplanes = the
    depth-class sums of dy (``depth_class_sums``) from the same pass.  Raises ``Unsupported`` for W > 1024."""
    _gpu(dy, "dy"); _gpu(shift, "shift")
    if dy.dtype != torch.float32 or dy.dim() != 5:
        raise RuntimeError("warped_expand_backward needs a float32 dy [N,C,D,H,W]")
    dy = dy.contiguous()
    n, c, d, h, w = dy.shape
    if shift.dtype != torch.float32 or tuple(shift.shape) != (n, d):
        raise RuntimeError("warped_expand_backward needs a float32 shift [N,D]")
    shift = shift.contiguous()
    a = torch.empty((n, 3, 3, c, h, w), dtype=torch.float32, device=dy.device)
    dpl = torch.empty((n, c, 3, h, w), dtype=torch.float32, device=dy.device) if want_planes else None
    with torch.cuda.device(dy.device):
        check(_lib.lib().snvc_warped_expand_backward(_ptr(dy), _ptr(shift), _ptr(a), _ptr(dpl), n, c, d, h, w, _stream(dy)),
              "snvc_warped_expand_backward")
    return a, dpl


def warped_expand_split(p, q, e, planes, shift, scale, bias, out, flags: int = 0, overflow=None):
    """``warped_expand`` with the result written as a split C8 pair (snvc_warped_expand_split): ``out`` = float16
    [N, 2, C/8, D, H, W, 8] holding the layer's result times the power of two the caller folded into scale / bias."""
    _gpu(p, "p"); _gpu(q, "q"); _gpu(e, "e")
    _split_check(out, "out")
    n, _, grp, d, h, w, _ = out.shape
    c = p.size(1) // 3
    for t, shp in ((p, (n, 3 * c, h, w)), (q, (n, 3 * c, h, w)), (e, (n, 9 * c, h, 4))):
        if t.dtype != torch.float32 or tuple(t.shape) != shp or not t.is_contiguous():
            raise RuntimeError("warped_expand_split needs contiguous float32 p / q [N,3C,H,W] and e [N,9C,H,4]")
    if (c + 7) // 8 != grp:
        raise RuntimeError("warped_expand_split: out has the wrong number of channel groups")
    if shift.dtype != torch.float32 or tuple(shift.shape) != (n, d):
        raise RuntimeError("warped_expand_split needs a float32 shift [N,D]")
    if planes is not None and (tuple(planes.shape) != (n, c, 3, h, w) or not planes.is_contiguous()):
        raise RuntimeError("planes must be a contiguous [N,C,3,H,W] tensor")
    shift = shift.contiguous()
    with torch.cuda.device(out.device):
        check(_lib.lib().snvc_warped_expand_split(_ptr(p), _ptr(q), _ptr(e), _ptr(planes), _ptr(shift), _ptr(scale), _ptr(bias), _ptr(out),
                                                  _lo_ptr(out), _ptr(overflow), n, c, d, h, w, _batch_stride(out), int(flags), _stream(out)),
              "snvc_warped_expand_split")
    return out


def shift_structure(shift):
    """(all shifts >= 0, rows == s0 + d, rows == s0 + d/2, s0) of a float32 [N, D] shift array: one launch and one 16-byte
    device -> host copy (snvc_shift_structure)."""
    _gpu(shift, "shift")
    if shift.dtype != torch.float32 or shift.dim() != 2 or shift.numel() == 0:
        raise RuntimeError("shift_structure needs a non-empty float32 [N, D] tensor")
    return shift_structure_result(shift_structure_begin(shift))


_PINNED4 = {}
_PINNED4_LOCK = threading.Lock()


def shift_structure_begin(shift):
    """Launches snvc_shift_structure and the 16-byte copy of its result to pinned host memory; returns a ticket for
    ``shift_structure_result``.  Work queued between the two calls overlaps the host's wait (``GlobalStack.forward_pair``
    queues the launches that do not depend on the answer first)."""
    _gpu(shift, "shift")
    if shift.dtype != torch.float32 or shift.dim() != 2 or shift.numel() == 0:
        raise RuntimeError("shift_structure needs a non-empty float32 [N, D] tensor")
    shift = shift.contiguous()
    out = torch.empty(4, dtype=torch.float32, device=shift.device)
    # a ticket of its own per outstanding call (two unresolved tickets, or two threads on one stream, must not share the
    # 16-byte result and its event); resolved tickets go back to a per-device free list
    with _PINNED4_LOCK:
        free = _PINNED4.setdefault(shift.device, [])
        host = free.pop() if free else None
    if host is None:
        host = (torch.empty(4, dtype=torch.float32).pin_memory(), torch.cuda.Event(), shift.device)
    with torch.cuda.device(shift.device):
        check(_lib.lib().snvc_shift_structure(_ptr(shift), _ptr(out), shift.size(0), shift.size(1), _stream(shift)),
              "snvc_shift_structure")
        host[0].copy_(out, non_blocking=True)
        host[1].record()
    return host


def spin_wait(event, spin_us: float = 4000.0):
    """Wait for a recorded event by polling it (hipEventQuery, ~1 us a look) before falling back to the blocking wait: the waits of a
    step are for results that are a few microseconds away, and a sleeping wait costs tens of microseconds to wake (r5: 0.076 ms of
    a 2.04 ms step between the overflow flag's arrival and the next launch).  The poll is bounded by TIME (r6, ADVICE r5: an
    iteration count let a deep queue or a shared device burn a core, and the GIL, for tens of milliseconds) and hands the GIL over
    every ~50 us (time.sleep(0): worker threads of a DataParallel caller keep running).  The bound is 4 ms: the flag's wait of a
    2 ms step is ~1.5 ms long once the host runs ahead of the GPU, and on some hosts the blocking wait wakes up 0.8 ms late -- one
    box measured the headline at 2.77 instead of 1.98 ms/step with a 1 ms bound (profiles/r6/kernel_experiments_r6.txt, item 26);
    anything further away than 4 ms sleeps."""
    if event.query():
        return
    now = time.perf_counter()
    deadline, nxt = now + spin_us * 1e-6, now + 50e-6
    while now < deadline:
        if event.query():
            return
        now = time.perf_counter()
        if now >= nxt:
            time.sleep(0)
            nxt = now + 50e-6
    event.synchronize()


def shift_structure_result(ticket):
    spin_wait(ticket[1])
    nonneg, u1, u2, first = ticket[0].tolist()
    with _PINNED4_LOCK:
        free = _PINNED4.setdefault(ticket[2], [])
        if len(free) < 8 and not any(t is ticket for t in free):
            free.append(ticket)
    return bool(nonneg), bool(u1), bool(u2), first


def shift_spacing_result(ticket, d: int):
    """(all shifts >= 0, (q, m0) or None): (q, m0) when every row of the shift array is (m0 + d) / q for d = 0..D-1 with q in
    {1, 2} -- uniformly spaced whole- or half-pixel disparity planes -- exactly, in fp32 (the sheared first convolution)."""
    nonneg, u1, u2, first = shift_structure_result(ticket)
    q = 1 if u1 else (2 if u2 else 0)
    if q == 0 or d < 4 or not math.isfinite(first):     # an all-+inf array is "non-negative" and "uniform": not a spacing
        return nonneg, None
    m0 = first * q
    if m0 != int(m0) or not (0 <= m0 < 1 << 20):
        return nonneg, None
    return nonneg, (q, int(m0))


def sheared_upsample(right, q: int, wu: int, off: int):
    """Rq on a padded grid (snvc_sheared_upsample): right [N,C,H,W] -> [N,C,H,wu], element i = Rq[i - off]."""
    _gpu(right, "right")
    if right.dtype != torch.float32 or right.dim() != 4:
        raise RuntimeError("sheared_upsample needs a float32 [N,C,H,W] tensor")
    right = right.contiguous()
    n, c, h, w = right.shape
    out = torch.empty((n, c, h, wu), dtype=torch.float32, device=right.device)
    with torch.cuda.device(right.device):
        check(_lib.lib().snvc_sheared_upsample(_ptr(right), _ptr(out), n, c, h, w, int(q), int(wu), int(off), _stream(right)),
              "snvc_sheared_upsample")
    return out


def sheared_expand(g, gcol, planes, scale, bias, out, q: int, m0: int, off: int, off_col: int, flags: int = 0,
                   amax: Optional[torch.Tensor] = None):
    """out[n,co,d,h,w] = epilogue(scale*G[n,cls(d),co,h,q*w-d-m0+off] + planes[n,co,cls(d),h,w] + bias), with G' (``gcol``,
    indexed with ``off_col``) in place of G at w = W-1 (snvc_sheared_expand); g [N,3C,H,WG], gcol [N,3C,H,WG2] (depth classes
    first / interior / last stacked class-major), planes [N,C,3,H,W] or None, out [N,C,D,H,W] (contiguous, written in place).
    ``amax`` (r6): zeroed words (amax_word) that receive the bit pattern of max|out| (snvc_sheared_expand_amax)."""
    _gpu(g, "g"); _gpu(gcol, "gcol"); _gpu(out, "out")
    n, c, d, h, w = out.shape
    for t in (g, gcol):
        if t.dtype != torch.float32 or tuple(t.shape[:3]) != (n, 3 * c, h) or not t.is_contiguous():
            raise RuntimeError("sheared_expand needs contiguous float32 g / gcol [N,3C,H,*] (depth classes stacked class-major)")
    if out.dtype != torch.float32 or not out.is_contiguous():
        raise RuntimeError("sheared_expand needs a contiguous float32 out [N,C,D,H,W]")
    if planes is not None and (tuple(planes.shape) != (n, c, 3, h, w) or not planes.is_contiguous()):
        raise RuntimeError("planes must be a contiguous [N,C,3,H,W] tensor")
    with torch.cuda.device(out.device):
        check(_lib.lib().snvc_sheared_expand_amax(_ptr(g), _ptr(gcol), _ptr(planes), _ptr(scale), _ptr(bias), _ptr(out), n, c, d, h, w,
                                                  int(q), int(m0), g.size(3), int(off), gcol.size(3), int(off_col), int(flags),
                                                  _ptr(amax), _stream(out)), "snvc_sheared_expand")
    return out


def sheared_expand_split(g, gcol, planes, scale, bias, out, q: int, m0: int, off: int, off_col: int, flags: int = 0, overflow=None):
    """``sheared_expand`` with the result written as a split C8 pair (snvc_sheared_expand_split): ``out`` = float16
    [N, 2, C/8, D, H, W, 8] holding the layer's result times the power of two the caller folded into scale / bias."""
    _gpu(g, "g"); _gpu(gcol, "gcol")
    _split_check(out, "out")
    n, _, grp, d, h, w, _ = out.shape
    c = g.size(1) // 3
    if (c + 7) // 8 != grp:
        raise RuntimeError("sheared_expand_split: out has the wrong number of channel groups")
    for t in (g, gcol):
        if t.dtype != torch.float32 or tuple(t.shape[:3]) != (n, 3 * c, h) or not t.is_contiguous():
            raise RuntimeError("sheared_expand_split needs contiguous float32 g / gcol [N,3C,H,*] (depth classes stacked class-major)")
    if planes is not None and (tuple(planes.shape) != (n, c, 3, h, w) or not planes.is_contiguous()):
        raise RuntimeError("planes must be a contiguous [N,C,3,H,W] tensor")
    with torch.cuda.device(out.device):
        check(_lib.lib().snvc_sheared_expand_split(_ptr(g), _ptr(gcol), _ptr(planes), _ptr(scale), _ptr(bias), _ptr(out), _lo_ptr(out),
                                                   _ptr(overflow), n, c, d, h, w, int(q), int(m0), g.size(3), int(off), gcol.size(3),
                                                   int(off_col), _batch_stride(out), int(flags), _stream(out)), "snvc_sheared_expand_split")
    return out


def sheared_expand_stats(g, gcol, planes, gamma, beta, shape, q: int, m0: int, off: int, off_col: int, eps: float):
    """Batch statistics of the sheared layer's raw result without storing it (snvc_sheared_expand_stats): returns
    (scale, shift, mean, var), each [1, C], for ``y = relu(scale * raw + shift)``; ``shape`` = (N, C, D, H, W) of the result."""
    _gpu(g, "g"); _gpu(gcol, "gcol"); _gpu(planes, "planes")
    n, c, d, h, w = shape
    for t in (g, gcol):
        if t.dtype != torch.float32 or tuple(t.shape[:3]) != (n, 3 * c, h) or not t.is_contiguous():
            raise RuntimeError("sheared_expand_stats needs contiguous float32 g / gcol [N,3C,H,*]")
    if tuple(planes.shape) != (n, c, 3, h, w) or not planes.is_contiguous():
        raise RuntimeError("planes must be a contiguous [N,C,3,H,W] tensor")
    scale = torch.empty((1, c), dtype=torch.float32, device=g.device)
    shift, mean, var = torch.empty_like(scale), torch.empty_like(scale), torch.empty_like(scale)
    ws = torch.empty(_lib.lib().snvc_sheared_stats_workspace_bytes(n, c, h), dtype=torch.uint8, device=g.device)
    with torch.cuda.device(g.device):
        check(_lib.lib().snvc_sheared_expand_stats(_ptr(g), _ptr(gcol), _ptr(planes), _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift),
                                                   _ptr(mean), _ptr(var), _ptr(ws), n, c, d, h, w, int(q), int(m0), g.size(3), int(off),
                                                   gcol.size(3), int(off_col), float(eps), _stream(g)), "snvc_sheared_expand_stats")
    return scale, shift, mean, var


def sheared_backward_reduce(g, gcol, planes, scale, shift, gy, q: int, m0: int, off: int, off_col: int):
    """One pass over ``gy`` = dL/d relu(scale*raw + shift) of the sheared layer (snvc_sheared_backward_reduce): returns
    (line [2,N,3C,H,WG], colsum [2,N,C,3,H,W], lastc [2,N,3C,H,WG2], sums [N,C,2] fp64) -- index 0: sums of the masked gradient,
    1: of the recomputed raw result -- from which the caller forms dG, dG', the depth-class planes and the BatchNorm terms."""
    _gpu(gy, "gy")
    if gy.dtype != torch.float32 or gy.dim() != 5 or not gy.is_contiguous():
        raise RuntimeError("sheared_backward_reduce needs a contiguous float32 [N,C,D,H,W] gradient")
    n, c, d, h, w = gy.shape
    wg, wg2 = g.size(3), gcol.size(3)
    dev = gy.device
    line = torch.empty((2, n, 3 * c, h, wg), dtype=torch.float32, device=dev)
    colsum = torch.empty((2, n, c, 3, h, w), dtype=torch.float32, device=dev)
    lastc = torch.empty((2, n, 3 * c, h, wg2), dtype=torch.float32, device=dev)
    sums = torch.empty((n, c, 2), dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.lib().snvc_sheared_backward_workspace_bytes(n, c, h), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().snvc_sheared_backward_reduce(_ptr(g), _ptr(gcol), _ptr(planes), _ptr(scale), _ptr(shift), _ptr(gy), _ptr(line),
                                                      _ptr(colsum), _ptr(lastc), _ptr(sums), _ptr(ws), n, c, d, h, w, int(q), int(m0), wg,
                                                      int(off), wg2, int(off_col), _stream(gy)), "snvc_sheared_backward_reduce")
    return line, colsum, lastc, sums


def sheared_reduce(dy, q: int, m0: int, wg: int, off: int, wg_col: int, off_col: int):
    """Adjoint of ``sheared_expand`` (snvc_sheared_reduce): dy [N,C,D,H,W] -> (dg [N,3C,H,wg], dgcol [N,3C,H,wg_col]), the sums
    of dy along each shear line per depth class (class-major); every element is written, the sums are deterministic."""
    _gpu(dy, "dy")
    if dy.dtype != torch.float32 or dy.dim() != 5 or not dy.is_contiguous():
        raise RuntimeError("sheared_reduce needs a contiguous float32 [N,C,D,H,W] gradient")
    n, c, d, h, w = dy.shape
    dg = torch.empty((n, 3 * c, h, wg), dtype=torch.float32, device=dy.device)
    dgcol = torch.empty((n, 3 * c, h, wg_col), dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        check(_lib.lib().snvc_sheared_reduce(_ptr(dy), _ptr(dg), _ptr(dgcol), n, c, d, h, w, int(q), int(m0), int(wg), int(off),
                                             int(wg_col), int(off_col), _stream(dy)), "snvc_sheared_reduce")
    return dg, dgcol


def sheared_wgrad(x, dy):
    """Weight gradient of the depth-1 3x7 layer (snvc_sheared_wgrad): x [N,C,H,WU], dy [N,CO,H,WU] -> dK [CO,C,3,7]."""
    _gpu(x, "x"); _gpu(dy, "dy")
    if (x.dtype != torch.float32 or dy.dtype != torch.float32 or x.dim() != 4 or dy.dim() != 4 or not x.is_contiguous()
            or not dy.is_contiguous() or x.size(0) != dy.size(0) or tuple(x.shape[2:]) != tuple(dy.shape[2:])):
        raise RuntimeError("sheared_wgrad needs contiguous float32 x [N,C,H,WU] and dy [N,CO,H,WU]")
    n, c, h, wu = x.shape
    co = dy.size(1)
    dk = torch.empty((co, c, 3, 7), dtype=torch.float32, device=x.device)
    nbytes = _lib.lib().snvc_sheared_wgrad_workspace_bytes(n, co, h, wu)
    if nbytes < 0:
        raise RuntimeError("snvc_sheared_wgrad_workspace_bytes: bad sizes")
    ws = torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_sheared_wgrad(_ptr(x), _ptr(dy), _ptr(dk), _ptr(ws), n, c, co, h, wu, _stream(x)), "snvc_sheared_wgrad")
    return dk


def sheared_upsample_backward(drq, q: int, w: int, off: int):
    """Adjoint of ``sheared_upsample`` (snvc_sheared_upsample_backward): drq [N,C,H,WU] (element i = dRq[i - off]) -> [N,C,H,w]."""
    _gpu(drq, "drq")
    if drq.dtype != torch.float32 or drq.dim() != 4 or not drq.is_contiguous():
        raise RuntimeError("sheared_upsample_backward needs a contiguous float32 [N,C,H,WU] tensor")
    n, c, h, wu = drq.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=drq.device)
    with torch.cuda.device(drq.device):
        check(_lib.lib().snvc_sheared_upsample_backward(_ptr(drq), _ptr(out), n, c, h, int(w), int(q), wu, int(off), _stream(drq)),
              "snvc_sheared_upsample_backward")
    return out


def conv3d_forward_avgpool_d4(layer: "Conv3dLayer", x, scale, bias, flags) -> Optional[torch.Tensor]:
    """AvgPool3d((4,1,1),(4,1,1)) of ``epilogue(conv(x))`` written by the layer's own launch (SNVC_EPI_AVGPOOL_D4): returns
    [N,Cout,D/4,H,W], or None when the layer does not qualify (the caller then pools with ``avgpool_depth4``)."""
    _gpu(x, "x")
    if layer.transposed or layer.planar or layer.ksize != 3 or layer.stride != 1 or layer.dilation != 1:
        return None
    if x.dtype != torch.float32 or x.dim() != 5 or x.size(1) != layer.cin or x.size(2) % 4 != 0 or layer.cout % 32 != 0:
        return None
    if not _dense_inner(x):
        x = x.contiguous()
    n = x.size(0)
    in_sp = tuple(x.shape[2:])
    out = torch.empty((n, layer.cout, in_sp[0] // 4, in_sp[1], in_sp[2]), dtype=torch.float32, device=x.device)
    if n == 0:
        return out
    d = layer._desc(n, in_sp, flags | _lib.EPI_AVGPOOL_D4, _batch_stride(x), 0, 0)
    d.algo = _algo()
    with torch.cuda.device(x.device):
        rc = _lib.lib().snvc_conv3d_forward_ex(ctypes.byref(d), _ptr(x), _ptr(layer.packed), _ptr(scale), _ptr(bias), _ptr(None),
                                               _ptr(None), _ptr(out), _stream(x))
    if rc == 2:        # SNVC_ERR_UNSUPPORTED
        return None
    check(rc, "snvc_conv3d_forward(pooled)")
    return out


def conv3d_forward_head(layer: "Conv3dLayer", x, scale, bias, residual, flags, head_weight) -> Optional[torch.Tensor]:
    """epilogue(deconv(x)) projected to one channel by ``head_weight`` [Cout] inside the layer's epilogue
    (snvc_conv3d_forward_head); returns [N,1,D,H,W], or None when the layer does not qualify."""
    if not layer.transposed or layer.cout != 32 or x.size(4) % 2 != 0:
        return None
    _gpu(x, "x")
    if not _dense_inner(x):
        x = x.contiguous()
    n = x.size(0)
    in_sp = tuple(x.shape[2:])
    out_sp = layer.out_spatial(in_sp)
    if residual is not None and (tuple(residual.shape) != (n, layer.cout) + out_sp or not _dense_inner(residual)
                                 or residual.data_ptr() % 16 or _batch_stride(residual) % 4):
        return None
    out = torch.empty((n, 1) + out_sp, dtype=torch.float32, device=x.device)
    if n == 0:
        return out
    d = layer._desc(n, in_sp, flags, _batch_stride(x), 0, _batch_stride(residual) if residual is not None else 0)
    d.algo = _algo()
    hw = head_weight.detach().reshape(-1).contiguous()
    with torch.cuda.device(x.device):
        rc = _lib.lib().snvc_conv3d_forward_head(ctypes.byref(d), _ptr(x), _ptr(layer.packed), _ptr(scale), _ptr(bias),
                                                 _ptr(residual), _ptr(hw), _ptr(out), _stream(x))
    if rc == 2:        # SNVC_ERR_UNSUPPORTED: run the two layers separately
        return None
    check(rc, "snvc_conv3d_forward_head")
    return out


_WORKSPACES = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Scratch for the kernels that reduce per-workgroup partials (weight gradients): one buffer per (device, stream), grown
    when a layer needs more -- the launches that use it are ordered on that stream, so layers can share it and a training
    step allocates nothing here after its first pass."""
    if nbytes < 0:
        raise RuntimeError("workspace size query failed")
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WORKSPACES[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
    return ws


def release_workspaces() -> None:
    """Drop the cached scratch buffers (e.g. before handing the device's memory to something else)."""
    _WORKSPACES.clear()


def conv3d_wgrad(x_big, g_small, ksize: int, stride: int, pad: int, dilation: int, amax_x: Optional[torch.Tensor] = None,
                 amax_g: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[cg][cx][k^3] = sum over batch and voxels of g_small[cg] * x_big[cx] (shifted by the tap):
    weight gradient of Conv3d(x_big -> g_small's shape); see snvc_conv3d_wgrad for the
    ConvTranspose3d usage (roles swapped).  Deterministic.  ``amax_x`` / ``amax_g``: int32 words holding the bit pattern of
    max|x_big| / max|g_small| (amax_word, filled by the pass that wrote the tensor): the split-operand form (3x3x3, stride 1) then
    skips its own reduction over that tensor."""
    _gpu(x_big, "x"); _gpu(g_small, "g")
    if not _dense_inner(x_big):
        x_big = x_big.contiguous()
    if not _dense_inner(g_small):
        g_small = g_small.contiguous()
    d = Conv3dDesc()
    d.N, d.Cin = x_big.shape[0], x_big.shape[1]
    d.Din, d.Hin, d.Win = x_big.shape[2:]
    d.Cout = g_small.shape[1]
    d.Dout, d.Hout, d.Wout = g_small.shape[2:]
    d.ksize, d.stride, d.dilation, d.pad = ksize, stride, dilation, pad
    d.transposed, d.flags, d.algo = 0, 0, _algo()
    d.x_batch_stride, d.y_batch_stride, d.res_batch_stride = _batch_stride(x_big), _batch_stride(g_small), 0
    ws = _workspace(_lib.lib().snvc_conv3d_wgrad_workspace_bytes(ctypes.byref(d)), x_big.device)
    dw = torch.empty((d.Cout, d.Cin, ksize, ksize, ksize), dtype=torch.float32, device=x_big.device)
    with torch.cuda.device(x_big.device):
        check(_lib.lib().snvc_conv3d_wgrad_amax(ctypes.byref(d), _ptr(x_big), _ptr(g_small), _ptr(dw), _ptr(ws), _ptr(amax_x),
                                                _ptr(amax_g), _stream(x_big)), "snvc_conv3d_wgrad")
    return dw


def bn_backward_coefs(sums, mean, var, gamma, count: float, eps: float):
    """Train-mode BatchNorm backward coefficients in one launch (snvc_bn_backward_coefs): sums [N, C, 2] fp64 from
    act_backward_reduce, mean / var [C] the forward's batch statistics, gamma [C] or None.
    Returns (coef_g, coef_raw, coef_const, dgamma, dbeta), float32 [C]."""
    n, c = sums.shape[0], sums.shape[1]
    out = torch.empty((5, c), dtype=torch.float32, device=sums.device)
    with torch.cuda.device(sums.device):
        check(_lib.lib().snvc_bn_backward_coefs(_ptr(sums), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(out[0]), _ptr(out[1]),
                                               _ptr(out[2]), _ptr(out[3]), _ptr(out[4]), n, c, float(count), float(eps),
                                               _stream(sums)), "snvc_bn_backward_coefs")
    return out[0], out[1], out[2], out[3], out[4]


def bn_track(norm, mean, var, cnt: float) -> bool:
    """nn.BatchNorm's train-mode bookkeeping in one launch (snvc_bn_track); False when this call is not the plain float32 CUDA case
    (the caller then does it with torch ops)."""
    rm, rv, nbt = norm.running_mean, norm.running_var, norm.num_batches_tracked
    if (norm.momentum is None or not rm.is_cuda or rm.dtype != torch.float32 or rv.dtype != torch.float32 or not rm.is_contiguous()
            or not rv.is_contiguous() or mean.dtype != torch.float32 or var.dtype != torch.float32 or not mean.is_contiguous()
            or not var.is_contiguous() or mean.numel() != rm.numel() or var.numel() != rm.numel() or mean.device != rm.device
            or (nbt is not None and (nbt.dtype != torch.int64 or nbt.device != rm.device))):
        return False
    with torch.cuda.device(rm.device):
        check(_lib.lib().snvc_bn_track(_ptr(rm), _ptr(rv), _ptr(nbt), _ptr(mean), _ptr(var), rm.numel(), float(norm.momentum),
                                       float(cnt / max(cnt - 1, 1)), _stream(rm)), "snvc_bn_track")
    # the kernel wrote through raw pointers: tell the version counters (the folded eval-mode BatchNorm is cached on ._version)
    for t in (rm, rv) + ((nbt,) if nbt is not None else ()):
        torch.autograd.graph.increment_version(t)
    return True


def act_backward_reduce(raw, gy, residual, scale, shift, flags: int, per_sample: bool, amax_gy: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per (n, c): [sum(g), sum(g*raw)] in fp64, g = gy * act'(raw*scale + shift [+ residual]).  ``amax_gy``: zeroed words (amax_word)
    that receive the bit pattern of max|gy| -- the pass reads all of gy anyway."""
    n, c = raw.shape[0], raw.shape[1]
    s = raw[0, 0].numel()
    sums = torch.empty((n, c, 2), dtype=torch.float64, device=raw.device)
    ws = torch.empty(_lib.lib().snvc_act_backward_workspace_bytes(n, c), dtype=torch.uint8, device=raw.device)
    with torch.cuda.device(raw.device):
        check(_lib.lib().snvc_act_backward_reduce_amax(_ptr(raw), _ptr(gy), _ptr(residual), _ptr(scale), _ptr(shift),
                                                       _ptr(sums), _ptr(ws), n, c, s, _batch_stride(raw), _batch_stride(gy),
                                                       _batch_stride(residual) if residual is not None else 0,
                                                       1 if per_sample else 0, flags, _ptr(amax_gy), _stream(raw)),
              "snvc_act_backward_reduce")
    return sums


AMAX_SLOTS = 64      # SNVC_AMAX_SLOTS


_AMAX_POOL = threading.local()
_AMAX_POOL_WORDS = 64


def amax_word(device) -> torch.Tensor:
    """Zeroed device words for the bit pattern of a tensor's max|.| (snvc_*_amax entry points: SNVC_AMAX_SLOTS slots, the maximum over
    them is the value; a single word would serialise the producers' atomics).  Handed out from a pool that is zeroed in ONE launch
    per 64 words (a training step takes ~15): a word is a view of its pool and is never zeroed or handed out again, so a tag that
    still refers to it stays valid; an exhausted pool is simply dropped (its views keep it alive as long as they live)."""
    pools = _AMAX_POOL.__dict__.setdefault("pools", {})
    key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ent = pools.get(key)
    if ent is None or ent[1] >= _AMAX_POOL_WORDS:
        ent = pools[key] = [torch.zeros((_AMAX_POOL_WORDS, AMAX_SLOTS), dtype=torch.int32, device=device), 0]
    w = ent[0][ent[1]]
    ent[1] += 1
    return w


def amax_from_bound(bound: torch.Tensor) -> torch.Tensor:
    """The words for an UPPER BOUND of max|t| given as a one-element float32 device tensor (no sync): a bound 2^k above the true
    maximum costs the split-operand weight gradient k of the 39 bits it keeps below the maximum."""
    w = amax_word(bound.device)
    w[0:1] = bound.detach().float().reshape(1).abs().view(torch.int32)
    return w


def tag_amax(t: torch.Tensor, amax: Optional[torch.Tensor]) -> torch.Tensor:
    """Remember the word that holds max|t| on the tensor object itself: the layer that consumes ``t`` hands it to its weight gradient
    (split-operand form) instead of reading the whole tensor again.  Purely an optimisation: a tensor without the tag works."""
    if amax is not None:
        t.snvc_amax_tag = (amax, t._version)
    return t


def amax_of(t: torch.Tensor) -> Optional[torch.Tensor]:
    """The tagged word, if the tensor has not been written to since it was tagged (an in-place update bumps ``_version``)."""
    tag = getattr(t, "snvc_amax_tag", None)
    if tag is None or tag[1] != t._version or tag[0].device != t.device:
        return None
    return tag[0]


def twin_ok(t: torch.Tensor) -> bool:
    """Can the pass that writes ``t`` (float32 [N,C,D,H,W]) write its split C8 twin as well (snvc_*_twin)?"""
    return (t.is_cuda and t.dtype == torch.float32 and t.dim() == 5 and t.size(1) % 8 == 0 and t[0, 0].numel() % 4 == 0
            and t.numel() > 0 and t.size(1) // 8 <= 65535)


def twin_empty(like: torch.Tensor) -> torch.Tensor:
    n, c = like.shape[0], like.shape[1]
    return torch.empty((n, 2, c // 8) + tuple(like.shape[2:]) + (8,), dtype=torch.float16, device=like.device)


def tag_twin(t: torch.Tensor, pair: torch.Tensor, mul_dev: torch.Tensor) -> torch.Tensor:
    """Remember ``t``'s split twin (``pair`` holds t * mul_dev) on the tensor object; void once t is written to (``_version``)."""
    t.snvc_twin_tag = (pair, mul_dev, t._version)
    return t


def twin_of(t: torch.Tensor):
    """(pair, mul_dev) tagged by the pass that wrote ``t``, or None."""
    tag = getattr(t, "snvc_twin_tag", None)
    if tag is None or tag[2] != t._version or tag[0].device != t.device:
        return None
    return tag[0], tag[1]


def split_scale_bound(rows: int, c: int, device, a=None, amax_p=None, b=None, l1=None, amax_x=None, cc=None, amax_r=None) -> torch.Tensor:
    """The scale of a twin from an upper bound of its tensor's maximum, on the device (snvc_split_scale_bound):
    bound = max_r(|a[r]| * P + |b[r]| * l1[r % C] * X + |cc[r]|) + R with P / X / R the values of the amax words given."""
    out = torch.empty(1, dtype=torch.float32, device=device)
    for v in (a, b, cc):
        if v is not None and (v.dtype != torch.float32 or v.numel() != rows or not v.is_contiguous()):
            raise RuntimeError("split_scale_bound: a / b / c must be contiguous float32 vectors of `rows` elements")
    if l1 is not None and (l1.dtype != torch.float32 or l1.numel() != c or not l1.is_contiguous()):
        raise RuntimeError("split_scale_bound: l1 must be a contiguous float32 vector of C elements")
    for w in (amax_p, amax_x, amax_r):
        if w is not None and (w.dtype != torch.int32 or w.numel() != AMAX_SLOTS or not w.is_contiguous()):
            raise RuntimeError("split_scale_bound: amax words come from amax_word()")
    with torch.cuda.device(device):
        check(_lib.lib().snvc_split_scale_bound(_ptr(a), _ptr(amax_p), _ptr(b), _ptr(l1), _ptr(amax_x), _ptr(cc), _ptr(amax_r), rows, c,
                                                _ptr(out), ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
              "snvc_split_scale_bound")
    return out


def act_backward_apply(raw, gy, residual, scale, shift, coef_g, coef_raw, coef_const, flags: int, per_sample: bool,
                       want_g: bool, amax: Optional[torch.Tensor] = None, twin_mul: Optional[torch.Tensor] = None):
    """draw = coef_g*g + coef_raw*raw + coef_const (per channel or per (n,c)); optionally also g.  ``amax``: a zeroed int32 word
    that receives the bit pattern of max|draw| (amax_word).  ``twin_mul`` (r6): also write draw's split twin scaled by that device
    float (split_scale_bound) and tag it onto draw (snvc_act_backward_apply_twin)."""
    n, c = raw.shape[0], raw.shape[1]
    s = raw[0, 0].numel()
    draw = torch.empty(raw.shape, dtype=torch.float32, device=raw.device)
    g_out = torch.empty(raw.shape, dtype=torch.float32, device=raw.device) if want_g else None
    if twin_mul is not None and twin_ok(draw) and _dense_inner(raw) and _dense_inner(gy) and (residual is None or _dense_inner(residual)):
        pair = twin_empty(draw)
        with torch.cuda.device(raw.device):
            check(_lib.lib().snvc_act_backward_apply_twin(
                _ptr(raw), _ptr(gy), _ptr(residual), _ptr(scale), _ptr(shift), _ptr(coef_g), _ptr(coef_raw), _ptr(coef_const), _ptr(draw),
                _ptr(g_out), _ptr(pair), _lo_ptr(pair), _ptr(twin_mul), n, c, s, _batch_stride(raw), _batch_stride(gy),
                _batch_stride(residual) if residual is not None else 0, _batch_stride(pair), 1 if per_sample else 0, flags, _ptr(amax),
                _stream(raw)), "snvc_act_backward_apply_twin")
        tag_twin(draw, pair, twin_mul)
        return draw, g_out
    with torch.cuda.device(raw.device):
        check(_lib.lib().snvc_act_backward_apply_amax(_ptr(raw), _ptr(gy), _ptr(residual), _ptr(scale), _ptr(shift),
                                                      _ptr(coef_g), _ptr(coef_raw), _ptr(coef_const), _ptr(draw), _ptr(g_out),
                                                      n, c, s, _batch_stride(raw), _batch_stride(gy),
                                                      _batch_stride(residual) if residual is not None else 0,
                                                      1 if per_sample else 0, flags, _ptr(amax), _stream(raw)), "snvc_act_backward_apply")
    return draw, g_out


# ------------------------------------------------------------------------------ norm / elementwise
def norm_stats(x, gamma, beta, groups: int, per_sample: bool, eps: float):
    """(scale, shift, mean, var) for GroupNorm (per_sample) or batch-stat BatchNorm."""
    _gpu(x, "x")
    if not _dense_inner(x):
        x = x.contiguous()
    n, c = x.shape[0], x.shape[1]
    s = x[0, 0].numel()
    outer = n if per_sample else 1
    scale = torch.empty((outer, c), dtype=torch.float32, device=x.device)
    shift = torch.empty_like(scale)
    mean = torch.empty((outer, groups), dtype=torch.float32, device=x.device)
    var = torch.empty_like(mean)
    ws = torch.empty(_lib.lib().snvc_norm_workspace_bytes(n, c, groups), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_norm_stats(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift), _ptr(mean),
                                         _ptr(var), _ptr(ws), n, c, s, _batch_stride(x), groups,
                                         1 if per_sample else 0, float(eps), _stream(x)), "snvc_norm_stats")
    return scale, shift, mean, var


def affine_act(x, scale, shift, residual=None, flags=0, per_sample=False, out=None, amax: Optional[torch.Tensor] = None,
               twin_mul: Optional[torch.Tensor] = None):
    """``amax``: a zeroed int32 word that receives the bit pattern of max|out| (amax_word).  ``twin_mul`` (r6): also write out's split
    twin scaled by that device float and tag it onto ``out`` (snvc_affine_act_twin)."""
    _gpu(x, "x")
    if not _dense_inner(x):
        x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    elif not _dense_inner(out) or tuple(out.shape) != tuple(x.shape):
        raise RuntimeError("affine_act `out` must be a channel-dense view of x's shape")
    if residual is not None and not _dense_inner(residual):
        residual = residual.contiguous()
    n, c = x.shape[0], x.shape[1]
    s = x[0, 0].numel() if n else 0
    if x.numel() == 0:
        return out
    if twin_mul is not None and twin_ok(out) and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 and (
            residual is None or residual.data_ptr() % 16 == 0):
        pair = twin_empty(out)
        with torch.cuda.device(x.device):
            check(_lib.lib().snvc_affine_act_twin(_ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out), _ptr(pair), _lo_ptr(pair),
                                                  _ptr(twin_mul), n, c, s, _batch_stride(x), _batch_stride(out),
                                                  _batch_stride(residual) if residual is not None else 0, _batch_stride(pair),
                                                  1 if per_sample else 0, flags, _ptr(amax), _stream(x)), "snvc_affine_act_twin")
        tag_twin(out, pair, twin_mul)
        return out
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_affine_act_amax(_ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out), n, c, s,
                                              _batch_stride(x), _batch_stride(out),
                                              _batch_stride(residual) if residual is not None else 0,
                                              1 if per_sample else 0, flags, _ptr(amax), _stream(x)), "snvc_affine_act")
    return out


def mul_broadcast(feat, occ, out=None):
    """out[n,c,...] = feat[n,c,...] * occ[n,0,...] (second half of the cat in vernier.py:433)."""
    _gpu(feat, "feat"); _gpu(occ, "occ")
    feat, occ = feat.contiguous(), occ.contiguous()
    n, c = feat.shape[0], feat.shape[1]
    s = feat[0, 0].numel() if n else 0
    if occ.numel() != n * s:
        raise RuntimeError("occupancy must be [N,1,D,H,W] matching the feature volume")
    if out is None:
        out = torch.empty_like(feat)
    elif not _dense_inner(out) or tuple(out.shape) != tuple(feat.shape):
        raise RuntimeError("mul_broadcast `out` must be a channel-dense view of feat's shape")
    if feat.numel() == 0:
        return out
    with torch.cuda.device(feat.device):
        check(_lib.lib().snvc_mul_broadcast(_ptr(feat), _ptr(occ), _ptr(out), n, c, s, _batch_stride(out),
                                            _stream(feat)), "snvc_mul_broadcast")
    return out


def avgpool_depth4(x):
    """AvgPool3d((4,1,1),(4,1,1)) (vernier.py:289) on [N,C,D,H,W] -> [N,C,D//4,H,W]."""
    _gpu(x, "x")
    x = x.contiguous()
    n, c, d, h, w = x.shape
    y = torch.empty((n, c, d // 4, h, w), dtype=torch.float32, device=x.device)
    if y.numel() == 0:
        return y
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_avgpool_depth4(_ptr(x), _ptr(y), n, c, d, h * w, _stream(x)), "snvc_avgpool_depth4")
    return y


def avgpool_depth4_backward(gy, d):
    """Adjoint of ``avgpool_depth4`` for an input of depth ``d``: [N,C,d//4,H,W] -> [N,C,d,H,W] (snvc_avgpool_depth4_backward)."""
    _gpu(gy, "gy")
    gy = gy.contiguous()
    n, c, dq, h, w = gy.shape
    if dq != d // 4 or gy.dtype != torch.float32:
        raise RuntimeError("avgpool_depth4_backward: grad must be float32 [N,C,d//4,H,W]")
    gx = torch.empty((n, c, d, h, w), dtype=torch.float32, device=gy.device)
    if gx.numel():
        with torch.cuda.device(gy.device):
            check(_lib.lib().snvc_avgpool_depth4_backward(_ptr(gy), _ptr(gx), n, c, d, h * w, _stream(gy)), "snvc_avgpool_depth4_backward")
    return gx


class AvgPoolDepth4Fn(torch.autograd.Function):
    """``F.avg_pool3d(x, (4,1,1), (4,1,1))`` (reference vernier.py:289,436) under autograd on the HIP kernels, both directions."""

    @staticmethod
    def forward(ctx, x):
        ctx.depth = x.size(2)
        return avgpool_depth4(x)

    @staticmethod
    def backward(ctx, gy):
        return avgpool_depth4_backward(gy, ctx.depth)


def zero_stuff2x(x):
    """[N,C,H,W] -> [N,C,2H,2W] with y[..., 2i, 2j] = x[..., i, j] and zeros elsewhere (see snvc_zero_stuff2x)."""
    _gpu(x, "x")
    x = x.contiguous()
    n, c, h, w = x.shape
    y = torch.empty((n, c, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    if y.numel():
        with torch.cuda.device(x.device):
            check(_lib.lib().snvc_zero_stuff2x(_ptr(x), _ptr(y), n * c, h, w, _stream(x)), "snvc_zero_stuff2x")
    return y


def disparity_regression(x, depth):
    """disparityregression.forward (submodule.py:81-83): [N,D,H,W] x [D] -> [N,H,W]."""
    _gpu(x, "x"); _gpu(depth, "depth")
    x, depth = x.contiguous(), depth.contiguous()
    n, d, h, w = x.shape
    if depth.numel() != d:
        raise RuntimeError("depth must have one entry per disparity plane")
    out = torch.empty((n, h, w), dtype=torch.float32, device=x.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_disparity_regression(_ptr(x), _ptr(depth), _ptr(out), n, d, h * w, _stream(x)),
              "snvc_disparity_regression")
    return out


def argmax_rows(x2d) -> Tuple[torch.Tensor, torch.Tensor]:
    """np.argmax(ncf.reshape(N, P, -1), axis=2) (vernier.py:693) on the GPU -> (int64 idx, max)."""
    _gpu(x2d, "x")
    x2d = x2d.contiguous()
    r, l = x2d.shape
    idx = torch.empty(r, dtype=torch.int64, device=x2d.device)
    val = torch.empty(r, dtype=torch.float32, device=x2d.device)
    if r == 0:
        return idx, val
    with torch.cuda.device(x2d.device):
        check(_lib.lib().snvc_argmax_rows(_ptr(x2d), _ptr(idx), _ptr(val), r, l, _stream(x2d)), "snvc_argmax_rows")
    return idx, val


# ------------------------------------------------------------------------------ N3: PSV -> 3D grid resampling
def volume_resample(x, grid, align_corners: bool = False) -> torch.Tensor:
    """F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros") for a 5-D input: x [N,C,D,H,W] float32,
    grid [N,Do,Ho,Wo,3] (or [N,V,3]) with (gx -> W, gy -> H, gz -> D) -> [N,C,Do,Ho,Wo] (or [N,C,V])."""
    _gpu(x, "x"); _gpu(grid, "grid")
    if x.dtype != torch.float32 or grid.dtype != torch.float32 or x.dim() != 5 or grid.size(-1) != 3 or grid.size(0) != x.size(0):
        raise RuntimeError("volume_resample needs float32 x [N,C,D,H,W] and grid [N,...,3]")
    if not _dense_inner(x):
        x = x.contiguous()
    n, c, d, h, w = x.shape
    out_sp = tuple(grid.shape[1:-1])
    g = grid.reshape(n, -1, 3).contiguous()
    v = g.size(1)
    out = torch.empty((n, c, v), dtype=torch.float32, device=x.device)
    if out.numel():
        with torch.cuda.device(x.device):
            check(_lib.lib().snvc_volume_resample(_ptr(x), _ptr(g), _ptr(out), n, c, d, h, w, v, 1 if align_corners else 0,
                                                  _batch_stride(x), 0, _stream(x)), "snvc_volume_resample")
    return out.view((n, c) + out_sp)


def rect_to_psv_grid(pts_rect, P, origin, span) -> torch.Tensor:
    """project_rect_to_image (snvc/utils/torch_utils.py:37-45) + normalisation: pts_rect [V,3] (GPU, float32), P a 3x4
    projection matrix (host), origin / span = (u0, v0, z0) / (u_span, v_span, z_span) -> grid [V,3] in [-1,1]."""
    import numpy as np
    _gpu(pts_rect, "pts_rect")
    pts = pts_rect.contiguous()
    if pts.dtype != torch.float32 or pts.dim() != 2 or pts.size(1) != 3:
        raise RuntimeError("pts_rect must be float32 [V,3]")
    p = np.ascontiguousarray(np.asarray(P.cpu() if torch.is_tensor(P) else P, dtype=np.float32).reshape(12))
    grid = torch.empty_like(pts)
    if pts.numel():
        with torch.cuda.device(pts.device):
            check(_lib.lib().snvc_rect_to_psv_grid(_ptr(pts), p.ctypes.data_as(ctypes.c_void_p), _ptr(grid), pts.size(0),
                                                   float(origin[0]), float(span[0]), float(origin[1]), float(span[1]),
                                                   float(origin[2]), float(span[2]), _stream(pts)), "snvc_rect_to_psv_grid")
    return grid


# ------------------------------------------------------------------------------ fp16-storage mode (C8 layout)
# A C8 tensor is a torch.float16 tensor of shape [N, C/8, D, H, W, 8] (include/snvc_hip.h, "fp16-storage mode"):
# channel c of voxel (d,h,w) is t[n, c // 8, d, h, w, c % 8].  t[:, g0:g1] is a channel slice (a view).
def _c8_check(t: torch.Tensor, name: str):
    _gpu(t, name)
    if t.dtype != torch.float16 or t.dim() != 6 or t.size(5) != 8:
        raise RuntimeError(f"{name} must be a C8 tensor: float16 [N, C/8, D, H, W, 8], got {tuple(t.shape)} {t.dtype}")
    if not _dense_inner(t) or t.data_ptr() % 16:
        raise RuntimeError(f"{name} must be dense below dim 0 and 16-byte aligned")


def to_c8(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float32 [N,C,D,H,W] -> C8 half [N,ceil(C/8),D,H,W,8] (missing channels are zero)."""
    _gpu(x, "x")
    if x.dtype != torch.float32 or x.dim() != 5:
        raise RuntimeError("to_c8 needs a float32 [N,C,D,H,W] tensor")
    if not _dense_inner(x):
        x = x.contiguous()
    n, c = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])
    if out is None:
        out = torch.empty((n, (c + 7) // 8) + sp + (8,), dtype=torch.float16, device=x.device)
    else:
        _c8_check(out, "out")
    if x.numel() == 0:
        return out
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_f16_from_ncdhw(_ptr(x), _ptr(out), n, c, math.prod(sp), _batch_stride(x), _batch_stride(out),
                                             _stream(x)), "snvc_f16_from_ncdhw")
    return out


def from_c8(x: torch.Tensor, channels: Optional[int] = None) -> torch.Tensor:
    """C8 half -> float32 [N,C,D,H,W] (C = channels or all 8*G)."""
    _c8_check(x, "x")
    n, g = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:5])
    c = channels if channels is not None else 8 * g
    y = torch.empty((n, c) + sp, dtype=torch.float32, device=x.device)
    if y.numel() == 0:
        return y
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_f16_to_ncdhw(_ptr(x), _ptr(y), n, c, math.prod(sp), _batch_stride(x), 0, _stream(x)),
              "snvc_f16_to_ncdhw")
    return y


def voxel_gather_forward_f16(left, right, l_pts, r_pts, resolution) -> torch.Tensor:
    """_sample_2d_feat(aggregate='concat') (vernier.py:323-349) with a C8 half result [N, 2F/8, V, 8]."""
    _check_gather(left, right, l_pts, r_pts)
    n, f, hf, wf = left.shape
    if f % 8:
        raise RuntimeError("the fp16-storage gather needs F % 8 == 0")
    v = l_pts.size(2)
    out = torch.empty((n, 2 * f // 8, v, 8), dtype=torch.float16, device=left.device)
    if out.numel() == 0:
        return out
    left, right, l_pts, r_pts = left.contiguous(), right.contiguous(), l_pts.contiguous(), r_pts.contiguous()
    with torch.cuda.device(left.device):
        ws = torch.empty(_lib.lib().snvc_voxel_gather_workspace_floats(n, f, hf, wf), dtype=torch.float32, device=left.device)
        check(_lib.lib().snvc_voxel_gather_forward_f16(_ptr(left), _ptr(right), _ptr(l_pts), _ptr(r_pts), _ptr(out),
                                                       _ptr(ws), n, f, hf, wf, v, float(resolution[1]),
                                                       float(resolution[0]), _stream(left)), "voxel_gather_forward_f16")
    return out


def voxel_gather_forward_split(left, right, l_pts, r_pts, resolution, mul_dev) -> torch.Tensor:
    """_sample_2d_feat(aggregate='concat') (vernier.py:323-349) written as a split C8 pair [N, 2, 2F/8, V, 8] (``to_split``
    layout) holding the gather's fp32 result times the device scalar ``mul_dev`` (``split_scale_for(left, right)``: a bilinear
    sample never exceeds the features' maximum) -- bit-identical to ``to_split(voxel_gather_forward(...), mul_dev=mul_dev)``
    without the fp32 tensor, its maximum pass and the layout pass.  Raises ``Unsupported`` for feature planes beyond the
    LDS-staged form."""
    _check_gather(left, right, l_pts, r_pts)
    n, f, hf, wf = left.shape
    if f % 8:
        raise RuntimeError("the split gather needs F % 8 == 0")
    if mul_dev.dtype != torch.float32 or mul_dev.numel() != 1 or mul_dev.device != left.device:
        raise RuntimeError("mul_dev: a one-element float32 tensor on the features' device")
    v = l_pts.size(2)
    out = torch.empty((n, 2, 2 * f // 8, v, 8), dtype=torch.float16, device=left.device)
    if out.numel() == 0:
        return out
    left, right, l_pts, r_pts = left.contiguous(), right.contiguous(), l_pts.contiguous(), r_pts.contiguous()
    with torch.cuda.device(left.device):
        ws = torch.empty(_lib.lib().snvc_voxel_gather_workspace_floats(n, f, hf, wf), dtype=torch.float32, device=left.device)
        check(_lib.lib().snvc_voxel_gather_forward_split(_ptr(left), _ptr(right), _ptr(l_pts), _ptr(r_pts), _ptr(out), _ptr(out[:, 1]),
                                                         _ptr(mul_dev), _ptr(ws), n, f, hf, wf, v, out.stride(0),
                                                         float(resolution[1]), float(resolution[0]), _stream(left)),
              "voxel_gather_forward_split")
    return out


X3_Q16 = [True]        # False: the 32x32x16 kernel forms everywhere (rounds up to mid r4)
X3_Q16_K5 = [True]     # ... and the plain 5^3 layers (quads over all 125 taps)
X3_Q16_S2 = [True]     # stride-2 3x3x3 layers with a split output and no residual: the 16x16x32 form (False: the serial-plane 32x32x16 form)
X3_Q16_MIN_JOBS = [256]   # 3x3x3 layers: (tile, 32-channel block) jobs from which the 16x16x32 form is picked (r4: 1024).  hg conv4 at cfg2
#                           (432 jobs): 0.008-0.023 ms per step in five interleaved A/B runs (tools/ab_step.py, profiles/r5/ab_step_*.txt)


class Conv3dLayerF16:
    """A Conv3d / ConvTranspose3d layer prepared for the fp16-storage kernels (snvc_f16_conv3d_*): same
    geometry rules as Conv3dLayer; the fp32 parameter is rounded to half when packed."""

    def __init__(self, weight: torch.Tensor, ksize: int, stride: int, pad: int, dilation: int, transposed: bool):
        _gpu(weight, "weight")
        if weight.dtype != torch.float32:
            raise RuntimeError("conv3d weights must be float32 (they are rounded to half when packed)")
        self.transposed = bool(transposed)
        if transposed:
            self.cin, self.cout = weight.shape[0], weight.shape[1]
        else:
            self.cout, self.cin = weight.shape[0], weight.shape[1]
        if tuple(weight.shape[2:]) != (ksize,) * 3:
            raise RuntimeError("only cubic kernels are on the path")
        self.ksize, self.stride, self.pad, self.dilation = int(ksize), int(stride), int(pad), int(dilation)
        # r4: the 7^3, 5^3 and dilated 5^3 layers take the 16x16x32 kernel form (decided here: packing and launch must agree)
        self.q16 = bool(X3_Q16[0] and not transposed and self.stride == 1 and self.cout % 32 == 0 and
                        (self.ksize == 7 or (self.ksize == 5 and (self.dilation == 2 or (self.dilation == 1 and X3_Q16_K5[0])))))
        probe = self._desc(1, (16, 16, 32), 0)
        nbytes = _lib.lib().snvc_f16_conv3d_packed_weight_bytes(ctypes.byref(probe))
        if nbytes < 0:
            check(1, "snvc_f16_conv3d_packed_weight_bytes")
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
        with torch.cuda.device(weight.device):
            check(_lib.lib().snvc_f16_conv3d_pack_weights(ctypes.byref(probe), _ptr(weight.detach().contiguous()),
                                                          _ptr(self.packed), _stream(weight)), "snvc_f16_conv3d_pack_weights")

    out_spatial = Conv3dLayer.out_spatial

    def _desc(self, *a, **k):
        d = Conv3dLayer._desc(self, *a, **k)
        if self.q16:
            d.algo |= _lib.ALGO_X3_Q16
        return d

    def __call__(self, x, scale=None, bias=None, residual=None, flags=0, out=None):
        """y = epilogue(conv(x)) on C8 tensors.  Cout == 1: returns the fp32 plane [N,1,D,H,W] (Sigmoid allowed)."""
        _c8_check(x, "x")
        if x.size(1) * 8 != self.cin:
            raise RuntimeError(f"conv3d input must have {self.cin} channels (C8), got {x.size(1) * 8}")
        n = x.size(0)
        in_sp = tuple(x.shape[2:5])
        out_sp = self.out_spatial(in_sp)
        if min(out_sp) < 1:
            raise RuntimeError("conv3d output would be empty")
        plane = self.cout == 1
        y32 = None
        if plane:
            y32 = torch.empty((n, 1) + out_sp, dtype=torch.float32, device=x.device)
            out = None
        elif out is None:
            out = torch.empty((n, self.cout // 8) + out_sp + (8,), dtype=torch.float16, device=x.device)
        else:
            _c8_check(out, "out")
            if tuple(out.shape) != (n, self.cout // 8) + out_sp + (8,):
                raise RuntimeError("conv3d `out` must be a C8 view of the output shape")
        if residual is not None:
            _c8_check(residual, "residual")
            if tuple(residual.shape) != (n, self.cout // 8) + out_sp + (8,):
                raise RuntimeError("residual must have the output's shape")
        if n == 0:
            return y32 if plane else out
        d = self._desc(n, in_sp, flags, _batch_stride(x), _batch_stride(out) if out is not None else 0,
                       _batch_stride(residual) if residual is not None else 0)
        with torch.cuda.device(x.device):
            check(_lib.lib().snvc_f16_conv3d_forward(ctypes.byref(d), _ptr(x), _ptr(self.packed), _ptr(scale), _ptr(bias),
                                                     _ptr(residual), _ptr(out), _ptr(y32), _stream(x)),
                  "snvc_f16_conv3d_forward")
        return y32 if plane else out


# ------------------------------------------------------------------------------------ split mode ("f16x3")
def _split_check(t: torch.Tensor, name: str):
    _gpu(t, name)
    if t.dtype != torch.float16 or t.dim() != 7 or t.size(1) != 2 or t.size(6) != 8:
        raise RuntimeError(f"{name} must be a split C8 tensor: float16 [N, 2 (hi | lo), C/8, D, H, W, 8], got {tuple(t.shape)} {t.dtype}")
    # dense from the channel-group axis down (a slice of channel groups of a larger pair qualifies: the plane and batch strides
    # are passed to the kernels), 16-byte aligned pieces
    exp = 1
    for size, stride in zip(reversed(t.shape[2:]), reversed(t.stride()[2:])):
        if size != 1 and stride != exp:
            raise RuntimeError(f"{name} must be dense below its channel-group axis")
        exp *= size
    if t.data_ptr() % 16 or t.stride(1) % 8 or t.stride(0) % 8:
        raise RuntimeError(f"{name} must be 16-byte aligned with plane / batch strides that are multiples of 8")


def _lo_ptr(t: torch.Tensor):
    return ctypes.c_void_p(t.data_ptr() + 2 * t.stride(1)) if t.numel() else ctypes.c_void_p(0)


def to_split(x: torch.Tensor, exp: int = 0, out: Optional[torch.Tensor] = None, mul_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float32 [N,C,D,H,W] -> split C8 [N, 2, ceil(C/8), D, H, W, 8] half with  x * 2**exp = hi + lo  (snvc_f16x3_from_ncdhw).
    ``mul_dev``: a one-element float32 device tensor holding the scale (a power of two) instead of ``2**exp`` -- for an exponent
    derived from the data on the device (``split_scale_for``) without a host round trip."""
    _gpu(x, "x")
    if x.dtype != torch.float32 or x.dim() != 5:
        raise RuntimeError("to_split needs a float32 [N,C,D,H,W] tensor")
    if not _dense_inner(x):
        x = x.contiguous()
    n, c = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])
    if out is None:
        out = torch.empty((n, 2, (c + 7) // 8) + sp + (8,), dtype=torch.float16, device=x.device)
    else:
        _split_check(out, "out")
    if x.numel() == 0:
        return out
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_f16x3_from_ncdhw(_ptr(x), _ptr(out), _lo_ptr(out), n, c, math.prod(sp), _batch_stride(x), _batch_stride(out),
                                               float(2.0 ** exp), _ptr(mul_dev), _stream(x)), "snvc_f16x3_from_ncdhw")
    return out


def affine_act_split(raw, scale, shift, out_exp: int, residual=None, res_exp: int = 0, flags: int = 0, per_sample: bool = True, out=None,
                     overflow=None):
    """The norm + residual + activation of a GroupNorm layer behind a split-mode convolution (snvc_f16x3_affine_from_ncdhw): ``raw`` the
    float32 [N,C,D,H,W] conv result, ``scale`` / ``shift`` [N or 1, C] from ``norm_stats``, ``residual`` a split pair holding values *
    2**res_exp; returns the split pair of ``act(scale * raw + shift [+ res]) [+ res]`` times 2**out_exp (clamped and flagged)."""
    _gpu(raw, "raw")
    if raw.dtype != torch.float32 or raw.dim() != 5:
        raise RuntimeError("affine_act_split needs a float32 [N,C,D,H,W] tensor")
    if not _dense_inner(raw):
        raw = raw.contiguous()
    n, c = raw.shape[0], raw.shape[1]
    sp = tuple(raw.shape[2:])
    shape = (n, 2, (c + 7) // 8) + sp + (8,)
    if out is None:
        out = torch.empty(shape, dtype=torch.float16, device=raw.device)
    else:
        _split_check(out, "out")
        if tuple(out.shape) != shape:
            raise RuntimeError("affine_act_split: out has the wrong shape")
    want_res = bool(flags & (EPI_ADD_PRE | EPI_ADD_POST))
    if want_res != (residual is not None):
        raise RuntimeError("affine_act_split: a residual exactly when EPI_ADD_PRE / EPI_ADD_POST is set")
    if residual is not None:
        _split_check(residual, "residual")
        if tuple(residual.shape) != shape:
            raise RuntimeError("residual must have the result's shape (split C8)")
    for t in (scale, shift):
        if t.dtype != torch.float32 or tuple(t.shape) != ((n if per_sample else 1), c) or not t.is_contiguous():
            raise RuntimeError("scale / shift must be contiguous float32 [N or 1, C]")
    if raw.numel() == 0:
        return out
    null = ctypes.c_void_p(0)
    with torch.cuda.device(raw.device):
        check(_lib.lib().snvc_f16x3_affine_from_ncdhw(
            _ptr(raw), _ptr(scale), _ptr(shift), _ptr(residual) if residual is not None else null,
            _lo_ptr(residual) if residual is not None else null, _ptr(out), _lo_ptr(out), _ptr(overflow), n, c, math.prod(sp),
            _batch_stride(raw), _batch_stride(out), _batch_stride(residual) if residual is not None else 0, 1 if per_sample else 0,
            int(flags), float(2.0 ** out_exp), float(2.0 ** -res_exp), _stream(raw)), "snvc_f16x3_affine_from_ncdhw")
    return out


def split_scale_for(*tensors) -> torch.Tensor:
    """A one-element device tensor holding the power of two that puts max|t| over the given tensors into [2^13, 2^14) (1 for an
    all-zero or non-finite input): the scale of a split pair whose range is only known from the data.  Device-side, no sync.
    Contiguous float32 tensors take one HIP launch each (``split_scale_of``) and one ``minimum`` (the scale is a decreasing function
    of the maximum); r4's form below -- ten small torch launches, 50-70 us in front of a 0.12-0.4 ms gather -- is the fallback."""
    if tensors and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 and t.numel() > 0
                       for t in tensors):
        out = None
        for t in tensors:
            s_ = split_scale_of(t.detach())
            out = s_ if out is None else torch.minimum(out, s_)
        return out
    amax = None
    for t in tensors:
        lo, hi = torch.aminmax(t.detach())                  # one read pass (abs().amax() would write a temporary of t's size)
        a = torch.maximum(-lo, hi)
        amax = a if amax is None else torch.maximum(amax, a)
    amax = amax.float().reshape(1)
    e = torch.floor(torch.log2(16384.0 / amax))
    e = torch.where(torch.isfinite(e), e, torch.zeros_like(e)).clamp_(-24, 40)
    return torch.exp2(e)


_SCALE_SCRATCH = {}
_SCALE_SCRATCH_LOCK = threading.Lock()


def _scale_scratch(device) -> torch.Tensor:
    """The 8-byte (running maximum, arrival counter) scratch of the scale launches, one per (device, STREAM): launches on one stream are
    ordered and may share it (the kernels leave it zeroed); two streams -- or two threads, each on its own stream -- of one device
    must not (a shared scratch races on the maximum and on the counter and yields a wrong scale)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    scratch = _SCALE_SCRATCH.get(key)
    if scratch is None:
        with _SCALE_SCRATCH_LOCK:
            scratch = _SCALE_SCRATCH.get(key)
            if scratch is None:
                scratch = _SCALE_SCRATCH[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return scratch


def split_scale_of(t: torch.Tensor) -> torch.Tensor:
    """``split_scale_for(t)`` for ONE contiguous float32 tensor in one launch (snvc_f16x3_split_scale): the power of two that puts
    max|t| into [2^13, 2^14), as a one-element device tensor; no sync."""
    _gpu(t, "t")
    if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() % 16:
        return split_scale_for(t)
    scratch = _scale_scratch(t.device)
    out = torch.empty(1, dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(_lib.lib().snvc_f16x3_split_scale(_ptr(t), t.numel(), _ptr(scratch), _ptr(out), _stream(t)), "snvc_f16x3_split_scale")
    return out


def sheared_upsample_split(right, q: int, wu: int, off: int, mul_dev: torch.Tensor):
    """``sheared_upsample`` written as the split pair [N, 2, C/8, 1, H, wu, 8] (value * mul_dev = hi + lo): what the split-mode 3 x 7
    layer reads (snvc_sheared_upsample_split)."""
    _gpu(right, "right")
    if right.dtype != torch.float32 or right.dim() != 4:
        raise RuntimeError("sheared_upsample_split needs a float32 [N,C,H,W] tensor")
    right = right.contiguous()
    n, c, h, w = right.shape
    out = torch.empty((n, 2, (c + 7) // 8, 1, h, wu, 8), dtype=torch.float16, device=right.device)
    if out.numel():
        with torch.cuda.device(right.device):
            check(_lib.lib().snvc_sheared_upsample_split(_ptr(right), _ptr(out), _lo_ptr(out), _ptr(mul_dev), n, c, h, w, int(q), int(wu), int(off),
                                                         _stream(right)), "snvc_sheared_upsample_split")
    return out


class Conv2dLayerX3:
    """A depth-1 convolution (3x7 or 3x3, stride 1, "same" zero padding, no bias) in split mode: x a split pair [N,2,Cin/8,1,H,W,8]
    holding values * x_mul_dev, result float32 [N,Cout,H,W] = scale * conv(x) + bias (snvc_f16x3_conv2d_*; r5: the sheared first
    layer's G / G' on the half pipe)."""

    def __init__(self, weight: torch.Tensor):
        _gpu(weight, "weight")
        if weight.dtype != torch.float32 or weight.dim() != 4:
            raise RuntimeError("Conv2dLayerX3 needs a float32 [Cout,Cin,kh,kw] weight")
        self.cout, self.cin, self.kh, self.kw = (int(v) for v in weight.shape)
        nbytes = _lib.lib().snvc_f16x3_conv2d_packed_weight_bytes(self.cout, self.cin, self.kh, self.kw)
        if nbytes < 0:
            raise Unsupported("Conv2dLayerX3: 3x7 or 3x3 kernels, Cin % 8 == 0, Cout % 32 == 0")
        w = weight.detach().contiguous()
        wmax = float(w.abs().max().item()) if w.numel() else 1.0
        self.w_exp = 14 - math.frexp(wmax)[1] if wmax > 0 and math.isfinite(wmax) else 0
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        with torch.cuda.device(w.device):
            check(_lib.lib().snvc_f16x3_conv2d_pack_weights(_ptr(w), self.cout, self.cin, self.kh, self.kw, _ptr(self.packed),
                                                            float(2.0 ** self.w_exp), _stream(w)), "snvc_f16x3_conv2d_pack_weights")

    def __call__(self, x, x_mul_dev=None, scale=None, bias=None, flags: int = 0, x_exp: int = 0):
        _split_check(x, "x")
        if x.size(2) * 8 != self.cin or x.size(3) != 1:
            raise RuntimeError(f"Conv2dLayerX3 input must be a split pair [N,2,{self.cin // 8},1,H,W,8]")
        n, h, w = x.size(0), x.size(4), x.size(5)
        y = torch.empty((n, self.cout, h, w), dtype=torch.float32, device=x.device)
        if y.numel():
            with torch.cuda.device(x.device):
                check(_lib.lib().snvc_f16x3_conv2d_forward(_ptr(x), _lo_ptr(x), _ptr(self.packed), _ptr(scale), _ptr(bias), _ptr(y), n, self.cin,
                                                           self.cout, h, w, self.kh, self.kw, float(2.0 ** -(self.w_exp + x_exp)), _ptr(x_mul_dev),
                                                           int(flags), _stream(x)), "snvc_f16x3_conv2d_forward")
        return y


def sheared_prep_x3(right, q: int, wu: int, off: int, wu_col: int, off_col: int, lay_g: "Conv2dLayerX3", lay_col: "Conv2dLayerX3", ws: dict):
    """The sheared first layer's 2D prep in one host call (snvc_sheared_prep_x3): returns (G [N,3C,H,wu], G' [N,3C,H,wu_col]) float32.
    ``ws``: a dict the caller keeps (workspaces and results are allocated once per shape and reused: the results are consumed by the
    expand pass of the same call)."""
    _gpu(right, "right")
    right = right.contiguous()
    n, c, h, w = right.shape
    key = (n, c, h, w, wu, wu_col, right.device)
    if ws.get("key") != key:
        dev = right.device
        ws.clear()
        ws.update(key=key, rq=torch.empty(n * 2 * c * h * wu, dtype=torch.float16, device=dev),
                  rq2=torch.empty(n * 2 * c * h * wu_col, dtype=torch.float16, device=dev), mul=torch.empty(1, dtype=torch.float32, device=dev),
                  g=torch.empty((n, lay_g.cout, h, wu), dtype=torch.float32, device=dev),
                  gcol=torch.empty((n, lay_col.cout, h, wu_col), dtype=torch.float32, device=dev))
    scratch = _scale_scratch(right.device)
    with torch.cuda.device(right.device):
        check(_lib.lib().snvc_sheared_prep_x3(_ptr(right), n, c, h, w, int(q), int(wu), int(off), int(wu_col), int(off_col), _ptr(lay_g.packed),
                                              _ptr(lay_col.packed), lay_g.cout, float(2.0 ** -lay_g.w_exp), float(2.0 ** -lay_col.w_exp),
                                              _ptr(ws["rq"]), _ptr(ws["rq2"]), _ptr(scratch), _ptr(ws["mul"]), _ptr(ws["g"]), _ptr(ws["gcol"]),
                                              _stream(right)), "snvc_sheared_prep_x3")
    return ws["g"], ws["gcol"]


def conv2d_x3_from_f32(x, layers, ws: dict):
    """``[layer(split(x)) for layer in layers]`` for depth-1 split layers of one kernel size on a float32 [N,C,H,W] tensor, in ONE host call
    (snvc_f16x3_conv2d_from_f32: the scale from x's own maximum, its split pair, the layers).  ``ws``: a dict the caller keeps
    (workspace and results allocated once per shape; the results are valid until the next call with the same dict)."""
    _gpu(x, "x")
    x = x.contiguous()
    if x.dtype != torch.float32 or x.dim() != 4:
        raise RuntimeError("conv2d_x3_from_f32 needs a float32 [N,C,H,W] tensor")
    n, c, h, w = x.shape
    key = (n, c, h, w, tuple(id(l) for l in layers), x.device)
    if ws.get("key") != key:
        ws.clear()
        ws.update(key=key, split=torch.empty(n * 2 * c * h * w, dtype=torch.float16, device=x.device),
                  mul=torch.empty(1, dtype=torch.float32, device=x.device),
                  y=[torch.empty((n, l.cout, h, w), dtype=torch.float32, device=x.device) for l in layers])
        k = len(layers)
        ws["c_packed"] = (ctypes.c_void_p * k)(*[l.packed.data_ptr() for l in layers])
        ws["c_cout"] = (ctypes.c_int64 * k)(*[l.cout for l in layers])
        ws["c_mul"] = (ctypes.c_float * k)(*[2.0 ** -l.w_exp for l in layers])
        ws["c_y"] = (ctypes.c_void_p * k)(*[t.data_ptr() for t in ws["y"]])
        ws["layers"] = list(layers)          # keeps the packed weights the pointer array refers to alive
    scratch = _scale_scratch(x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_f16x3_conv2d_from_f32(_ptr(x), n, c, h, w, layers[0].kh, layers[0].kw, len(layers), ws["c_packed"], ws["c_cout"],
                                                    ws["c_mul"], ws["c_y"], _ptr(ws["split"]), _ptr(scratch), _ptr(ws["mul"]), _stream(x)),
              "snvc_f16x3_conv2d_from_f32")
    return ws["y"]


def from_split(x: torch.Tensor, exp: int = 0, channels: Optional[int] = None) -> torch.Tensor:
    """split C8 -> float32 [N,C,D,H,W]: (hi + lo) * 2**-exp."""
    _split_check(x, "x")
    n, g = x.shape[0], x.shape[2]
    sp = tuple(x.shape[3:6])
    c = channels if channels is not None else 8 * g
    y = torch.empty((n, c) + sp, dtype=torch.float32, device=x.device)
    if y.numel() == 0:
        return y
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_f16x3_to_ncdhw(_ptr(x), _lo_ptr(x), _ptr(y), n, c, math.prod(sp), _batch_stride(x), 0, float(2.0 ** -exp),
                                             _stream(x)), "snvc_f16x3_to_ncdhw")
    return y


# workgroups below which the half-height tile forms are picked (tools flip these).  Measured at cfg2 (tools/time_hg.py): the transposed
# layer gains 3-8 % (hg conv5 141 -> 137 us); the stride-2 layers LOSE -- hg conv3 (432 workgroups) 91 -> 170 us, hg conv1 398 -> 473:
# twice the workgroups stream twice the weights (17 TB/s of L1 -> register traffic at hg conv3's size), so that form is never picked by itself
X3_SMALL_BELOW = {"stride2": 0, "transposed": 4096}


_ONES_ZEROS = {}


def _ones_zeros(c: int, device):
    """Read-only per-channel constants (ones, zeros) shared by every layer of that width on the device: the training step rebuilds its
    split layers whenever the weights move, and two fill launches per layer and step add up."""
    key = (c, device)
    hit = _ONES_ZEROS.get(key)
    if hit is None:
        hit = _ONES_ZEROS[key] = (torch.ones(c, device=device), torch.zeros(c, device=device))
    return hit


class Conv3dLayerX3:
    """nn.Conv3d(k3, p1, stride 1 | 2) / nn.ConvTranspose3d(k3, s2, p1, op1) prepared for the split-mode kernels
    (snvc_f16x3_conv3d_*): the fp32 contraction at fp32 accuracy on the half pipe.  The weights are packed as (hi, lo) of
    w * 2**w_exp  with w_exp chosen so that max|w| lands in [2^13, 2^14): both parts then keep their full 11 bits.
    Stride-1 layers have three kernel forms (64-channel blocks on 4x4x32 tiles; 32-channel blocks; 32-channel blocks on 2x4x32
    tiles) whose packed weights differ: the form is picked per call from the number of workgroups the launch would have
    (``algo`` forces one), its weights packed on first use."""

    def __init__(self, weight: torch.Tensor, ksize: int = 3, stride: int = 1, pad: int = 1, dilation: int = 1, transposed: bool = False,
                 algo: Optional[int] = None, w_mul_dev: Optional[torch.Tensor] = None):
        """``w_mul_dev`` (r6, the training step): a one-element device float, the power of two ``split_scale_of(weight)`` -- the weights
        are scaled by it ON THE DEVICE instead of by 2**w_exp from a host read of max|w| (weights change every step: no sync)."""
        _gpu(weight, "weight")
        if weight.dtype != torch.float32:
            raise RuntimeError("conv3d weights must be float32")
        self.transposed = bool(transposed)
        if transposed:
            self.cin, self.cout = weight.shape[0], weight.shape[1]
        else:
            self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.ksize, self.stride, self.pad, self.dilation = int(ksize), int(stride), int(pad), int(dilation)
        self.forced_algo = algo         # None: chosen per call
        self.algo = int(algo or 0)
        self.w_mul_dev = w_mul_dev
        self._sc_w = self._bi_0 = None
        if w_mul_dev is not None:
            self.weight = (weight.detach() * w_mul_dev).contiguous()
            self.w_exp = 0
        else:
            self.weight = weight.detach().contiguous()
            wmax = float(self.weight.abs().max().item()) if weight.numel() else 1.0
            self.w_exp = 14 - math.frexp(wmax)[1] if wmax > 0 and math.isfinite(wmax) else 0      # wmax * 2^w_exp in [2^13, 2^14)
        self._packed = {}
        # device-scaled weights (the training step: rebuilt every time the weights move) are packed on first use, in the form that use
        # picks -- not in a default form nobody may ask for
        self.packed = self._pack(self.algo) if w_mul_dev is None else None
        self._affine = {}

    def _pack(self, algo: int):
        hit = self._packed.get(algo)
        if hit is None:
            self.algo = algo
            probe = self._desc(1, (16, 16, 32), 0)
            nbytes = _lib.lib().snvc_f16x3_conv3d_packed_weight_bytes(ctypes.byref(probe))
            if nbytes < 0:
                check(2, "snvc_f16x3_conv3d_packed_weight_bytes")
            hit = torch.empty(nbytes, dtype=torch.uint8, device=self.weight.device)
            with torch.cuda.device(self.weight.device):
                check(_lib.lib().snvc_f16x3_conv3d_pack_weights(ctypes.byref(probe), _ptr(self.weight), _ptr(hit), float(2.0 ** self.w_exp),
                                                                _stream(self.weight)), "snvc_f16x3_conv3d_pack_weights")
            self._packed[algo] = hit
        return hit

    def _pick_form(self, n: int, out_sp, plain: bool = False, split_out: bool = False):
        """Kernel form of a stride-1 layer for this launch: enough workgroups to cover the 256 CUs about four times.
        ``plain``: no residual and a split output -- what the 16x16x32 form (r4) covers."""
        if self.forced_algo is None and self.stride == 1 and not self.transposed and self.ksize in (5, 7) and self.cout % 32 == 0:
            # 7^3 / 5^3 / dilated 5^3: the 16x16x32 form (four taps per MFMA, planes serial) when the output is a split tensor and the
            # launch fills the chip (released conv1 2.47 -> 2.10 ms/crop, conv3 0.49 -> 0.42; the plain 5^3 layer with its quads over
            # all 125 taps 0.53 -> 0.48 -- slice by slice, 28 tap slots for 25 taps gained nothing over the 26 of 13 tap pairs)
            tiles5 = n * -(-out_sp[0] // 4) * -(-out_sp[1] // 4) * -(-out_sp[2] // 32) * (self.cout // 32)
            q16 = X3_Q16[0] and split_out and tiles5 >= 512 and (self.ksize == 7 or self.dilation == 2 or X3_Q16_K5[0])
            return _lib.ALGO_X3_Q16 if q16 else 0
        if self.forced_algo is None and self.ksize == 3 and self.cout % 64 == 0 and (self.stride == 2 or self.transposed):
            # half-height tiles when the launch would not fill the chip's 512 workgroup slots a few times over (r5; the hourglass's
            # quarter-resolution level): stride 2 counts 2x4x32 output tiles, a transposed layer 8 classes of 4x4x32 input tiles
            if self.transposed:
                wgs = 8 * n * -(-out_sp[0] // 8) * -(-out_sp[1] // 8) * -(-out_sp[2] // 64)
                return _lib.ALGO_X3_SMALL if wgs < X3_SMALL_BELOW["transposed"] else 0
            wgs = n * -(-out_sp[0] // 2) * -(-out_sp[1] // 4) * -(-out_sp[2] // 32)
            if X3_Q16_S2[0] and plain and split_out:
                return _lib.ALGO_X3_Q16      # r5: both planes, three image slots, one workgroup per CU (conv3d_x3s2q_kernel)
            return _lib.ALGO_X3_SMALL if wgs < X3_SMALL_BELOW["stride2"] else 0
        if self.forced_algo is not None or self.stride != 1 or self.transposed or self.ksize != 3 or self.cout == 1:
            return self.algo
        tiles = n * -(-out_sp[0] // 4) * -(-out_sp[1] // 4) * -(-out_sp[2] // 32)
        if plain and X3_Q16[0] and self.cout % 32 == 0 and tiles * (self.cout // 32) >= X3_Q16_MIN_JOBS[0]:
            return _lib.ALGO_X3_Q16         # v_mfma_f32_16x16x32_f16: ~15 % faster under the chip's power limit (conv2 0.94 -> 0.81 ms)
        if self.cout % 64 == 0 and tiles * (self.cout // 64) >= 1024:
            return _lib.ALGO_X3_SERIAL      # 64-channel blocks: the serial-plane form measures 6 % faster (0.424 vs 0.453 ms, hg conv2)
        if tiles * (self.cout // 32) >= 1024 or self.cout == 32 and tiles >= 512:
            return _lib.ALGO_X3_NARROW if self.cout != 32 else 0
        return _lib.ALGO_X3_SMALL

    out_spatial = Conv3dLayer.out_spatial

    def _desc(self, *a, **k):
        d = Conv3dLayer._desc(self, *a, **k)
        d.algo = self.algo
        return d

    def folded(self, scale, bias, x_exp: int, out_exp: int):
        """(scale', bias') of the epilogue with the exponents folded in: conv sums are in units of 2^(x_exp + w_exp), the
        stored result in units of 2^out_exp.  Cached per (scale, bias, exponents)."""
        key = (None if scale is None else (scale.data_ptr(), scale._version), None if bias is None else (bias.data_ptr(), bias._version),
               x_exp, out_exp)
        hit = self._affine.get(key)
        if hit is not None and (hit[2] is not scale or hit[3] is not bias):
            hit = None      # another tensor at a recycled address (the entry keeps its sources alive, so this only follows a clear())
        if hit is None:
            dev = self.weight.device
            sc = (scale.detach().float() if scale is not None else torch.ones(self.cout, device=dev)) * (2.0 ** (out_exp - x_exp - self.w_exp))
            bi = (bias.detach().float() if bias is not None else torch.zeros(self.cout, device=dev)) * (2.0 ** out_exp)
            if len(self._affine) > 8:
                self._affine.clear()
            hit = self._affine[key] = (sc.contiguous(), bi.contiguous(), scale, bias)
        return hit[:2]

    def __call__(self, x, x_exp: int = 0, scale=None, bias=None, residual=None, flags: int = 0, out=None, out_exp: int = 0,
                 out_f32=None, to_f32: bool = False, head=None, overflow=None, x_mul_dev=None, res_exp: Optional[int] = None,
                 residual_f32: Optional[torch.Tensor] = None):
        """y = epilogue(conv(x)).  x: split C8 tensor holding values * 2**x_exp.  Result: a split C8 tensor holding
        y * 2**out_exp, or -- ``to_f32`` / ``out_f32`` -- float32 NCDHW.  ``residual``: a split tensor holding values *
        2**res_exp (default: out_exp).  ``head`` [Cout = 32] weights: returns ``(y, y_head)`` with ``y_head`` the
        float32 [N,1,D,H,W] projection sum_c head[c] * y[:, c] written by the same launch.  ``overflow``: an int32 device
        tensor that is set to 1 if a value had to be clamped to half's range.  ``residual_f32`` (r6, float32 result only): a float32
        [N,Cout,D,H,W] tensor added to the stored result (EPI_ADD_POST implied)."""
        _split_check(x, "x")
        if x.size(2) * 8 != self.cin:
            raise RuntimeError(f"conv3d input must have {self.cin} channels (split C8), got {x.size(2) * 8}")
        n = x.size(0)
        in_sp = tuple(x.shape[3:6])
        out_sp = self.out_spatial(in_sp)
        f32 = to_f32 or out_f32 is not None or self.cout == 1        # a one-channel layer (the occupancy head) writes an fp32 plane
        if f32:
            if out_f32 is None:
                out_f32 = torch.empty((n, self.cout) + out_sp, dtype=torch.float32, device=x.device)
            elif tuple(out_f32.shape) != (n, self.cout) + out_sp or out_f32.dtype != torch.float32 or not _dense_inner(out_f32):
                raise RuntimeError("out_f32 must be a float32 [N,Cout,D,H,W] tensor, dense below dim 0")
        elif out is None:
            out = torch.empty((n, 2, self.cout // 8) + out_sp + (8,), dtype=torch.float16, device=x.device)
        else:
            _split_check(out, "out")
            if tuple(out.shape) != (n, 2, self.cout // 8) + out_sp + (8,):
                raise RuntimeError("conv3d `out` must be a split C8 tensor of the output shape")
        if residual is not None:
            _split_check(residual, "residual")
            if tuple(residual.shape) != (n, 2, self.cout // 8) + out_sp + (8,):
                raise RuntimeError("residual must have the output's shape (split C8)")
        if residual_f32 is not None:
            if not f32 or residual is not None or flags & (EPI_ADD_PRE | EPI_ADD_POST) or self.cout == 1:
                raise RuntimeError("residual_f32 goes with a float32 result and no other residual")
            if (residual_f32.dtype != torch.float32 or tuple(residual_f32.shape) != (n, self.cout) + out_sp or not _dense_inner(residual_f32)
                    or _batch_stride(residual_f32) != _batch_stride(out_f32)):
                raise RuntimeError("residual_f32 must be a float32 tensor of the result's shape and layout")
            flags = flags | EPI_ADD_POST
        y_head = None
        if head is not None:
            head = head.detach().reshape(-1).float().contiguous()
            y_head = torch.empty((n, 1) + out_sp, dtype=torch.float32, device=x.device)
        # with a float32 result the epilogue works in units of 2^out_exp too (the residual's) and scales back on the way out: exact
        if self.w_mul_dev is not None and scale is None and bias is None and x_exp == 0 and out_exp == 0:
            # the training step's plain convolution: 1 / w_mul per channel is formed once per weight version, one small launch per call
            if self._sc_w is None:
                self._sc_w, self._bi_0 = _ones_zeros(self.cout, x.device)[0] / self.w_mul_dev, _ones_zeros(self.cout, x.device)[1]
            sc, bi = (self._sc_w if x_mul_dev is None else (self._sc_w / x_mul_dev)), self._bi_0
        else:
            sc, bi = self.folded(scale, bias, x_exp, out_exp)
            if x_mul_dev is not None:           # x holds values * x_mul_dev (a device-side power of two, see split_scale_for); x_exp is 0
                sc = (sc / x_mul_dev).contiguous()
            if self.w_mul_dev is not None:
                sc = (sc / self.w_mul_dev).contiguous()
        # r6: the training step's layers (device-scaled weights) take the 16x16x32 stride-1 form with a float32 result too
        self.algo = self._pick_form(n, out_sp, plain=residual is None and (not f32 or (self.w_mul_dev is not None and head is None)),
                                    split_out=not f32 and head is None)
        packed = self._pack(self.algo)
        if n == 0:
            return out_f32 if f32 else ((out, y_head) if head is not None else out)
        d = self._desc(n, in_sp, flags, _batch_stride(x), _batch_stride(out_f32 if f32 else out),
                       _batch_stride(residual) if residual is not None else 0)
        null = ctypes.c_void_p(0)
        with torch.cuda.device(x.device):
            check(_lib.lib().snvc_f16x3_conv3d_forward(ctypes.byref(d), _ptr(x), _lo_ptr(x), _ptr(packed), _ptr(sc), _ptr(bi),
                                                       _ptr(residual if residual_f32 is None else residual_f32),
                                                       _lo_ptr(residual) if residual is not None else null,
                                                       null if f32 else _ptr(out), null if f32 else _lo_ptr(out),
                                                       _ptr(out_f32) if f32 else null, _ptr(head), _ptr(y_head), float(2.0 ** -out_exp),
                                                       float(2.0 ** (out_exp - (out_exp if res_exp is None else res_exp))),
                                                       _ptr(overflow), _stream(x)), "snvc_f16x3_conv3d_forward")
        if f32:
            return out_f32          # the kernel multiplied by 2^-out_exp on the way out
        return (out, y_head) if head is not None else out

    def forward_f32(self, x, x_mul_dev, residual_f32: Optional[torch.Tensor] = None, relu: bool = False):
        """r6 (training): the plain convolution of a device-scaled layer with a float32 NCDHW result (snvc_f16x3_conv3d_forward_f32):
        ``x`` holds values * x_mul_dev (one device float, taken out by the kernel: no launch to fold it into the scale vector),
        ``residual_f32`` is added to the stored result (a skip connection's gradient)."""
        _split_check(x, "x")
        if self.w_mul_dev is None or x.size(2) * 8 != self.cin:
            raise RuntimeError("forward_f32: a device-scaled layer (w_mul_dev) and an input of its channel count")
        n = x.size(0)
        in_sp = tuple(x.shape[3:6])
        out_sp = self.out_spatial(in_sp)
        if self._sc_w is None:
            self._sc_w, self._bi_0 = _ones_zeros(self.cout, x.device)[0] / self.w_mul_dev, _ones_zeros(self.cout, x.device)[1]
        y = torch.empty((n, self.cout) + out_sp, dtype=torch.float32, device=x.device)
        if residual_f32 is not None and (residual_f32.dtype != torch.float32 or tuple(residual_f32.shape) != tuple(y.shape)
                                         or not _dense_inner(residual_f32) or _batch_stride(residual_f32) != _batch_stride(y)):
            raise RuntimeError("residual_f32 must be a float32 tensor of the result's shape and layout")
        self.algo = self._pick_form(n, out_sp, plain=True, split_out=False)
        packed = self._pack(self.algo)
        if n == 0:
            return y
        d = self._desc(n, in_sp, EPI_RELU if relu else 0, _batch_stride(x), _batch_stride(y), 0)
        with torch.cuda.device(x.device):
            check(_lib.lib().snvc_f16x3_conv3d_forward_f32(ctypes.byref(d), _ptr(x), _lo_ptr(x), _ptr(packed), _ptr(self._sc_w), _ptr(self._bi_0),
                                                           _ptr(x_mul_dev), _ptr(residual_f32), _ptr(y), 1.0, _stream(x)),
                  "snvc_f16x3_conv3d_forward_f32")
        return y

    def forward_stats(self, x, x_mul_dev, gamma, beta, eps: float):
        """r6 (training): the plain convolution with a float32 result AND the batch statistics of that result from the same launch
        (snvc_f16x3_conv3d_forward_stats): (raw, scale, shift, mean, var) as Conv3dLayer.forward_stats, or None when the layer's
        kernel form carries no statistics epilogue.  Device-scaled weights only (``w_mul_dev``)."""
        _split_check(x, "x")
        if self.w_mul_dev is None or x.size(2) * 8 != self.cin or self.cout % 32:
            return None
        n = x.size(0)
        in_sp = tuple(x.shape[3:6])
        out_sp = self.out_spatial(in_sp)
        if self._sc_w is None:
            self._sc_w, self._bi_0 = _ones_zeros(self.cout, x.device)[0] / self.w_mul_dev, _ones_zeros(self.cout, x.device)[1]
        sc = self._sc_w                        # 1 / x_mul is applied by the kernel (x_mul: a device pointer)
        self.algo = self._pick_form(n, out_sp, plain=True, split_out=False)
        packed = self._pack(self.algo)
        raw = torch.empty((n, self.cout) + out_sp, dtype=torch.float32, device=x.device)
        d = self._desc(n, in_sp, 0, _batch_stride(x), _batch_stride(raw), 0)
        nbytes = _lib.lib().snvc_f16x3_conv3d_stats_workspace_bytes(ctypes.byref(d))
        if nbytes < 0:
            return None
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        out = torch.empty((4, self.cout), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.lib().snvc_f16x3_conv3d_forward_stats(ctypes.byref(d), _ptr(x), _lo_ptr(x), _ptr(packed), _ptr(sc), _ptr(self._bi_0),
                                                            _ptr(x_mul_dev), _ptr(raw), 1.0, _ptr(gamma), _ptr(beta), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                                            _ptr(out[3]), _ptr(ws), float(eps), _stream(x))
        if rc == 2:            # SNVC_ERR_UNSUPPORTED: the caller runs the convolution and the statistics pass separately
            return None
        check(rc, "snvc_f16x3_conv3d_forward_stats")
        return raw, out[0:1], out[1:2], out[2:3], out[3:4]

    def forward_tail(self, x, x_exp: int, scale, bias, tail: "TailWeightsX3", residual=None, flags: int = 0, out_exp: int = 0,
                     overflow=None, res_exp: Optional[int] = None, out=None):
        """A transposed layer whose result y = epilogue(conv(x)) (held in units 2**out_exp, clamped to half's range and flagged like any
        split output) feeds only a one-channel transposed layer with folded weights ``tail``: returns the per-voxel tap contractions
        T [N, 27, 8, Din, Hin, Win] float32 (class-major over y's parity classes) -- y itself is never stored
        (snvc_f16x3_deconv3d_tail_forward; ``deconv_tail_gather`` finishes the layer)."""
        _split_check(x, "x")
        if not self.transposed or self.stride != 2 or x.size(2) * 8 != self.cin or tail.cin != self.cout:
            raise RuntimeError("forward_tail: a ConvTranspose3d(k3,s2,p1,op1) layer whose Cout equals the tail's Cin")
        n = x.size(0)
        in_sp = tuple(x.shape[3:6])
        out_sp = self.out_spatial(in_sp)
        shape = (n, 27, 8) + in_sp
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise RuntimeError("forward_tail `out` must be a contiguous float32 [N, 27, 8, Din, Hin, Win] tensor")
        if residual is not None:
            _split_check(residual, "residual")
            if tuple(residual.shape) != (n, 2, self.cout // 8) + out_sp + (8,):
                raise RuntimeError("residual must have the layer's output shape (split C8)")
        sc, bi = self.folded(scale, bias, x_exp, out_exp)
        self.algo = self._pick_form(n, out_sp)
        packed = self._pack(self.algo)
        if n == 0:
            return out
        d = self._desc(n, in_sp, flags, _batch_stride(x), 0, _batch_stride(residual) if residual is not None else 0)
        null = ctypes.c_void_p(0)
        with torch.cuda.device(x.device):
            check(_lib.lib().snvc_f16x3_deconv3d_tail_forward(
                ctypes.byref(d), _ptr(x), _lo_ptr(x), _ptr(packed), _ptr(sc), _ptr(bi), _ptr(residual),
                _lo_ptr(residual) if residual is not None else null, float(2.0 ** (out_exp - (out_exp if res_exp is None else res_exp))),
                _ptr(tail.packed), _ptr(out), float(2.0 ** -(out_exp + tail.w_exp)), _ptr(overflow), _stream(x)),
                "snvc_f16x3_deconv3d_tail_forward")
        return out


class TailWeightsX3:
    """The folded one-channel transposed layer behind a split transposed layer (``Conv3dLayerX3.forward_tail``): W' [Cin, 27]
    (tap = (kd*3 + kh)*3 + kw) packed as split MFMA A fragments, values * 2**w_exp with max|W'| * 2**w_exp in [2^13, 2^14)."""

    def __init__(self, weight: torch.Tensor):
        _gpu(weight, "weight")
        w = weight.detach().reshape(weight.shape[0], -1).float().contiguous()
        if w.shape[1] != 27 or w.shape[0] % 32 != 0:
            raise RuntimeError("tail weights must be [Cin, 27] (or [Cin, 1, 3, 3, 3]) with Cin % 32 == 0")
        self.cin = int(w.shape[0])
        wmax = float(w.abs().max().item()) if w.numel() else 1.0
        self.w_exp = 14 - math.frexp(wmax)[1] if wmax > 0 and math.isfinite(wmax) else 0
        nbytes = _lib.lib().snvc_f16x3_tail_packed_weight_bytes(self.cin)
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        with torch.cuda.device(w.device):
            check(_lib.lib().snvc_f16x3_tail_pack_weights(_ptr(w), self.cin, _ptr(self.packed), float(2.0 ** self.w_exp), _stream(w)),
                  "snvc_f16x3_tail_pack_weights")


def deconv_tail_gather(t: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, out=None):
    """The scatter half of a transposed layer (k3, s2, p1, op1) to one channel: ``t`` [N, 27, 8, nd, nh, nw] (per-voxel tap
    contractions, class-major: ``Conv3dLayerX3.forward_tail``) -> float32 [N, 1, 4nd, 4nh, 4nw] = bias + residual + the sums over
    (voxel, tap) pairs that land on each output (snvc_deconv_tail_gather)."""
    _gpu(t, "t")
    if t.dtype != torch.float32 or t.dim() != 6 or t.size(1) != 27 or t.size(2) != 8 or not t.is_contiguous():
        raise RuntimeError("t must be a contiguous float32 [N, 27, 8, nd, nh, nw] tensor")
    n, nd, nh, nw = t.size(0), t.size(3), t.size(4), t.size(5)
    shape = (n, 1, 4 * nd, 4 * nh, 4 * nw)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=t.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise RuntimeError("out must be a contiguous float32 tensor of the output shape")
    if residual is not None:
        _gpu(residual, "residual")
        if residual.dtype != torch.float32 or residual.numel() != out.numel() or not residual.is_contiguous():
            raise RuntimeError("residual must be a contiguous float32 tensor of the output shape")
    if bias is not None:
        bias = bias.detach().reshape(-1)[:1].float().contiguous()
    with torch.cuda.device(t.device):
        check(_lib.lib().snvc_deconv_tail_gather(_ptr(t), _ptr(bias), _ptr(residual), _ptr(out), n, nd, nh, nw, _stream(t)),
              "snvc_deconv_tail_gather")
    return out


def mul_broadcast_split(feat, occ, out=None):
    """out = split((hi + lo) * occ[n, 0]) on split pairs (snvc_f16x3_mul_broadcast); occ: float32 [N,1,D,H,W]."""
    _split_check(feat, "feat")
    _gpu(occ, "occ")
    occ = occ.contiguous()
    n, g = feat.shape[0], feat.shape[2]
    s_ = math.prod(feat.shape[3:6])
    if occ.dtype != torch.float32 or occ.numel() != n * s_:
        raise RuntimeError("occupancy must be float32 [N,1,D,H,W] matching the feature volume")
    if out is None:
        out = torch.empty(feat.shape, dtype=torch.float16, device=feat.device)
    else:
        _split_check(out, "out")
        if tuple(out.shape) != tuple(feat.shape):
            raise RuntimeError("mul_broadcast_split `out` must have feat's shape")
    if feat.numel() == 0:
        return out
    with torch.cuda.device(feat.device):
        check(_lib.lib().snvc_f16x3_mul_broadcast(_ptr(feat), _lo_ptr(feat), _ptr(occ), _ptr(out), _lo_ptr(out), n, 8 * g, s_,
                                                  _batch_stride(feat), _batch_stride(out), _stream(feat)), "snvc_f16x3_mul_broadcast")
    return out


def mul_broadcast_c8(feat, occ, out=None):
    """out[n,c,...] = feat[n,c,...] * occ[n,0,...] on C8 tensors (occ: float32 [N,1,D,H,W])."""
    _c8_check(feat, "feat")
    _gpu(occ, "occ")
    occ = occ.contiguous()
    n, g = feat.shape[0], feat.shape[1]
    s = math.prod(feat.shape[2:5])
    if occ.dtype != torch.float32 or occ.numel() != n * s:
        raise RuntimeError("occupancy must be float32 [N,1,D,H,W] matching the feature volume")
    if out is None:
        out = torch.empty_like(feat)
    else:
        _c8_check(out, "out")
        if tuple(out.shape) != tuple(feat.shape):
            raise RuntimeError("mul_broadcast_c8 `out` must have feat's shape")
    if feat.numel() == 0:
        return out
    with torch.cuda.device(feat.device):
        check(_lib.lib().snvc_f16_mul_broadcast(_ptr(feat), _ptr(occ), _ptr(out), n, 8 * g, s, _batch_stride(feat),
                                                _batch_stride(out), _stream(feat)), "snvc_f16_mul_broadcast")
    return out


def avgpool_depth4_c8(x):
    """AvgPool3d((4,1,1),(4,1,1)) (vernier.py:289) of a C8 tensor -> float32 [N,C,D//4,H,W]."""
    _c8_check(x, "x")
    n, g, d, h, w, _ = x.shape
    y = torch.empty((n, 8 * g, d // 4, h, w), dtype=torch.float32, device=x.device)
    if y.numel() == 0:
        return y
    with torch.cuda.device(x.device):
        check(_lib.lib().snvc_f16_avgpool_depth4(_ptr(x), _ptr(y), n, 8 * g, d, h * w, _batch_stride(x), _stream(x)),
              "snvc_f16_avgpool_depth4")
    return y


# ------------------------------------------------------------------------------ roiaware_pool3d
def roiaware_pool3d_forward(rois, pts, pts_feature, argmax, pts_idx_of_voxels, pooled_features, pool_method: int):
    """roiaware_pool3d_cuda.forward (roiaware_pool3d.cpp:29-66): outputs are pre-zeroed by the caller."""
    for t, nm in ((rois, "rois"), (pts, "pts"), (pts_feature, "pts_feature"), (argmax, "argmax"),
                  (pts_idx_of_voxels, "pts_idx_of_voxels"), (pooled_features, "pooled_features")):
        _gpu(t, nm)
        if not t.is_contiguous():
            raise RuntimeError(f"{nm} must be contiguous")
    b, p, c = rois.size(0), pts.size(0), pts_feature.size(1)
    _, ox, oy, oz, max_pts = pts_idx_of_voxels.shape
    if not (ox < 256 and oy < 256 and oz < 256):
        raise AssertionError("we encode index with 8bit")  # roiaware_pool3d.cpp:53
    ws = torch.empty((max(b * p, 1),), dtype=torch.int32, device=rois.device)
    with torch.cuda.device(rois.device):
        check(_lib.lib().snvc_roiaware_pool3d_forward(_ptr(rois), _ptr(pts), _ptr(pts_feature), _ptr(argmax),
                                                      _ptr(pts_idx_of_voxels), _ptr(pooled_features), _ptr(ws), b, p,
                                                      c, max_pts, ox, oy, oz, int(pool_method), _stream(rois)),
              "roiaware_pool3d forward")
    return 1


def roiaware_pool3d_backward(pts_idx_of_voxels, argmax, grad_out, grad_in, pool_method: int):
    for t, nm in ((pts_idx_of_voxels, "pts_idx_of_voxels"), (argmax, "argmax"), (grad_out, "grad_out"), (grad_in, "grad_in")):
        _gpu(t, nm)
        if not t.is_contiguous():
            raise RuntimeError(f"{nm} must be contiguous")
    b, ox, oy, oz, max_pts = pts_idx_of_voxels.shape
    c = grad_out.size(4)
    with torch.cuda.device(grad_out.device):
        check(_lib.lib().snvc_roiaware_pool3d_backward(_ptr(pts_idx_of_voxels), _ptr(argmax), _ptr(grad_out),
                                                       _ptr(grad_in), b, c, max_pts, ox, oy, oz, int(pool_method),
                                                       _stream(grad_out)), "roiaware_pool3d backward")
    return 1


def points_in_boxes_gpu(boxes, pts, box_idx_of_points):
    for t, nm in ((boxes, "boxes"), (pts, "pts"), (box_idx_of_points, "box_idx_of_points")):
        _gpu(t, nm)
        if not t.is_contiguous():
            raise RuntimeError(f"{nm} must be contiguous")
    bs, t_, m = boxes.size(0), boxes.size(1), pts.size(1)
    with torch.cuda.device(boxes.device):
        check(_lib.lib().snvc_points_in_boxes_gpu(_ptr(boxes), _ptr(pts), _ptr(box_idx_of_points), bs, t_, m,
                                                  _stream(boxes)), "points_in_boxes_gpu")
    return 1


def points_in_boxes_cpu(boxes, pts, pts_indices):
    """Host op in the reference too (roiaware_pool3d.cpp:137-168): CPU tensors in, CPU tensor out."""
    for t, nm in ((boxes, "boxes"), (pts, "pts"), (pts_indices, "pts_indices")):
        if t.is_cuda:
            raise RuntimeError(f"{nm} must be a CPU tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{nm} must be contiguous")
    check(_lib.lib().snvc_points_in_boxes_cpu(ctypes.c_void_p(boxes.data_ptr()), ctypes.c_void_p(pts.data_ptr()),
                                              ctypes.c_void_p(pts_indices.data_ptr()), boxes.size(0), pts.size(0)),
          "points_in_boxes_cpu")
    return 1
