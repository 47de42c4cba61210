"""ctypes binding of ``libsnvc_hip.so`` (the C ABI declared in ``include/snvc_hip.h``).

The product path has no fallback: if the shared library is missing or a call fails, a
``RuntimeError`` is raised.  Nothing here (or anywhere under ``snvc_amd``) imports ``oracle``.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNVC_HIP_LIB") or os.path.join(_HERE, "libsnvc_hip.so")   # env: development override
_lib = None
_ABI = 6   # snvc_abi_version() this binding was written against

c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_p = ctypes.c_void_p
c_int = ctypes.c_int


class Conv3dDesc(ctypes.Structure):
    """Mirror of ``snvc_conv3d_desc`` (include/snvc_hip.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "N", "Cin", "Din", "Hin", "Win", "Cout", "Dout", "Hout", "Wout",
        "ksize", "stride", "dilation", "pad", "transposed", "flags", "algo", "ksize_d", "ksize_h")] + [
        (n, ctypes.c_int64) for n in ("x_batch_stride", "y_batch_stride", "res_batch_stride")]


EPI_RELU, EPI_ADD_PRE, EPI_ADD_POST, EPI_SIGMOID, EPI_AVGPOOL_D4, EPI_STREAM_OUT = 1, 2, 4, 8, 16, 32
# snvc_conv3d_desc.algo (include/snvc_hip.h): arithmetic in the low byte, kernel-form selectors above it
ALGO_AUTO, ALGO_DIRECT = 0, 1
ALGO_WINO_TILE_BIG, ALGO_WINO_TILE_STD, ALGO_WINO_TILE_NARROW_REG = 0x100, 0x200, 0x300
ALGO_GENERIC_EPILOGUE, ALGO_SCALAR_STAGING = 0x400, 0x800
ALGO_X3_SERIAL, ALGO_X3_NARROW, ALGO_X3_SMALL, ALGO_X3_Q16 = 0x1000, 0x2000, 0x4000, 0x8000
ALGO_WGRAD_FP32 = 0x10000      # snvc_conv3d_wgrad: the fp32-MFMA forms instead of the split-operand (f16x3) one
F32, F64 = 0, 1

# name -> (restype, argtypes); kept next to the header so the symbol test can walk it
SIGNATURES = {
    "snvc_last_error_string": (ctypes.c_char_p, []),
    "snvc_abi_version": (c_int, []),
    "snvc_cost_volume_forward": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_p]),
    "snvc_cost_volume_forward_right": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_cost_volume_backward": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_p]),
    "snvc_cost_volume_backward_right": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_depth_class_sums": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_p]),
    "snvc_voxel_gather_forward": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_p]),
    "snvc_voxel_gather_workspace_floats": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "snvc_voxel_gather_forward_ws": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_p]),
    "snvc_voxel_gather_backward_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64, c_i64, c_i64]),
    "snvc_voxel_gather_backward_det": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_p]),
    "snvc_voxel_atten_scale": (c_int, [c_p, c_i64, c_i64, c_i64, c_p]),
    "snvc_voxel_gather_backward": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_p]),
    "snvc_grid_projection": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_p, c_p, c_p, c_int, c_p]),
    "snvc_conv3d_packed_weight_count": (c_i64, [ctypes.POINTER(Conv3dDesc)]),
    "snvc_conv3d_pack_weights": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p]),
    "snvc_conv3d_forward": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "snvc_conv3d_forward_ex": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "snvc_conv3d_forward_head": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "snvc_conv3d_forward_side_head": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "snvc_conv3d_stats_workspace_bytes": (c_i64, [c_p]),
    "snvc_conv3d_forward_stats": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_float, c_p]),
    "snvc_warped_expand": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_p]),
    "snvc_warped_expand_backward": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_warped_expand_split": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_p]),
    "snvc_shift_structure": (c_int, [c_p, c_p, c_i64, c_i64, c_p]),
    "snvc_sheared_upsample": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_int, c_p]),
    "snvc_sheared_stats_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "snvc_sheared_expand_stats": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int,
                                          c_i64, c_int, c_i64, c_int, ctypes.c_float, c_p]),
    "snvc_sheared_backward_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "snvc_sheared_backward_reduce": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64,
                                             c_int, c_int, c_i64, c_int, c_i64, c_int, c_p]),
    "snvc_sheared_reduce": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_i64, c_int, c_i64, c_int, c_p]),
    "snvc_sheared_wgrad_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "snvc_sheared_wgrad": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_sheared_upsample_backward": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_int, c_p]),
    "snvc_sheared_expand": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_i64, c_int, c_i64, c_int,
                                    c_int, c_p]),
    "snvc_sheared_expand_amax": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_i64, c_int, c_i64, c_int,
                                    c_int, c_p, c_p]),
    "snvc_sheared_expand_split": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_i64, c_int,
                                          c_i64, c_int, c_i64, c_int, c_p]),
    "snvc_norm_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "snvc_norm_stats": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_f32, c_p]),
    "snvc_affine_act": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p]),
    "snvc_affine_act_amax": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p, c_p]),
    "snvc_conv3d_wgrad_workspace_bytes": (c_i64, [ctypes.POINTER(Conv3dDesc)]),
    "snvc_conv3d_wgrad": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p]),
    "snvc_conv3d_wgrad_amax": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "snvc_act_backward_workspace_bytes": (c_i64, [c_i64, c_i64]),
    "snvc_act_backward_reduce": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p]),
    "snvc_act_backward_reduce_amax": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p, c_p]),
    "snvc_affine_act_twin": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p, c_p]),
    "snvc_act_backward_apply_twin": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                             c_i64, c_int, c_int, c_p, c_p]),
    "snvc_bn_track": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_f32, c_f32, c_p]),
    "snvc_split_scale_bound": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p]),
    "snvc_bn_backward_coefs": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, ctypes.c_double, ctypes.c_double, c_p]),
    "snvc_act_backward_apply": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p]),
    "snvc_act_backward_apply_amax": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p, c_p]),
    "snvc_mul_broadcast": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_avgpool_depth4": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_avgpool_depth4_backward": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_zero_stuff2x": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_p]),
    "snvc_disparity_regression": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p]),
    "snvc_argmax_rows": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_p]),
    "snvc_f16_from_ncdhw": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_f16_to_ncdhw": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_voxel_gather_forward_split": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_p]),
    "snvc_voxel_gather_forward_f16": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_p]),
    "snvc_f16_conv3d_packed_weight_bytes": (c_i64, [ctypes.POINTER(Conv3dDesc)]),
    "snvc_f16_conv3d_pack_weights": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p]),
    "snvc_f16_conv3d_forward": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "snvc_f16x3_from_ncdhw": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_p, c_p]),
    "snvc_f16x3_affine_from_ncdhw": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int,
                                             c_f32, c_f32, c_p]),
    "snvc_f16x3_mul_broadcast": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_f16x3_to_ncdhw": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_p]),
    "snvc_f16x3_conv3d_packed_weight_bytes": (c_i64, [ctypes.POINTER(Conv3dDesc)]),
    "snvc_f16x3_conv3d_pack_weights": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_f32, c_p]),
    "snvc_f16x3_conv3d_forward": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f32, c_f32, c_p, c_p]),
    "snvc_f16x3_conv3d_stats_workspace_bytes": (c_i64, [ctypes.POINTER(Conv3dDesc)]),
    "snvc_f16x3_conv3d_forward_stats": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f32, c_p]),
    "snvc_f16x3_conv3d_forward_f32": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f32, c_p]),
    "snvc_f16x3_split_scale": (c_int, [c_p, c_i64, c_p, c_p, c_p]),
    "snvc_sheared_upsample_split": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_int, c_p]),
    "snvc_sheared_prep_x3": (c_int, [c_p, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_int, c_i64, c_int, c_p, c_p, c_i64, c_f32, c_f32, c_p, c_p, c_p,
                                     c_p, c_p, c_p, c_p]),
    "snvc_f16x3_conv2d_from_f32": (c_int, [c_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "snvc_f16x3_conv2d_packed_weight_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "snvc_f16x3_conv2d_pack_weights": (c_int, [c_p, c_int, c_int, c_int, c_int, c_p, c_f32, c_p]),
    "snvc_f16x3_conv2d_forward": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_f32, c_p, c_int, c_p]),
    "snvc_f16x3_tail_packed_weight_bytes": (c_i64, [c_int]),
    "snvc_f16x3_tail_pack_weights": (c_int, [c_p, c_int, c_p, c_f32, c_p]),
    "snvc_f16x3_deconv3d_tail_forward": (c_int, [ctypes.POINTER(Conv3dDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f32, c_p, c_p, c_f32, c_p, c_p]),
    "snvc_deconv_tail_gather": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_f16_mul_broadcast": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_f16_avgpool_depth4": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p]),
    "snvc_volume_resample": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_i64, c_p]),
    "snvc_rect_to_psv_grid": (c_int, [c_p, c_p, c_p, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_p]),
    "snvc_roiaware_pool3d_forward": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p] + [c_int] * 8 + [c_p]),
    "snvc_roiaware_pool3d_backward": (c_int, [c_p, c_p, c_p, c_p] + [c_int] * 7 + [c_p]),
    "snvc_points_in_boxes_gpu": (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_p]),
    "snvc_points_in_boxes_cpu": (c_int, [c_p, c_p, c_p, c_int, c_int]),
    "snvc_kitti_eval": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_int, c_p, c_p, c_int]),
}


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into snvc_amd/libsnvc_hip.so (hipcc, no GPU needed)."""
    cmd = ["make", "-s", "-j8", "-C", os.path.join(_HERE, "csrc")]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C snvc_amd/csrc`). "
                "snvc_amd has no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        if handle.snvc_abi_version() != _ABI:
            raise RuntimeError("libsnvc_hip.so ABI version mismatch; rebuild it")
        _lib = handle
    return _lib


class Unsupported(RuntimeError):
    """SNVC_ERR_UNSUPPORTED (status 2): the arguments are valid but this kernel form does not cover the shape (row too wide,
    rows that do not fit the LDS, ...).  Callers that have a more general path catch it and take that path."""


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().snvc_last_error_string().decode("utf-8", "replace")
        raise (Unsupported if rc == 2 else RuntimeError)(f"{what or 'snvc_hip'} failed (status {rc}): {msg}")
