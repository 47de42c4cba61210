"""TEST INFRASTRUCTURE ONLY -- the reference's own C++ `points_in_boxes_cpu`, built by `make -C oracle ref` into oracle/_ref/ from
/root/reference/snvc/extension/roiaware_pool3d/src/roiaware_pool3d.cpp (:121-168) where it lies.  Only tests/, bench.py's CPU leg and
tests/golden/make_golden_ref_native.py may import this; nothing under snvc_amd/ does.

The module's translation unit also declares the three CUDA launchers of the .cu file (roiaware_pool3d.cpp:19-27); they are undefined
symbols of the .so, so it is loaded with lazy binding and only the CPU function is ever called."""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "_ref", "snvc_ref_roiaware.so")
_MOD = [None]


def available() -> bool:
    return os.path.isfile(SO_PATH)


def module():
    """The reference's pybind module (forward / backward / points_in_boxes_gpu are CUDA-only and unresolved: do not call)."""
    if _MOD[0] is None:
        if not available():
            raise FileNotFoundError(f"{SO_PATH} not built: run `make -C oracle ref` where /root/reference exists")
        import torch  # noqa: F401  (libtorch must be loaded first)
        flags = sys.getdlopenflags()
        sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
        try:
            spec = importlib.util.spec_from_file_location("snvc_ref_roiaware", SO_PATH)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        finally:
            sys.setdlopenflags(flags)
        _MOD[0] = mod
    return _MOD[0]


def points_in_boxes_cpu(points, boxes):
    """points [P,3], boxes [B,7] float32 numpy -> int32 [B,P] flags, computed by the reference's compiled function."""
    import numpy as np
    import torch
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32))
    bx = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32))
    out = torch.zeros((bx.shape[0], pts.shape[0]), dtype=torch.int32)
    module().points_in_boxes_cpu(bx, pts, out)
    return out.numpy()
