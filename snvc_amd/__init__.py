"""snvc_amd -- MI355X (gfx950) implementation of SNVC's cost-volume / voxel-resampling /
3D-CNN hot path behind the reference's own operator API.

    snvc_amd.extension.build_cost_volume   <->  snvc.extension.build_cost_volume
    snvc_amd.extension.roiaware_pool3d     <->  snvc.extension.roiaware_pool3d
    snvc_amd.models.submodule              <->  snvc.models.submodule (3D blocks)
    snvc_amd.models.vernier                <->  snvc.models.vernier   (VernierScale 3D trunk)

Everything executes in hand-written HIP kernels from ``libsnvc_hip.so`` (C ABI:
``include/snvc_hip.h``).  There is no CPU path: CPU tensors raise, and a missing shared
library raises at first use.
"""
__version__ = "0.1.0"
