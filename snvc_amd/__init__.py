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

# reference module name -> the module of this package that stands in for it
_ALIASES = {
    "snvc.extension.build_cost_volume": "snvc_amd.extension.build_cost_volume",
    "snvc.extension.roiaware_pool3d.roiaware_pool3d_utils": "snvc_amd.extension.roiaware_pool3d.roiaware_pool3d_utils",
    "snvc.models.submodule": "snvc_amd.models.submodule",
    "snvc.models.vernier": "snvc_amd.models.vernier",
}


def install_as_snvc(backbone: bool = True):
    """Make the reference's own import lines resolve to this package: after ``snvc_amd.install_as_snvc()`` (once, before
    the reference's modules are imported) ``from snvc.models.vernier import get_model``,
    ``from snvc.extension.build_cost_volume import build_cost_volume`` ... give the MI355X implementations, while every
    other ``snvc.*`` module (HRNet, dataset, utils: off the path) still comes from the reference checkout on
    ``sys.path``.  ``backbone=True`` also wires ``snvc.models.hrnet.get_model`` into ``VernierScale`` as its feature
    extractor factory when the reference package is importable (vernier.py:57-66), so that
    ``tools/inference_agnostic.py`` runs with no edit but its DataParallel line."""
    import importlib
    import sys
    for ref_name, ours in _ALIASES.items():
        mod = importlib.import_module(ours)
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition(".")
        if parent in sys.modules:                       # `import snvc.models.vernier as v` reads the attribute chain
            setattr(sys.modules[parent], leaf, mod)
    if backbone:
        try:
            hrnet = importlib.import_module("snvc.models.hrnet")
        except Exception:
            return
        vernier = sys.modules["snvc.models.vernier"]
        vernier.get_feat_extraction = lambda cfg, is_train=False, **kw: hrnet.get_model(cfg, is_train, **kw)
