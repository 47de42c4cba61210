"""Generates tests/golden/decode_outputs.npz by IMPORTING THE REFERENCE (this container only; CPU).

Run:  python tests/golden/make_golden_decode.py

Pins SURVEY.md 8f row N4: VernierScale.ncf_to_update_2d (snvc/models/vernier.py:665-738) with the CLI's Filter,
and the KITTI label formatter get_instance_str / roty2alpha / update_record (tools/inference_agnostic.py:277-364).
Inputs are re-drawn from seeds by tests/test_decode.py::decode_case; only the reference's OUTPUTS are stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.environ.get("SNVC_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
for _m in ("cv2", "torchvision", "torchvision.transforms", "imageio", "numba", "mayavi", "mayavi.mlab", "tensorboardX"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["tensorboardX"].SummaryWriter = object
sys.path.insert(0, REF)

import snvc.models.vernier as ref_vernier  # noqa: E402
spec = importlib.util.spec_from_file_location("ref_cli", os.path.join(REF, "tools", "inference_agnostic.py"))
ref_cli = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_cli)

from test_decode import decode_case  # noqa: E402

out = {}
for name in ("argmax", "coordinates", "one_part_only"):
    c = decode_case(name)
    model = types.SimpleNamespace(cfg=c["cfg"], xrange=c["cfg"].x_range[1] - c["cfg"].x_range[0],
                                  zrange=c["cfg"].z_range[1] - c["cfg"].z_range[0])
    for fn in ("_get_basis", "get_canonical", "get_euler_2d", "register_BEV"):
        setattr(model, fn, types.MethodType(getattr(ref_vernier.VernierScale, fn), model))
    res = ref_vernier.VernierScale.ncf_to_update_2d(model, torch.from_numpy(c["ncf"]), c["samples"].copy(), c["grid"].copy(),
                                                   ref_cli.Filter(), coordinates=None if c["coordinates"] is None else c["coordinates"].copy())
    out[f"{name}/confidence"] = np.asarray(res["confidence"])
    out[f"{name}/keep_flags"] = np.asarray(res["keep_flags"])
    for k, v in res["pred"].items():
        out[f"{name}/pred_{k}"] = np.asarray(v, dtype=np.float64)
    if "all_parts" in res["pred"]:
        record = {}
        ref_cli.update_record(record, res, c["meta"])
        lines = []
        for fname in sorted(record):
            lines += [fname] + record[fname]["all_parts"]
        out[f"{name}/kitti_lines"] = np.array("\n".join(lines))
out["roty2alpha"] = np.array([ref_cli.roty2alpha(x, z, r) for x, z, r in [(1.0, 10.0, 0.3), (-5.0, 20.0, -3.0), (3.0, 8.0, 3.1), (0.0, 5.0, -1.6)]])
path = os.path.join(HERE, "decode_outputs.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {len(out)} arrays")
