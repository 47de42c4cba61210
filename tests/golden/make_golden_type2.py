"""Golden vectors of VernierScale with vernier_type='BEV_type2' by IMPORTING THE REFERENCE (this container only):
    python tests/golden/make_golden_type2.py          (needs /root/reference; CPU only)
Same procedure as make_golden.py: the reference module and the torch restatement (oracle.torch_ref.VernierTrunk(vernier_type=
'BEV_type2')) must have identical state-dict keys, get the same seeded parameters, must agree bit for bit on the same seeded input;
the REFERENCE's outputs are stored (tests/golden/vernier_type2.npz).  Only data is committed."""
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
for _m in ("cv2", "torchvision", "torchvision.transforms", "imageio", "numba", "mayavi", "mayavi.mlab"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.path.insert(0, REF)

import snvc.models.vernier as ref_vernier  # noqa: E402

from oracle import torch_ref as T  # noqa: E402
import golden_cases as GC  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)
ref_vernier.get_feat_extraction = lambda cfg, is_train=False, **kw: torch.nn.Identity()


def cfg_of(grid, gn, dim=32):
    cfg = types.SimpleNamespace(vernier_type="BEV_type2", backbone="hrfeat", gn=gn, grid_resolution=list(grid), resolution=GC.RESOLUTION,
                                x_range=(-1.0, 1.0), z_range=(-1.0, 1.0), num_parts=9)
    cfg.hrfeat = types.SimpleNamespace(output_channel=dim, name="hrnet-w32")
    cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
    return cfg


out = {}
with torch.no_grad():
    for name, (grid, gn, n, fh, fw, seed) in GC.TYPE2_CASES.items():
        cfg = cfg_of(grid, gn)
        ref = ref_vernier.VernierScale(cfg)
        ours = T.VernierTrunk(dim=32, grid=grid, gn=gn, vernier_type="BEV_type2")
        ka = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
        kb = [(k, tuple(v.shape)) for k, v in ours.state_dict().items()]
        assert ka == kb, set(ka) ^ set(kb)
        sd = T.seeded_state_dict(ref, seed)
        ref.load_state_dict(sd, strict=True)
        ours.load_state_dict(sd, strict=True)
        ref.eval(); ours.eval()
        lf, rf, gpl, gpr = GC.trunk_inputs(n, 32, fh, fw, grid, seed + 1)
        res = ref(lf, rf, gpl.clone(), gpr.clone())
        vox = T.sample_2d_feat(lf, rf, gpl, gpr, cfg.resolution, grid)
        h2, o2, _, c2, _ = ours.predict_3d_heatmaps(vox)
        assert res["coordinates"] is None and c2 is None
        assert torch.equal(res["ncf"], h2) and torch.equal(res["occupancy"], o2), name
        out[f"{name}/ncf"] = res["ncf"].numpy()
        out[f"{name}/occupancy"] = res["occupancy"].numpy()
        print(name, tuple(res["ncf"].shape), tuple(res["occupancy"].shape), "bit-equal to the restatement")
path = os.path.join(HERE, "vernier_type2.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
