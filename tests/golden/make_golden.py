"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (this container only).

Run:  python tests/golden/make_golden.py         (needs /root/reference; CPU only)

For every case the script
  1. builds the reference module (snvc.models.submodule / snvc.models.vernier.VernierScale),
  2. builds our torch restatement (oracle.torch_ref) and checks the two have IDENTICAL
     state-dict keys and shapes,
  3. loads the same seeded parameters into both (oracle.torch_ref.seeded_state_dict),
  4. runs both on the same seeded input and requires agreement (exact where the op
     sequence is identical; the trunk's in-place adds make it bit-identical too),
  5. stores the REFERENCE's outputs as the golden vector.

Only data (seeds + expected outputs) is committed; inputs and weights are regenerated
from the seeds by tests/golden_cases.py.  No reference source or bytecode is stored.
"""
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
# modules the reference imports at module scope but never touches on the code paths used here
for _m in ("cv2", "torchvision", "torchvision.transforms", "imageio", "numba", "mayavi", "mayavi.mlab"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.path.insert(0, REF)

import snvc.models.submodule as ref_sub  # noqa: E402
import snvc.models.vernier as ref_vernier  # noqa: E402

from oracle import torch_ref as T  # noqa: E402
import golden_cases as GC  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def same_keys(a, b):
    ka = [(k, tuple(v.shape)) for k, v in a.state_dict().items()]
    kb = [(k, tuple(v.shape)) for k, v in b.state_dict().items()]
    assert ka == kb, (set(ka) ^ set(kb))


def load_both(ref, ours, seed):
    same_keys(ref, ours)
    sd = T.seeded_state_dict(ref, seed)
    ref.load_state_dict(sd, strict=True)
    ours.load_state_dict(sd, strict=True)
    ref.eval()
    ours.eval()


def check(a, b, what, exact=True):
    a, b = a.detach(), b.detach()
    if exact:
        assert torch.equal(a, b), f"{what}: max diff {(a - b).abs().max().item()}"
    else:
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), f"{what}: {(a - b).abs().max().item()}"


def vernier_cfg(grid, gn, dim=32):
    cfg = types.SimpleNamespace()
    cfg.vernier_type = "BEV_type3"
    cfg.backbone = "hrfeat"
    cfg.hrfeat = types.SimpleNamespace(output_channel=dim, name="hrnet-w32")
    cfg.gn = gn
    cfg.grid_resolution = list(grid)
    cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
    cfg.resolution = GC.RESOLUTION
    cfg.x_range = (-1.0, 1.0)
    cfg.z_range = (-1.0, 1.0)
    cfg.num_parts = 9
    return cfg


out = {}
with torch.no_grad():
    # ------------------------------------------------------------- convbn_3d cases
    for name, (cin, cout, k, s, p, dil, gn, shape, seed) in GC.CONV_CASES.items():
        ref = ref_sub.convbn_3d(cin, cout, k, s, p, dilation=dil, gn=gn)
        ours = T.convbn_3d(cin, cout, k, s, p, dilation=dil, gn=gn)
        load_both(ref, ours, seed)
        x = GC.randn((1, cin) + shape, seed + 1)
        yr, yo = ref(x), ours(x)
        check(yr, yo, name)
        out[f"conv/{name}"] = yr.numpy()

    # ------------------------------------------------------------- hourglass
    for name, (c, gn, shape, seed) in GC.HOURGLASS_CASES.items():
        ref, ours = ref_sub.hourglass(c, gn=gn), T.hourglass(c, gn=gn)
        load_both(ref, ours, seed)
        x = GC.randn((1, c) + shape, seed + 1)
        (r0, r1, r2), (o0, o1, o2) = ref(x.clone(), None, None), ours(x.clone(), None, None)
        for a, b, w in ((r0, o0, "out"), (r1, o1, "pre"), (r2, o2, "post")):
            check(a, b, f"{name}.{w}")
        out[f"hourglass/{name}/out"], out[f"hourglass/{name}/pre"], out[f"hourglass/{name}/post"] = \
            r0.numpy(), r1.numpy(), r2.numpy()
        # presqu / postsqu variant (stacked-hourglass call pattern, submodule.py:153-164)
        pres = GC.randn(tuple(r1.shape), seed + 2)
        posts = GC.randn(tuple(r2.shape), seed + 3)
        (q0, q1, q2), (p0, p1, p2) = ref(x.clone(), pres, posts), ours(x.clone(), pres, posts)
        for a, b, w in ((q0, p0, "out"), (q1, p1, "pre"), (q2, p2, "post")):
            check(a, b, f"{name}.sq.{w}")
        out[f"hourglass/{name}/sq_out"] = q0.numpy()

    for name, (c, gn, shape, seed) in GC.HOURGLASS16_CASES.items():
        ref, ours = ref_sub.hourglass_downsample_16(c, gn=gn), T.hourglass_downsample_16(c, gn=gn)
        load_both(ref, ours, seed)
        x = GC.randn((1, c) + shape, seed + 1)
        yr, yo = ref(x), ours(x)
        check(yr, yo, name)
        # 1-1.5 MB each: keep a strided subsample + fp64 checksums of the full tensor
        out[f"hourglass16/{name}_sub"] = yr[:, ::2, :, ::2, ::2].numpy()
        out[f"hourglass16/{name}_sum"] = np.array([yr.double().sum().item(), yr.double().abs().sum().item()])

    # ------------------------------------------------------------- disparityregression
    x = GC.randn((2, 12, 5, 7), 901)
    depth = torch.from_numpy(np.linspace(2.0, 40.0, 12).astype(np.float32))
    yr = ref_sub.disparityregression.forward(None, x, depth)
    check(yr, T.disparityregression(x, depth), "disparityregression")
    out["disparityregression"] = yr.numpy()

    # ------------------------------------------------------------- VernierScale trunk + gather
    ref_vernier.get_feat_extraction = lambda cfg, is_train=False, **kw: torch.nn.Identity()
    for name, (grid, gn, n, fh, fw, seed) in GC.TRUNK_CASES.items():
        cfg = vernier_cfg(grid, gn)
        ref = ref_vernier.VernierScale(cfg)
        ours = T.VernierTrunk(dim=32, grid=grid, gn=gn)
        # the reference also owns feat_net (Identity here: no parameters)
        load_both(ref, ours, seed)
        lf, rf, gpl, gpr = GC.trunk_inputs(n, 32, fh, fw, grid, seed + 1)
        vox_ref = ref.construct_voxel(lf, rf, gpl.clone(), gpr.clone())
        vox_our = T.sample_2d_feat(lf, rf, gpl, gpr, cfg.resolution, grid)
        check(vox_ref, vox_our, f"{name}.voxel")
        heat, occ, offset, coords, bbox = ref.predict_3d_heatmaps(vox_ref.clone())
        h2, o2, _, c2, _ = ours.predict_3d_heatmaps(vox_our.clone())
        check(heat, h2, f"{name}.ncf")
        check(occ, o2, f"{name}.occupancy")
        check(coords, c2, f"{name}.coordinates")
        bev, occ5, _ = ours.trunk_3d(vox_our.clone())
        # the full forward() contract (Identity backbone): dict keys + shapes
        res = ref(lf, rf, gpl.clone(), gpr.clone())
        check(res["ncf"], heat, f"{name}.forward.ncf")
        # voxel itself is N*64*nh*nw*nl floats: store a strided subsample + checksums
        out[f"trunk/{name}/voxel_sub"] = vox_ref[:, ::7, ::3, ::5, ::5].numpy()
        out[f"trunk/{name}/voxel_sum"] = np.array([vox_ref.double().sum().item(),
                                                   vox_ref.double().abs().sum().item()])
        out[f"trunk/{name}/ncf"] = heat.numpy()
        out[f"trunk/{name}/occupancy"] = occ.numpy()
        out[f"trunk/{name}/coordinates"] = coords.numpy()
        out[f"trunk/{name}/bev_sub"] = bev[:, ::5].numpy()
        out[f"trunk/{name}/bev_sum"] = np.array([bev.double().sum().item(), bev.double().abs().sum().item()])
        flat = heat.numpy().reshape(n, 9, -1)
        out[f"trunk/{name}/argmax"] = np.argmax(flat, axis=2).astype(np.int64)  # vernier.py:693

    # ------------------------------------------------------------- in-place side effect of the gather
    grid = (16, 16, 24)
    cfg = vernier_cfg(grid, False)
    ref = ref_vernier.VernierScale(cfg)
    lf, rf, gpl, gpr = GC.trunk_inputs(1, 32, 16, 16, grid, 77)
    before = gpl.clone()
    ref.construct_voxel(lf, rf, gpl, gpr)
    out["gather/inplace_normalised"] = np.array([not torch.equal(before, gpl)])
    exp = before.clone()
    exp[:, 0] = exp[:, 0] / cfg.resolution[1] * 2 - 1
    exp[:, 1] = exp[:, 1] / cfg.resolution[0] * 2 - 1
    assert torch.equal(exp, gpl), "reference normalises the caller's tensor in place (vernier.py:335-338)"

    # ------------------------------------------------------------- global stack (cfg-1 shape family)
    # there is no reference class for it (the '3D' branch is dead code upstream); build it from
    # the reference's own convbn_3d / hourglass and compare with our GlobalStack.
    for name, (c, shape, seed) in GC.GLOBAL_CASES.items():
        ours = T.GlobalStack(c)
        ref_parts = torch.nn.Module()
        ref_parts.conv1 = torch.nn.Sequential(ref_sub.convbn_3d(2 * c, c, 3, 1, 1), torch.nn.ReLU(inplace=True))
        ref_parts.conv2 = torch.nn.Sequential(ref_sub.convbn_3d(c, c, 3, 1, 1), torch.nn.ReLU(inplace=True))
        ref_parts.hg_conv3d = ref_sub.hourglass(c)
        ref_parts.classifier = torch.nn.Conv3d(c, 1, kernel_size=1, padding=0, stride=1, bias=False)
        load_both(ref_parts, ours, seed)
        vol = GC.randn((1, 2 * c) + shape, seed + 1)
        v = ref_parts.conv2(ref_parts.conv1(vol))
        v1, _, _ = ref_parts.hg_conv3d(v, None, None)
        yr = ref_parts.classifier(v + v1)
        check(yr, ours(vol), name)
        out[f"global/{name}"] = yr.numpy()

# ------------------------------------------------------------- a11: grid projection producer
import snvc.dataset.KITTIRefinement_dataset as ref_ds  # noqa: E402
import snvc.dataset.kitti_util as ref_ku  # noqa: E402
from oracle import numpy_ref as NR  # noqa: E402

gp = GC.grid_proj_case()
dummy = types.SimpleNamespace(cfg=types.SimpleNamespace(x_range=gp["x_range"], y_range=gp["y_range"],
                                                        z_range=gp["z_range"], grid_resolution=gp["grid"]))
ref_ds.refinementDataset._init_3d_grid(dummy)
assert np.array_equal(dummy.grid_3d, NR.init_3d_grid(gp["x_range"], gp["y_range"], gp["z_range"], gp["grid"]))
dummy._to_cam = lambda pts, sample: ref_ds.refinementDataset._to_cam(dummy, pts, sample)
calib_l = ref_ku.Calibration(gp["P_left"], np.eye(3, 4), np.eye(3))
calib_r = ref_ku.Calibration(gp["P_right"], np.eye(3, 4), np.eye(3))
meta = {"trans_l": gp["trans_l"], "trans_r": gp["trans_r"]}
cl, cr, g3 = ref_ds.refinementDataset._generate_grid_proj(dummy, gp["samples"], calib_l, calib_r, meta)
ol, orr, og = NR.grid_projection(gp["samples"], gp["P_left"], gp["P_right"], gp["trans_l"], gp["trans_r"], dummy.grid_3d)
assert cl.dtype == np.float32 and cl.shape == ol.shape
assert np.array_equal(cl, ol) and np.array_equal(cr, orr) and np.array_equal(g3, og), "grid projection restatement differs"
out["gridproj/left_sub"] = cl[:, :, ::37]
out["gridproj/right_sub"] = cr[:, :, ::37]
out["gridproj/sum"] = np.array([cl.astype(np.float64).sum(), cr.astype(np.float64).sum(), np.abs(cl).astype(np.float64).sum()])
out["gridproj/grid3d_sub"] = g3[:, ::97]

path = os.path.join(HERE, "reference_outputs.npz")
np.savez_compressed(path, **out)
total = sum(v.nbytes for v in out.values())
print(f"wrote {path}: {len(out)} arrays, {total/1e6:.2f} MB raw, {os.path.getsize(path)/1e6:.2f} MB on disk")
