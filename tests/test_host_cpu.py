"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol that
include/snvc_hip.h declares, argument validation happens before any device work, the product
modules have the reference's state-dict keys, CPU tensors are refused (no CPU fallback), and the
one host-side op (points_in_boxes_cpu) matches the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import golden_cases as GC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from snvc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(L):
    from snvc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "snvc_hip.h")).read()
    declared = set(re.findall(r"SNVC_API\s+[\w\s\*]+?\b(snvc_\w+)\s*\(", hdr))
    assert len(declared) == 99, sorted(declared)
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert L.snvc_abi_version() == 6


def test_struct_layout_matches_header():
    from snvc_amd._lib import Conv3dDesc
    assert ctypes.sizeof(Conv3dDesc) == 18 * 4 + 3 * 8
    assert Conv3dDesc.x_batch_stride.offset == 72


def test_argument_validation_needs_no_gpu(L):
    from snvc_amd._lib import Conv3dDesc
    err = lambda: L.snvc_last_error_string().decode()  # noqa: E731
    assert L.snvc_cost_volume_forward(None, None, None, None, 1, 1, 3, 4, 1, 2, 0, None) == 1
    assert "multiples of downsample" in err()
    assert L.snvc_cost_volume_forward(None, None, None, None, 1, 1, 4, 4, 1, 1, 7, None) == 2 or "null" in err()
    assert L.snvc_cost_volume_forward(None, None, None, None, 0, 3, 4, 4, 5, 1, 0, None) == 0     # empty -> ok
    assert L.snvc_argmax_rows(None, None, None, 3, 0, None) == 1 and "empty sequence" in err()
    assert L.snvc_roiaware_pool3d_forward(None, None, None, None, None, None, None, 1, 1, 1, 128, 256, 4, 4, 0, None) == 1
    assert "< 256" in err()
    d = Conv3dDesc()
    d.N, d.Cin, d.Din, d.Hin, d.Win = 1, 8, 4, 4, 32
    d.Cout, d.Dout, d.Hout, d.Wout = 32, 4, 4, 32
    d.ksize, d.stride, d.dilation, d.pad = 3, 1, 1, 1
    # direct packing groups*chunks*taps*KP*64*MI + Winograd F(4,3) packing groups*chunks(KC=2)*9*6*KP*64
    assert L.snvc_conv3d_packed_weight_count(ctypes.byref(d)) == 1 * 2 * 27 * 2 * 64 * 1 + 1 * 4 * 9 * 6 * 1 * 64
    d.ksize, d.pad = 9, 4
    assert L.snvc_conv3d_packed_weight_count(ctypes.byref(d)) == -1 and "not in" in err()
    d.ksize, d.pad, d.Dout = 3, 1, 5
    assert L.snvc_conv3d_packed_weight_count(ctypes.byref(d)) == -1 and "convolution arithmetic" in err()
    d.Dout, d.transposed = 4, 1
    assert L.snvc_conv3d_packed_weight_count(ctypes.byref(d)) == -1 and "transposed" in err()


def test_no_cpu_fallback():
    from snvc_amd import ops
    from snvc_amd.extension.build_cost_volume import build_cost_volume
    from snvc_amd.models import submodule as S
    z = torch.zeros(1, 2, 4, 8)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        build_cost_volume(z, z, torch.zeros(1, 3), 1)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.voxel_gather_forward(z, z, torch.zeros(1, 2, 5), torch.zeros(1, 2, 5), (8, 8))
    m = S.convbn_3d(4, 32, 3, 1, 1).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        m(torch.zeros(1, 4, 4, 4, 8))
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.argmax_rows(torch.zeros(2, 5))
    # the 2D neck has no CPU route either (its torch forward is for autograd / GroupNorm on the GPU)
    for blk, x in ((S.hourglass2d(4).eval(), torch.zeros(1, 4, 8, 8)), (S.BasicBlock2d(4, 4).eval(), torch.zeros(1, 4, 8, 8)),
                   (S.hourglass2d_downsample_16(4).eval(), torch.zeros(1, 4, 16, 16))):
        with torch.no_grad(), pytest.raises(RuntimeError, match="Not implemented on the CPU"):
            blk(x, None, None) if isinstance(blk, S.hourglass2d) else blk(x)


def test_invalidate_plans_clears_every_cache():
    """Every `_snvc_*` attribute the package writes on a module is one of submodule.CACHE_ATTRS, and
    invalidate_plans(module) drops them all (a stale split-weight cache of the factored first convolution would keep
    running old weights after a broadcast / `.data` swap)."""
    from snvc_amd.models import submodule as S
    from snvc_amd.models.stereo_volume import GlobalStack
    written = set()
    for dirpath, _, files in os.walk(os.path.join(ROOT, "snvc_amd")):
        for f in files:
            if f.endswith(".py"):
                written |= set(re.findall(r"[\"\'](_snvc_\w+)[\"\']", open(os.path.join(dirpath, f)).read()))
    assert written and written <= set(S.CACHE_ATTRS), written - set(S.CACHE_ATTRS)
    m = GlobalStack(4)
    for i, mod in enumerate(m.modules()):
        for name in S.CACHE_ATTRS:
            mod.__dict__[name] = {"stale": i}
    S.invalidate_plans(m)
    assert not any(k.startswith("_snvc_") for mod in m.modules() for k in mod.__dict__)
    # the persistent inference workspace is scratch: not pickled / deep-copied, dropped by train()
    import copy
    m.__dict__["_snvc_ws"] = {("v1", (1,), "cpu"): torch.zeros(1)}
    assert "_snvc_ws" not in copy.deepcopy(m).__dict__
    m.train()
    assert "_snvc_ws" not in m.__dict__


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "snvc_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dirpath, f)


def test_state_dict_keys_equal_the_reference(tmp_path):
    """oracle.torch_ref's keys were checked against the imported reference by make_golden.py."""
    import types
    from oracle import torch_ref as T
    from snvc_amd.models import submodule as S
    from snvc_amd.models.stereo_volume import GlobalStack
    from snvc_amd.models.vernier import VernierScale

    def keys(m):
        return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]

    for gn in (False, True):
        assert keys(S.hourglass(32, gn)) == keys(T.hourglass(32, gn))
        assert keys(S.hourglass_downsample_16(32, gn)) == keys(T.hourglass_downsample_16(32, gn))
        assert keys(S.convbn_3d(64, 32, 7, 1, 3, gn=gn)) == keys(T.convbn_3d(64, 32, 7, 1, 3, gn=gn))
        assert keys(GlobalStack(32, gn)) == keys(T.GlobalStack(32, gn))
        for grid in ((16, 16, 24), (32, 128, 192)):
            cfg = types.SimpleNamespace(vernier_type="BEV_type3", backbone="hrfeat", gn=gn, grid_resolution=list(grid),
                                        resolution=(64, 64), x_range=(-1, 1), z_range=(-1, 1), num_parts=9)
            cfg.hrfeat = types.SimpleNamespace(output_channel=32, name="identity")
            cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
            assert keys(VernierScale(cfg)) == keys(T.VernierTrunk(32, grid, gn))
    # documented example keys
    k = dict(keys(VernierScale(cfg)))
    assert k["conv1.0.0.weight"] == (32, 64, 7, 7, 7)
    assert k["hg_conv3d.conv1.0.0.weight"] == (64, 32, 3, 3, 3)
    assert k["hg_conv3d.conv9.0.weight"] == (64, 64, 3, 3, 3)
    assert k["hg_conv3d.conv12.0.weight"] == (64, 32, 3, 3, 3)       # ConvTranspose: [Cin, Cout, ...]
    assert k["fg_cls_head.2.weight"] == (1, 32, 3, 3, 3)


def test_points_in_boxes_cpu_host_op(L):
    from oracle import native as O
    from snvc_amd.extension.roiaware_pool3d import roiaware_pool3d_utils as U
    r = np.random.default_rng(5)
    boxes = np.concatenate([r.uniform(-3, 3, (6, 3)), r.uniform(1, 4, (6, 3)), r.uniform(-3.2, 3.2, (6, 1))], 1).astype(np.float32)
    pts = r.uniform(-5, 5, (2000, 3)).astype(np.float32)
    exp = O.points_in_boxes_cpu(pts, boxes)
    assert np.array_equal(U.points_in_boxes_cpu(pts, boxes), exp) and exp.sum() > 20
    got = U.points_in_boxes_cpu(torch.from_numpy(pts), torch.from_numpy(boxes))
    assert isinstance(got, torch.Tensor) and np.array_equal(got.numpy(), exp)
    idmap = U.points_in_boxes_cpu_idmap(pts, boxes)
    assert idmap.shape == (2000,) and idmap.max() >= 0 and idmap.min() == -1
    assert U.points_in_boxes_cpu_idmap(pts, boxes[:0]).tolist() == [-1] * 2000
    # idmap = the largest index among the containing boxes (what the reference's loop + max(0) yields)
    exp_id = np.where(exp > 0, np.arange(len(boxes))[:, None], -1).max(0)
    assert np.array_equal(idmap, exp_id) and idmap.dtype == np.int32


def test_plan_caches_follow_parameter_updates():
    """Folded-BatchNorm / packed-weight caches are keyed on tensor versions: autograd-visible in-place
    updates invalidate them by themselves; `.data` writes need invalidate_plans()."""
    from snvc_amd.models import submodule as S
    bn = torch.nn.BatchNorm3d(4).eval()
    plan = S._Plan()
    sc0, _ = S._folded_bn(bn, plan)
    sc0 = sc0.clone()
    with torch.no_grad():
        bn.weight.mul_(2.0)                       # bumps _version
    sc1, _ = S._folded_bn(bn, plan)
    assert torch.allclose(sc1, 2 * sc0)
    bn.weight.data.mul_(2.0)                      # does NOT bump _version: stale until invalidated
    assert torch.allclose(S._folded_bn(bn, plan)[0], 2 * sc0)
    S.invalidate_plans()                          # generation counter: every cache of the process
    assert torch.allclose(S._folded_bn(bn, plan)[0], 4 * sc0)
    m = S.convbn_3d(4, 4, 3, 1, 1)
    m[0].__dict__["_snvc_plans"] = {"cpu": S._Plan()}
    S.invalidate_plans(m)                         # per-module form drops the per-device plan table
    assert "_snvc_plans" not in m[0].__dict__


def test_lazy_cost_volume_host_semantics():
    """LazyCostVolume (snvc_amd/lazy.py) without a GPU: a stand-in builder shows that the wrapper carries the volume's
    shape / dtype / device, builds once, and hands every operator, index and pointer request the built tensor."""
    import torch
    from snvc_amd.lazy import LazyCostVolume
    calls = []

    def build(left, right, shift, ds):
        calls.append(1)
        n, c, h, w = left.shape
        d = shift.shape[1]
        return torch.cat([left[:, :, None].expand(n, c, d, h, w), right[:, :, None].expand(n, c, d, h, w)], 1).contiguous()

    left, right = torch.arange(24.0).reshape(1, 2, 3, 4), -torch.arange(24.0).reshape(1, 2, 3, 4)
    shift = torch.zeros(1, 5)
    v = LazyCostVolume(left, right, shift, 1, build)
    assert tuple(v.shape) == (1, 4, 5, 3, 4) and v.dtype == torch.float32 and v.device == left.device and v.dim() == 5
    assert v.is_contiguous() and not v.is_materialized and not calls and v.sources[0] is left
    assert "materialized=False" in repr(v) and not calls
    ref = build(left, right, shift, 1)
    calls.clear()
    assert torch.equal(v + 0.0, ref) and v.is_materialized and len(calls) == 1
    assert torch.equal(v[:, 2:], ref[:, 2:]) and v.sum().item() == ref.sum().item() and len(calls) == 1     # built once
    assert v.data_ptr() == v.materialize().data_ptr() != 0
    w = LazyCostVolume(left, right, shift, 1, build)
    assert w.data_ptr() != 0 and w.is_materialized                     # a pointer request builds it too
    assert torch.equal(torch.relu(LazyCostVolume(left, right, shift, 1, build)), torch.relu(ref))


def test_install_as_snvc_redirects_the_reference_imports(tmp_path, monkeypatch):
    """snvc_amd.install_as_snvc(): the reference's import lines give this package's modules; the rest of a `snvc`
    checkout on sys.path stays the reference's own (a stand-in package tree here: the real one is not on the GPU box)."""
    import importlib
    import sys
    root = tmp_path / "ref"
    for pkg in ("snvc", "snvc/models", "snvc/extension", "snvc/extension/roiaware_pool3d", "snvc/utils"):
        (root / pkg).mkdir(parents=True)
        (root / pkg / "__init__.py").write_text("")
    (root / "snvc/models/hrnet.py").write_text("def get_model(cfg, is_train=False, **kw):\n    return ('hrnet', cfg, is_train)\n")
    (root / "snvc/models/vernier.py").write_text("raise ImportError('the reference module must not be imported')\n")
    (root / "snvc/utils/misc.py").write_text("ANSWER = 42\n")
    monkeypatch.syspath_prepend(str(root))
    for name in [m for m in sys.modules if m == "snvc" or m.startswith("snvc.")]:
        monkeypatch.delitem(sys.modules, name)
    import snvc_amd
    import snvc_amd.models.vernier as ours
    saved = ours.get_feat_extraction
    try:
        snvc_amd.install_as_snvc()
        from snvc.models.vernier import get_model, VernierScale          # noqa: F401  (the reference's import line)
        from snvc.extension.build_cost_volume import build_cost_volume  # noqa: F401
        from snvc.models.submodule import convbn_3d, hourglass           # noqa: F401
        from snvc.extension.roiaware_pool3d.roiaware_pool3d_utils import RoIAwarePool3d  # noqa: F401
        import snvc.models.vernier as v
        assert v is ours and get_model is ours.get_model and build_cost_volume.__module__.startswith("snvc_amd")
        assert importlib.import_module("snvc.utils.misc").ANSWER == 42          # everything else: the checkout's own
        assert ours.get_feat_extraction("cfg", True) == ("hrnet", "cfg", True)  # backbone factory wired in
    finally:
        ours.get_feat_extraction = saved
        for name in [m for m in sys.modules if m == "snvc" or m.startswith("snvc.")]:
            sys.modules.pop(name, None)


@pytest.mark.parametrize("q,m0", [(2, 0), (2, 3), (1, 0), (1, 2)])
def test_sheared_first_convolution_algebra_on_the_cpu(q, m0):
    """The identity behind csrc/sheared_conv.hip, checked without a GPU: for shift[d] = (m0 + d) / q the 3x3x3 convolution
    over the WARPED half of the oracle's cost volume equals the depth-1 3x7 convolutions G / G' of the interpolated right
    feature (kernels from ``sheared_kernels``: three depth classes, all columns | last column) read along the shear with
    ``sheared_geometry``'s offsets -- every border included (d = 0, D-1, w = 0, W-1, samples left of the image)."""
    import torch.nn.functional as F
    from oracle import native as O
    from snvc_amd.models.submodule import sheared_geometry, sheared_kernels
    r = np.random.default_rng(7 + 10 * q + m0)
    C, CO, D, H, W = 3, 4, 9, 5, 12
    L = r.standard_normal((1, C, H, W)).astype(np.float32)
    R = r.standard_normal((1, C, H, W)).astype(np.float32)
    shift = ((m0 + np.arange(D)) / q).astype(np.float32)[None]
    wr = torch.from_numpy(r.standard_normal((CO, C, 3, 3, 3))).double()
    vol_r = torch.from_numpy(O.cost_volume_forward(L, R, shift, 1)[:, C:]).double()          # the warped half [1,C,D,H,W]
    ref = F.conv3d(vol_r, wr, padding=1)[0].numpy()
    # Rq on the padded grid, exactly as snvc_sheared_upsample lays it out
    off, wu, off_col, wu_col = sheared_geometry(q, m0, D, W)

    def rq(width, o):
        out = np.zeros((C, H, width))
        for i in range(width):
            u = i - o
            if 0 <= u <= q * (W - 1):
                j = u // q
                out[:, :, i] = R[0, :, :, j] if u % q == 0 else 0.5 * R[0, :, :, j] + 0.5 * R[0, :, :, j + 1]
        return torch.from_numpy(out)[None]

    k = sheared_kernels(wr, q)                                                                 # [2,3,CO,C,3,7]
    g = [F.conv2d(rq(wu, off), k[0, cls], padding=(1, 3))[0].numpy() for cls in range(3)]
    gc = [F.conv2d(rq(wu_col, off_col), k[1, cls], padding=(1, 3))[0].numpy() for cls in range(3)]
    got = np.zeros_like(ref)
    for d in range(D):
        cls = 0 if d == 0 else (2 if d == D - 1 else 1)
        for w in range(W - 1):
            i = q * w - d - m0 + off
            if 0 <= i < wu:
                got[:, d, :, w] = g[cls][:, :, i]
        got[:, d, :, W - 1] = gc[cls][:, :, q * (W - 1) - d - m0 + off_col]
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


def test_sheared_term_counts_cover_every_element():
    """The multiplicities the fused BatchNorm backward multiplies its constant term with: every element of the layer's
    [D, W] plane lies on exactly one shear line (columns < W-1, minus the lines left of the padded grid, where G is zero) or
    in one last-column slot, and in one depth class."""
    from snvc_amd.models.submodule import _sheared_term_counts, sheared_geometry
    for q, m0, D, W in ((2, 0, 192, 312), (1, 3, 12, 40), (2, 5, 9, 24)):
        line, col, per_class = _sheared_term_counts(q, m0, D, W, torch.device("cpu"))
        off, wu, off_col, wu_col = sheared_geometry(q, m0, D, W)
        dropped = sum(1 for d in range(D) for w in range(W - 1) if not 0 <= q * w - d - m0 + off < wu)
        assert line.shape == (3, wu) and col.shape == (3, wu_col)
        assert int(line.sum()) == D * (W - 1) - dropped
        assert int(col.sum()) == D and float(col.max()) == 1.0
        assert per_class.tolist() == [1.0, float(D - 2), 1.0]
        assert int(line[0].sum()) + int(line[2].sum()) <= 2 * (W - 1)


def _warped_first_conv_reference(R, wr, shift, D):
    """csrc/sheared_conv.hip's warp-after-convolution identity in plain fp64 numpy/torch (the arithmetic of warped_expand_kernel):
    returns conv3d(warped half of the volume, wr) for an arbitrary shift array [D]."""
    import torch.nn.functional as F
    C, H, W = R.shape
    CO = wr.shape[0]
    Rt = torch.from_numpy(R)[None].double()
    P = [F.conv2d(Rt, wr[:, :, kd], padding=1)[0].numpy() for kd in range(3)]                    # [CO,H,W]
    kq = [torch.zeros_like(wr[:, :, 0]) for _ in range(3)]
    for kd in range(3):
        kq[kd][:, :, :, 1] = wr[:, :, kd, :, 2]                                                  # the kw = +1 taps on the centre column
    Q = [F.conv2d(Rt, kq[kd], padding=1)[0].numpy() for kd in range(3)]
    E = np.zeros((3, 3, CO, H))
    col0 = F.pad(Rt[:, :, :, :1], (0, 0, 1, 1))                                                   # first column, padded in h
    for kd in range(3):
        for kw in range(3):
            E[kd, kw] = F.conv2d(col0, wr[:, :, kd, :, kw:kw + 1])[0, :, :, 0].numpy()

    def z(a, j):            # zero-extended read
        return a[:, :, j] if 0 <= j < W else np.zeros(a.shape[:2])

    out = np.zeros((CO, D, H, W))
    for d in range(D):
        for kd in range(3):
            dd = d + kd - 1
            if not 0 <= dd < D:
                continue
            s = float(shift[dd]); m = int(np.floor(s)); f = s - m; g = 1.0 - f
            for w in range(W):
                out[:, d, :, w] += f * z(P[kd], w - m - 1) + g * z(P[kd], w - m)
            out[:, d, :, W - 1] -= f * z(Q[kd], W - m - 1) + g * z(Q[kd], W - m)
            if m <= W:
                add = [g * E[kd, 2], f * E[kd, 2], 0.0 * E[kd, 2]]
                if f > 0:
                    if m - 1 != W - 1:
                        add[0] = add[0] - g * E[kd, 2]
                    add[1] = add[1] - g * E[kd, 1]
                    add[2] = add[2] - g * E[kd, 0]
                for j in range(3):
                    w = m - 1 + j
                    if 0 <= w < W:
                        out[:, d, :, w] += add[j]
    return out


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_warp_after_convolution_algebra_on_the_cpu(seed):
    """For ANY shift array the 3x3x3 convolution over the warped half equals three interpolations of three 2D convolutions of the
    right feature plus two border terms (csrc/sheared_conv.hip, warped_expand_kernel): checked in fp64 against F.conv3d over the
    oracle's cost volume for random fractional / integer / zero / beyond-the-image shifts, all borders included."""
    import torch.nn.functional as F
    from oracle import native as O
    r = np.random.default_rng(50 + seed)
    C, CO, D, H, W = 3, 4, 10, 5, 12
    L = r.standard_normal((1, C, H, W)).astype(np.float32)
    R = r.standard_normal((1, C, H, W)).astype(np.float32)
    shift = r.uniform(0, W + 2, D)
    shift[::3] = np.floor(shift[::3])                 # whole-pixel planes
    shift[1] = 0.0
    shift[2] = float(W)                               # every sample left of the image
    shift[4] = W - 1 + 0.5
    shift[5] = 0.25
    shift = shift.astype(np.float32)[None]
    wr = torch.from_numpy(r.standard_normal((CO, C, 3, 3, 3))).double()
    vol_r = torch.from_numpy(O.cost_volume_forward(L, R, shift, 1)[:, C:]).double()
    ref = F.conv3d(vol_r, wr, padding=1)[0].numpy()
    got = _warped_first_conv_reference(R[0].astype(np.float64), wr, shift[0].astype(np.float64), D)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)      # the oracle interpolates in fp32: ~1e-7 of the values


def test_amax_tags_follow_the_tensor_version():
    """ops.tag_amax / amax_of (r6): the word a producer pass left is handed on only while the tensor is unchanged."""
    from snvc_amd import ops
    t = torch.zeros(4)
    w = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32)
    assert ops.amax_of(t) is None
    ops.tag_amax(t, w)
    assert ops.amax_of(t) is w
    t.add_(1.0)
    assert ops.amax_of(t) is None
    ops.tag_amax(t, None)
    assert ops.amax_of(t) is None
    b = ops.amax_from_bound(torch.tensor(-3.5))
    assert b.shape == (ops.AMAX_SLOTS,) and b[0].item() == torch.tensor(3.5).view(torch.int32).item() and int(b[1:].abs().sum()) == 0


def test_twin_tags_and_the_training_route_on_cpu_tensors():
    """ops.tag_twin / twin_of (r6): a split twin is handed on only while its float32 tensor is unchanged; twin_ok / the route say no to
    anything that is not a float32 CUDA tensor with whole channel groups (a CPU tensor keeps the reference's own torch path)."""
    import torch.nn as nn
    from snvc_amd import ops
    from snvc_amd.models import submodule as S
    t = torch.zeros(1, 8, 2, 2, 4)
    pair, mul = torch.zeros(1, 2, 1, 2, 2, 4, 8, dtype=torch.float16), torch.ones(1)
    assert ops.twin_of(t) is None
    ops.tag_twin(t, pair, mul)
    got = ops.twin_of(t)
    assert got is not None and got[0] is pair and got[1] is mul
    t.mul_(2.0)
    assert ops.twin_of(t) is None
    assert not ops.twin_ok(t)                                            # CPU
    assert tuple(ops.twin_empty(t).shape) == (1, 2, 1, 2, 2, 4, 8)
    assert not S._x3_train_route(nn.Conv3d(32, 32, 3, 1, 1, bias=False), torch.zeros(1, 32, 4, 4, 8))
    assert S.X3_TRAIN == [True] and S.X3_TRAIN_MIN_CC == [1024]
    # a skip connection's view keeps the maximum its source was tagged with (the consumer's twin bound needs it)
    x = torch.zeros(2, 3, requires_grad=True) * 1.0
    w = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32)
    ops.tag_amax(x, w)
    v = S._SkipTap.apply(x, S._GradBox())
    assert ops.amax_of(v) is w


def test_build_entry_checks_the_abi_the_binding_was_written_against():
    """__graft_entry__.build() compares the library's ABI with snvc_amd/_lib.py's _ABI, not with a literal that an ABI bump leaves behind
    (r6: the literal said 5 after the bump to 6 -- the driver's build check would have failed on a library that was fine)"""
    import os, re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "__graft_entry__.py")).read()
    assert "snvc_abi_version() == _lib._ABI" in src
    assert not re.search(r"snvc_abi_version\(\) == \d", src)
