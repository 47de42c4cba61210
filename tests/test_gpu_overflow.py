"""Split mode ("f16x3") must never hand a clamped result to the caller.

A split tensor's exponent is chosen from its BatchNorm's PARAMETERS (|beta| + 64 |gamma|).  Statistics that do not match the
data -- here: a running mean 3000 standard deviations off on one layer -- push activations beyond half's range; the epilogue clamps
and raises the overflow flag.  r4 read that flag one call late.  r5 reads it inside the call (the copy is queued behind the last
layer that can clamp) and redoes a flagged call on the fp32-MFMA kernels: the RETURNED tensor is the fp32 path's.
"""
import copy
import types
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _global_stack(break_layer=None):
    import bench
    from snvc_amd.models.stereo_volume import GlobalStack
    m = GlobalStack(32)
    m.load_state_dict(bench.seeded_state(m, 11))
    if break_layer is not None:
        with torch.no_grad():       # statistics that do not match the data: the layer's output is ~50 times what they promise
            break_layer(m).running_mean.sub_(3000.0)
    return m.eval().to(dev())


def _pair(d=16, h=8, w=40, c=32, seed=3):
    g = np.random.default_rng(seed)
    left = torch.from_numpy(g.standard_normal((1, c, h, w)).astype(np.float32)).to(dev())
    right = torch.from_numpy(g.standard_normal((1, c, h, w)).astype(np.float32)).to(dev())
    shift = torch.from_numpy((np.arange(d, dtype=np.float32) * 0.5)[None]).to(dev())
    return left, right, shift


def _close(got, exp, what, tol=1e-4):
    err = (got.double() - exp.double()).abs().max().item() / max(exp.double().abs().max().item(), 1e-30)
    assert torch.isfinite(got).all() and err <= tol, f"{what}: max|err|/max|ref| = {err:.2e}"


BREAKS = {"conv2": lambda m: m.conv2[0][1], "hg_conv1": lambda m: m.hg_conv3d.conv1[0][1], "hg_conv4": lambda m: m.hg_conv3d.conv4[0][1],
          "conv1": lambda m: m.conv1[0][1]}


@pytest.mark.parametrize("where", list(BREAKS))
@pytest.mark.parametrize("entry", ["forward_pair", "reference_api", "general_shift"])
def test_global_stack_overflowing_call_returns_the_fp32_result(where, entry):
    from snvc_amd.extension.build_cost_volume import build_cost_volume
    from snvc_amd.models import submodule as S
    m = _global_stack(BREAKS[where])
    ref = copy.deepcopy(m)
    ref.arithmetic = "fp32"
    left, right, shift = _pair()
    if entry == "general_shift":
        shift = shift * 0.73 + 0.1

    def run(model):
        if entry == "reference_api":
            return model(build_cost_volume(left, right, shift, 1))
        return model.forward_pair(left, right, shift, 1)

    with torch.no_grad():
        exp = run(ref)
        redo = S._ROUTES["x3_overflow_redo"]
        with pytest.warns(UserWarning, match="overflow"):
            got = run(m)
        assert S._ROUTES["x3_overflow_redo"] == redo + 1          # the call ran in split mode, was flagged and redone
        _close(got, exp, f"overflow at {where} via {entry}: returned tensor vs the fp32 path")
        assert m.__dict__.get("_snvc_x3_off")
        with warnings.catch_warnings():                                 # later calls: fp32 kernels, silently
            warnings.simplefilter("error")
            _close(run(m), exp, "call after the overflow")
        m.arithmetic = "x3"
        with pytest.raises(RuntimeError, match="x3"):
            run(m)


def test_global_stack_demanded_split_mode_raises_instead_of_returning_a_clamped_result():
    m = _global_stack(BREAKS["conv2"])
    m.arithmetic = "x3"
    left, right, shift = _pair()
    with torch.no_grad(), pytest.raises(RuntimeError, match="overflow"):
        m.forward_pair(left, right, shift, 1)


def test_global_stack_deferred_check_is_opt_in_and_explicit():
    """``overflow_check = "deferred"`` (what r4 did; kept to measure what the check costs): the flagged call's result IS clamped,
    ``check_overflow()`` says so synchronously, and the flag survives a rebuild of the packed split-mode state."""
    from snvc_amd.models.submodule import invalidate_plans
    m = _global_stack(BREAKS["conv2"])
    ref = copy.deepcopy(m)
    ref.arithmetic = "fp32"
    m.overflow_check = "deferred"
    left, right, shift = _pair()
    with torch.no_grad():
        exp = ref.forward_pair(left, right, shift, 1)
        got = m.forward_pair(left, right, shift, 1)
        err = (got - exp).abs().max().item() / exp.abs().max().item()
        assert err > 1e-3                                               # clamped: this is the result r4 returned silently
        m.__dict__.pop("_snvc_x3", None)                                # the packed state is rebuilt; the pending flag is not lost
        with pytest.warns(UserWarning, match="overflow"):
            assert m.check_overflow() is True
        assert m.check_overflow() is False
        _close(m.forward_pair(left, right, shift, 1), exp, "after check_overflow(): fp32 kernels")
        m.reset_split_mode()
        invalidate_plans(m)
        with pytest.warns(UserWarning, match="overflow"):               # the next call looks at the previous call's flag
            m.forward_pair(left, right, shift, 1)
            got = m.forward_pair(left, right, shift, 1)
        _close(got, exp, "deferred: the call after a flagged one runs in fp32")


def test_global_stack_matching_statistics_stay_in_split_mode():
    from snvc_amd.models import submodule as S
    m = _global_stack()
    left, right, shift = _pair()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        before = S._ROUTES["x3_tail"]
        for _ in range(3):
            m.forward_pair(left, right, shift, 1)
        assert S._ROUTES["x3_tail"] == before + 3 and not m.__dict__.get("_snvc_x3_off")


@pytest.mark.parametrize("factor", [100.0, 1e-6])
def test_lazy_volume_written_in_place_is_scaled_by_its_own_maximum(factor):
    """``vol = build_cost_volume(...); vol.mul_(100); model(vol)`` -- the split scale must come from the modified
    volume, not from the features it was built from (hi = inf, lo = -inf, NaN out before)."""
    from snvc_amd.extension.build_cost_volume import build_cost_volume
    m = _global_stack()
    ref = copy.deepcopy(m)
    ref.arithmetic = "fp32"
    left, right, shift = _pair()
    with torch.no_grad():
        vol = build_cost_volume(left, right, shift, 1)
        vol.mul_(factor)
        eager = vol.clone()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")         # factor 100 may or may not leave the range conv1's statistics promise
            got = m(vol)
        _close(got, ref(eager), f"model(vol.mul_({factor}))")


def _local_model(grid=(16, 16, 24), F_=32, seed=2024):
    import bench
    from snvc_amd.models.vernier import VernierScale
    cfg = types.SimpleNamespace(vernier_type="BEV_type3", backbone="hrfeat", gn=False, grid_resolution=list(grid),
                                resolution=(256, 256), x_range=(-1.0, 1.0), z_range=(-1.0, 1.0), num_parts=9)
    cfg.hrfeat = types.SimpleNamespace(output_channel=F_, name="identity")
    cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
    m = VernierScale(cfg)
    m.load_state_dict(bench.seeded_state(m, seed))
    return m


LOCAL_BREAKS = {"conv1": lambda m: m.conv1[0][1], "conv3": lambda m: m.conv3[0][1], "vimg_feat": lambda m: m.vimg_feat[0][1],
                "fg_cls_head": lambda m: m.fg_cls_head[0][1]}


@pytest.mark.parametrize("grid", [(16, 16, 24), (16, 32, 48)], ids=["hourglass", "hourglass16"])
@pytest.mark.parametrize("where", list(LOCAL_BREAKS))
def test_local_trunk_overflowing_call_returns_the_fp32_result(grid, where):
    from snvc_amd.models import submodule as S
    m = _local_model(grid)
    with torch.no_grad():
        LOCAL_BREAKS[where](m).running_mean.sub_(3000.0)
    m = m.eval().to(dev())
    ref = copy.deepcopy(m)
    ref.precision = "f32"
    g = np.random.default_rng(5)
    v = grid[0] * grid[1] * grid[2]
    lf = torch.from_numpy(g.standard_normal((2, 32, 64, 64)).astype(np.float32)).to(dev())
    rf = torch.from_numpy(g.standard_normal((2, 32, 64, 64)).astype(np.float32)).to(dev())
    gl = torch.from_numpy(g.uniform(-8, 264, (2, 2, v)).astype(np.float32)).to(dev())
    gr = torch.from_numpy(g.uniform(-8, 264, (2, 2, v)).astype(np.float32)).to(dev())
    with torch.no_grad():
        bev32, occ32, _ = ref.trunk_3d(ref.construct_voxel(lf, rf, gl, gr))
        redo = S._ROUTES["x3_overflow_redo"]
        with pytest.warns(UserWarning, match="overflow"):
            bev, occ, _ = m.trunk_3d(m.construct_voxel(lf, rf, gl, gr))
        assert S._ROUTES["x3_overflow_redo"] == redo + 1
        _close(bev, bev32, f"local trunk, overflow at {where}: bev")
        _close(occ, occ32, f"local trunk, overflow at {where}: occupancy")
        # the whole model, starting from the gather that writes the split pair itself
        m.reset_split_mode()
        with pytest.warns(UserWarning, match="overflow"):
            out = m(lf, rf, gl.clone(), gr.clone())
        exp = ref(lf, rf, gl.clone(), gr.clone())
        for k in ("ncf", "occupancy"):
            _close(out[k], exp[k], f"VernierScale.forward, overflow at {where}: {k}", 2e-4)
        m.reset_split_mode()
        m.precision = "x3"
        with pytest.raises(RuntimeError, match="overflow"):
            m.trunk_3d(m.construct_voxel(lf, rf, gl, gr))
        with pytest.raises(RuntimeError, match="switched off"):       # precision='x3' is split mode or an error
            m.trunk_3d(m.construct_voxel(lf, rf, gl, gr))


@pytest.mark.parametrize("algo", ["default", "q16", "serial", "small"])
def test_flag_looks_at_what_is_stored_after_the_relu(algo):
    """a large NEGATIVE pre-activation that the ReLU zeroes is not an overflow (the 16x16x32 form measured |v| before
    the activation and dropped the model to fp32 for nothing); a large positive one is, in every kernel form."""
    from snvc_amd import _lib, ops
    torch.manual_seed(1)
    a = {"default": 0, "q16": _lib.ALGO_X3_Q16, "serial": _lib.ALGO_X3_SERIAL, "small": _lib.ALGO_X3_SMALL}[algo]
    x = torch.randn(1, 32, 9, 10, 70, device=dev())
    w = torch.randn(32, 32, 3, 3, 3, device=dev()) * 0.05
    layer = ops.Conv3dLayerX3(w, 3, 1, 1, 1, algo=a)
    scale = torch.ones(32, device=dev())
    xs = ops.to_split(x, 3)
    flag = torch.zeros(1, dtype=torch.int32, device=dev())
    y = layer(xs, 3, scale, torch.full((32,), -1e6, device=dev()), flags=ops.EPI_RELU, out_exp=2, overflow=flag)
    assert int(flag.item()) == 0 and ops.from_split(y, 2).abs().max().item() == 0.0
    layer(xs, 3, scale, torch.full((32,), 1e6, device=dev()), flags=ops.EPI_RELU, out_exp=2, overflow=flag)
    assert int(flag.item()) == 1
    flag.zero_()
    layer(xs, 3, scale, torch.full((32,), -1e6, device=dev()), flags=0, out_exp=2, overflow=flag)      # no activation: -1e6 IS stored
    assert int(flag.item()) == 1


def test_groupnorm_trunk_extreme_input_stays_exact():
    """A GroupNorm layer's result is gamma * xhat + beta with xhat normalised per sample and group: |xhat| <= sqrt(m) for a group of m
    elements -- bounded by construction, unlike a frozen BatchNorm's.  The worst case for the 64-sigma exponent is one voxel carrying
    the whole signal through the 1x1x1 layer (vimg_feat: xhat = sqrt(16 * 32 * 48) = 157 there): the split pass must either represent it
    (the concat's shared exponent leaves room for 2 x (|beta| + 64 |gamma|) x 4 tensors) or flag and redo -- never return a clamped value.
    (The clamp + flag of the pass itself: tests/test_gpu_tail.py::test_affine_act_split_vs_torch.)"""
    import bench
    from snvc_amd.models import submodule as S
    grid = (16, 32, 48)
    m = bench.local_model(grid, 32, dev(), gn=True)
    ref = copy.deepcopy(m)
    ref.precision = "f32"
    vox = torch.zeros(1, 64, *grid, device=dev())
    vox[0, :, 7, 13, 21] = torch.linspace(1.0, 3.0, 64, device=dev())
    with torch.no_grad():
        bev32, occ32, _ = ref.trunk_3d(vox)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            bev, occ, _ = m.trunk_3d(vox)
        _close(bev, bev32, "GroupNorm trunk, one-voxel input: bev")
        _close(occ, occ32, "GroupNorm trunk, one-voxel input: occupancy")
        # ordinary data stays in split mode
        m.reset_split_mode()
        g = np.random.default_rng(9)
        dense = torch.from_numpy(g.standard_normal((1, 64) + grid).astype(np.float32)).to(dev())
        b = S._ROUTES["x3_local_trunk"]
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            m.trunk_3d(dense)
        assert S._ROUTES["x3_local_trunk"] == b + 1 and not m.__dict__.get("_snvc_x3_off")


def test_conv5_cannot_clamp_at_the_limit_its_inputs_allow():
    """the overflow flag is posted BEFORE conv5 (the host's look at it overlaps conv5 + the tail gather) on the argument that
    conv5's result carries a HARD exponent -- L1 norm of its folded weights x the largest value its input's exponent can hold + the
    largest `pre` -- and so cannot clamp.  Driven to that limit here: every weight made non-negative, the BatchNorm scale positive,
    `o` and `pre` at the largest split value (hi = 65504, lo = +16) everywhere, so that every term of every interior output adds up
    to the bound itself: the flag must stay 0 and the tail finite."""
    from snvc_amd import ops
    m = _global_stack()
    hg = m.hg_conv3d
    with torch.no_grad():
        hg.conv5[0].weight.abs_()
        hg.conv5[1].weight.abs_().add_(0.5)
        hg.conv5[1].running_mean.zero_()
        hg.conv5[1].bias.abs_()
    st = m._x3_state(dev())
    assert st is not None and st["tail"] is not None
    L, A, E = st["layers"], st["affine"], st["exp"]
    n, d4, h4, w4 = 1, 4, 4, 20
    hi = torch.full((n, 1, 8, d4, h4, w4, 8), 65504.0, dtype=torch.float16, device=dev())
    lo = torch.full_like(hi, 16.0)
    o = torch.cat([hi, lo], dim=1).contiguous()                                   # [n, 2, 64/8, D/4, H/4, W/4, 8] at exponent E["h4"]
    pre = torch.cat([torch.full((n, 1, 8, 2 * d4, 2 * h4, 2 * w4, 8), 65504.0, dtype=torch.float16, device=dev()),
                     torch.full((n, 1, 8, 2 * d4, 2 * h4, 2 * w4, 8), 16.0, dtype=torch.float16, device=dev())], dim=1).contiguous()
    flag = torch.zeros(1, dtype=torch.int32, device=dev())
    t = L["h5"].forward_tail(o, E["h4"], *A["h5"], st["tail"], residual=pre, res_exp=E["h2"], flags=ops.EPI_RELU | ops.EPI_ADD_PRE,
                             out_exp=E["post"], overflow=flag, out=torch.empty((n, 27, 8, d4, h4, w4), device=dev()))
    torch.cuda.synchronize()
    assert int(flag.item()) == 0, "conv5 clamped inside the range its hard exponent promises"
    assert torch.isfinite(t).all()
    # and the bound is not vacuous: one exponent step less headroom does clamp on this input
    flag.zero_()
    L["h5"].forward_tail(o, E["h4"], *A["h5"], st["tail"], residual=pre, res_exp=E["h2"], flags=ops.EPI_RELU | ops.EPI_ADD_PRE,
                         out_exp=E["post"] + 2, overflow=flag, out=torch.empty((n, 27, 8, d4, h4, w4), device=dev()))
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
