"""The BENCHMARKED step against the CPU oracle at the size it is benchmarked at.

cfg2 (BASELINE.json configs[1]: features [1,32,96,312], 192 disparity planes) and cfg1 (configs[0]: 64 planes,
SURVEY 8(d)'s mixed whole / half-pixel shifts) run ONCE through the oracle on the host -- C restatement of
BuildCostVolume_cuda.cu:63-98 + the torch-CPU restatement of submodule.py:85-168 that tests/golden pins bit-equal to
the imported reference -- with every intermediate of the stack kept.  Then:

  * every entry point of the HIP path (sheared, any-shift, built right half, materialised, the reference's own call
    sequence) against the oracle's cost on ALL 5.75 M (1.9 M) outputs: max-normalised error <= 1e-4 and north_star's
    1e-3 relative criterion element by element (``check``);
  * every layer of the stack on the ORACLE's input of that layer against the oracle's output of it, whole tensors
    at the exact-fp32 tolerance: conv2 + its side head, the hourglass's stride-2 / stride-1 / transposed layers and
    the folded one-channel tail -- so no layer's error can hide behind the next one's.

bench.py uses the same inputs and weights (make_inputs(0), seeded_state): its ``parity_vs_cpu_baseline`` field is this
comparison repeated on the timed run's own output.
"""
import numpy as np
import pytest
import torch

from test_gpu_parity import TIGHT, check

pytestmark = pytest.mark.gpu
C, H, W = 32, 96, 312


def dev():
    return torch.device("cuda:0")


def _oracle_run(d, shift_np):
    """One pair through the oracle, every intermediate kept (host tensors)."""
    import bench
    from oracle import native as O
    from oracle import torch_ref as T
    torch.set_num_threads(max(1, torch.get_num_threads()))
    left, right, _ = bench.make_inputs(0, "cpu", d)
    ref = T.GlobalStack(C)
    ref.load_state_dict(bench.seeded_state(ref))
    ref.eval()
    o = {"left": left, "right": right, "shift": torch.from_numpy(shift_np), "ref": ref}
    with torch.no_grad():
        vol = torch.from_numpy(O.cost_volume_forward(left.numpy(), right.numpy(), shift_np, 1))
        o["v1"] = ref.conv1(vol)
        del vol
        o["v2"] = ref.conv2(o["v1"])
        hg = ref.hg_conv3d
        o["h1"] = hg.conv1(o["v2"])
        o["pre"] = torch.relu(hg.conv2(o["h1"]))
        o["h3"] = hg.conv3(o["pre"])
        o["h4"] = hg.conv4(o["h3"])
        o["post"] = torch.relu(hg.conv5(o["h4"]) + o["pre"])
        o["hv"] = ref.classifier(o["v2"])
        o["cost"] = ref.classifier(o["v2"] + hg.conv6(o["post"]))
    return o


def _model():
    import bench
    from snvc_amd.models.stereo_volume import GlobalStack
    m = GlobalStack(C)
    m.load_state_dict(bench.seeded_state(m))
    return m.eval().to(dev())


@pytest.fixture(scope="module")
def cfg2():
    import bench
    shift = np.linspace(0.0, (bench.D - 1) / 2.0, bench.D, dtype=np.float32)[None].copy()    # == bench.make_inputs
    return _oracle_run(bench.D, shift)


@pytest.fixture(scope="module")
def cfg1():
    d = 64
    shift = (np.linspace(0, 63, d) + 0.5 * (np.arange(d) % 2)).astype(np.float32)[None]
    return _oracle_run(d, shift)


def _entry_points(o, expect_route, arithmetic):
    """Every way into the step, against the oracle's cost: all outputs, check()'s three criteria.  ``arithmetic``: "fp32" (the
    fp32-MFMA kernels) or "x3" (conv2 + hourglass on the split-mode kernels: three half-precision MFMAs per product)."""
    from snvc_amd.extension.build_cost_volume import build_cost_volume
    from snvc_amd.models import submodule as S
    m = _model()
    m.arithmetic = arithmetic
    dl, dr, ds = o["left"].to(dev()), o["right"].to(dev()), o["shift"].to(dev())
    exp = o["cost"].numpy()
    with torch.no_grad():
        before, before_x3 = S._ROUTES[expect_route], S._ROUTES["x3_tail"]
        got = m.forward_pair(dl, dr, ds, 1).cpu().numpy()
        assert S._ROUTES[expect_route] == before + 1, f"forward_pair did not take the {expect_route} route"
        assert S._ROUTES["x3_tail"] == before_x3 + (1 if arithmetic == "x3" else 0)
        check(got, exp, 1e-4, f"forward_pair ({expect_route}) vs oracle, full size")
        v1 = m.last_first_layer()
        check(v1.cpu().numpy(), o["v1"].numpy(), TIGHT, f"first layer ({expect_route}) vs oracle, full size")
        if expect_route != "commuted_first_conv":
            before = S._ROUTES["commuted_first_conv"]
            got = m.forward_pair(dl, dr, ds, 1, sheared=False).cpu().numpy()
            assert S._ROUTES["commuted_first_conv"] == before + 1
            check(got, exp, 1e-4, "forward_pair (any shift array: warp after convolution) vs oracle, full size")
            v1 = m.last_first_layer()
            check(v1.cpu().numpy(), o["v1"].numpy(), TIGHT, "first layer (warp after convolution) vs oracle, full size")
        got = m.forward_pair(dl, dr, ds, 1, sheared=False, commuted=False).cpu().numpy()
        check(got, exp, 1e-4, "forward_pair (right half built) vs oracle, full size")
        got = m.forward_pair(dl, dr, ds, 1, factored=False).cpu().numpy()
        check(got, exp, 1e-4, "forward_pair (materialised volume) vs oracle, full size")
        got = m(build_cost_volume(dl, dr, ds, 1)).cpu().numpy()            # the reference's call sequence (lazy volume)
        check(got, exp, 1e-4, "model(build_cost_volume(...)) vs oracle, full size")
        vol = build_cost_volume(dl, dr, ds, 1)
        vol.data_ptr()                                                       # somebody looked: the real volume
        got = m(vol).cpu().numpy()
        check(got, exp, 1e-4, "model(materialised lazy volume) vs oracle, full size")
    if arithmetic == "x3":      # no value left the range its BatchNorm statistics promised (nothing was clamped)
        assert int(m.__dict__["_snvc_x3"]["flag"].item()) == 0 and not m.__dict__.get("_snvc_x3_off")


@pytest.mark.parametrize("arithmetic", ["fp32", "x3"])
def test_cfg2_every_entry_point_vs_oracle_full_size(cfg2, arithmetic):
    _entry_points(cfg2, "sheared_first_conv", arithmetic)


@pytest.mark.parametrize("arithmetic", ["fp32", "x3"])
def test_cfg1_every_entry_point_vs_oracle_full_size(cfg1, arithmetic):
    """cfg1's shifts (linspace(0,63,64) + 0.5 on odd planes) are not uniformly spaced: the warp-after-convolution path."""
    _entry_points(cfg1, "commuted_first_conv", arithmetic)


def test_cfg2_every_layer_on_the_oracles_input_full_size(cfg2):
    """conv2 + side head, hourglass conv1 (stride 2 on 192x96x312), conv2-conv5, and the folded one-channel tail
    (deconv3d_cout1): each fed the oracle's input of that layer, whole output tensors at the exact-fp32 tolerance."""
    from snvc_amd.models import submodule as S
    o = cfg2
    m = _model()
    m.arithmetic = "fp32"
    hg = m.hg_conv3d
    g = lambda k: o[k].to(dev())                                                                     # noqa: E731
    with torch.no_grad():
        b_s = S._ROUTES["side_head"]
        v2, hv = m.conv2.fused(g("v1"), side_head=m.classifier)
        assert S._ROUTES["side_head"] == b_s + 1
        check(v2.cpu().numpy(), o["v2"].numpy(), TIGHT, "conv2 (Winograd F(4,3), 32->32 on 192x96x312)")
        check(hv.cpu().numpy(), o["hv"].numpy(), TIGHT, "conv2's side head = classifier(v2)")
        del v2, hv
        h1 = hg.conv1.fused(g("v2"))
        check(h1.cpu().numpy(), o["h1"].numpy(), TIGHT, "hourglass conv1 (k3 stride 2, 32->64)")
        del h1
        pre = hg.conv2.fused(g("h1"), relu=True)
        check(pre.cpu().numpy(), o["pre"].numpy(), TIGHT, "hourglass conv2 (64->64 on 96x48x156)")
        del pre
        h3 = hg.conv3.fused(g("pre"))
        check(h3.cpu().numpy(), o["h3"].numpy(), TIGHT, "hourglass conv3 (k3 stride 2, 64->64)")
        del h3
        h4 = hg.conv4(g("h3"))
        check(h4.cpu().numpy(), o["h4"].numpy(), TIGHT, "hourglass conv4 (64->64 on 48x24x78)")
        del h4
        post = hg.conv5.fused(g("h4"), relu=True, residual=g("pre"))
        check(post.cpu().numpy(), o["post"].numpy(), TIGHT, "hourglass conv5 (transposed 64->64 + pre, ReLU)")
        del post
        b_f = S._ROUTES["folded_head"]
        cost = hg.conv6.fused(g("post"), residual=g("v2"), head=m.classifier, head_residual=g("hv"))
        assert S._ROUTES["folded_head"] == b_f + 1
        check(cost.cpu().numpy(), o["cost"].numpy(), TIGHT, "folded tail: classifier(bn(deconv(post)) + v2) as one transposed layer to one channel")


def test_cfg2_every_split_mode_layer_on_the_oracles_input_full_size(cfg2):
    """The same layer-by-layer comparison for the split-mode (f16x3) kernels the default inference path runs conv2 and the
    hourglass on: each layer fed the oracle's input of that layer (split with the exponent the model chose), whole output
    tensors against the oracle's at the SAME exact-fp32 tolerance as the fp32 kernels."""
    from snvc_amd import ops
    o = cfg2
    m = _model()
    st = m._x3_state(dev())
    assert st is not None
    L, A, E = st["layers"], st["affine"], st["exp"]
    sp = lambda k, e: ops.to_split(o[k].to(dev()), e)                                               # noqa: E731
    with torch.no_grad():
        flag = torch.zeros(1, dtype=torch.int32, device=dev())
        v2s, hv = L["conv2"](sp("v1", E["v1"]), E["v1"], *A["conv2"], flags=ops.EPI_RELU, out_exp=E["conv2"], head=m.classifier.weight,
                             overflow=flag)
        check(ops.from_split(v2s, E["conv2"]).cpu().numpy(), o["v2"].numpy(), TIGHT, "split conv2 (32->32 on 192x96x312)")
        check(hv.cpu().numpy(), o["hv"].numpy(), TIGHT, "split conv2's side head = classifier(v2)")
        del v2s, hv
        h1 = L["h1"](sp("v2", E["conv2"]), E["conv2"], *A["h1"], flags=ops.EPI_RELU, out_exp=E["h1"], overflow=flag)
        check(ops.from_split(h1, E["h1"]).cpu().numpy(), o["h1"].numpy(), TIGHT, "split hourglass conv1 (k3 stride 2, 32->64)")
        del h1
        pre = L["h2"](sp("h1", E["h1"]), E["h1"], *A["h2"], flags=ops.EPI_RELU, out_exp=E["h2"], overflow=flag)
        check(ops.from_split(pre, E["h2"]).cpu().numpy(), o["pre"].numpy(), TIGHT, "split hourglass conv2 (64->64 on 96x48x156)")
        del pre
        h3 = L["h3"](sp("pre", E["h2"]), E["h2"], *A["h3"], flags=ops.EPI_RELU, out_exp=E["h3"], overflow=flag)
        check(ops.from_split(h3, E["h3"]).cpu().numpy(), o["h3"].numpy(), TIGHT, "split hourglass conv3 (k3 stride 2, 64->64)")
        del h3
        h4 = L["h4"](sp("h3", E["h3"]), E["h3"], *A["h4"], flags=ops.EPI_RELU, out_exp=E["h4"], overflow=flag)
        check(ops.from_split(h4, E["h4"]).cpu().numpy(), o["h4"].numpy(), TIGHT, "split hourglass conv4 (64->64 on 48x24x78)")
        del h4
        post = L["h5"](sp("h4", E["h4"]), E["h4"], *A["h5"], residual=sp("pre", E["h2"]), flags=ops.EPI_RELU | ops.EPI_ADD_PRE,
                       out_exp=E["h2"], to_f32=True)
        check(post.cpu().numpy(), o["post"].numpy(), TIGHT, "split hourglass conv5 (transposed 64->64 + pre, ReLU) -> fp32")
        del post
        # r5: the same layer with its result contracted with the folded tail's taps in the epilogue, then the gather -- on the oracle's
        # h4 / pre / classifier(v2), against the oracle's COST (the stack's output), whole tensor
        t = L["h5"].forward_tail(sp("h4", E["h4"]), E["h4"], *A["h5"], st["tail"], residual=sp("pre", E["h2"]),
                                 flags=ops.EPI_RELU | ops.EPI_ADD_PRE, out_exp=E["h2"], overflow=flag)
        cost = ops.deconv_tail_gather(t, st["tail_bias"], o["hv"].to(dev()))
        check(cost.cpu().numpy(), o["cost"].numpy(), TIGHT, "conv5 with the tail projection + gather = classifier(v2 + conv6(post))")
        assert flag.item() == 0, "a value was clamped to half's range"
