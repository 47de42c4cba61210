"""KITTI AP / AOS evaluator: the oracle restatement and the product against tables produced by the
REFERENCE'S OWN prebuilt binary (tests/golden/kitti_eval_outputs.npz, made by tests/golden/make_golden_eval.py from
tools/kitti-eval/evaluate_object_3d_offline_r40 on the seeded directories of tests/kitti_eval_cases.py), and the product
against the oracle beyond the binary's 6 printed decimals.  Host code only (the reference's evaluator is a host program
too): runs without a GPU."""
import os

import numpy as np
import pytest

from kitti_eval_cases import SCENES, random_scene, write_scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_eval_outputs.npz")
PRINTED = 5.1e-7        # the binary prints its tables with %f: half a unit of the 6th decimal
AP_TOL = 1e-4           # "match the fixtures to 1e-4 AP" (percent)


@pytest.fixture(scope="module")
def G():
    return np.load(GOLDEN)


def _dirs(tmp_path, name):
    gt_dir, res_dir = str(tmp_path / "label_2"), str(tmp_path / "result")
    write_scene(SCENES[name](), gt_dir, res_dir, extra_gt=(900001,) if name == "sparse_indices" else ())
    return gt_dir, res_dir


@pytest.mark.parametrize("name", list(SCENES))
def test_oracle_reproduces_the_reference_binary(name, G, tmp_path):
    from oracle import kitti_eval_ref as K
    res = K.evaluate(*_dirs(tmp_path, name))
    tables = str(G[f"{name}/tables"]).split()
    assert sorted(res) == tables
    for t in tables:
        assert np.abs(np.array(res[t]["precision"]) - G[f"{name}/{t}/curve"]).max() <= PRINTED, t
        assert np.abs(np.array(res[t]["ap"]) - G[f"{name}/{t}/ap"]).max() <= AP_TOL, t


@pytest.mark.parametrize("name", list(SCENES))
def test_product_reproduces_the_reference_binary_and_the_oracle(name, G, tmp_path):
    from oracle import kitti_eval_ref as K
    from snvc_amd import evaluate as E
    gt_dir, res_dir = _dirs(tmp_path, name)
    res = E.evaluate(gt_dir, res_dir, write=True)
    tables = str(G[f"{name}/tables"]).split()
    assert sorted(res) == tables                                   # same set of tables as the binary enabled
    ref = K.evaluate(gt_dir, res_dir)
    for t in tables:
        assert np.abs(res[t]["curve"] - G[f"{name}/{t}/curve"]).max() <= PRINTED, t
        assert np.abs(np.array(res[t]["ap_r40"]) - G[f"{name}/{t}/ap"]).max() <= AP_TOL, t
        # against the oracle in full double precision: same matching decisions, same quotients
        assert np.abs(res[t]["curve"] - np.array(ref[t]["precision"])).max() <= 1e-12, t
        assert np.allclose(res[t]["ap_r40"], ref[t]["ap"], rtol=0, atol=1e-5), t
        # the text outputs the tool leaves behind parse back to the same numbers
        tab = np.loadtxt(os.path.join(res_dir, "plot", t + ".txt"))
        assert tab.shape == (41, 4) and np.abs(tab[:, 1:].T - G[f"{name}/{t}/curve"]).max() <= 2 * PRINTED
        cls, kind = t.split("_", 1)
        stats = np.loadtxt(os.path.join(res_dir, f"stats_{cls}_{kind}.txt"))
        assert stats.shape == (3, 41)
    # the report lines, in the binary's order and format
    lines = E.report(res).splitlines()
    assert [ln.split(" AP: ")[0] for ln in lines] == [t for t in _binary_order() if t in tables]
    for ln in lines:
        t, vals = ln.split(" AP: ")
        assert np.abs(np.array([float(v) for v in vals.split()]) - G[f"{name}/{t}/ap"]).max() <= AP_TOL


def _binary_order():
    cls = ("car", "pedestrian", "cyclist")
    return [c + s for c in cls for s in ("_detection", "_orientation")] + [c + "_detection_ground" for c in cls] + \
           [c + "_detection_3d" for c in cls]


def test_product_vs_oracle_on_a_larger_directory(tmp_path):
    """400 frames / ~1700 detections, thread pool on: every table equal to the oracle's to 1e-12."""
    from oracle import kitti_eval_ref as K
    from snvc_amd import evaluate as E
    gt_dir, res_dir = str(tmp_path / "label_2"), str(tmp_path / "result")
    write_scene(random_scene(77, 400, noise=0.5), gt_dir, res_dir)
    res, ref = E.evaluate(gt_dir, res_dir, threads=4), K.evaluate(gt_dir, res_dir)
    assert sorted(res) == sorted(ref) and len(res) == 12
    for t in res:
        assert np.abs(res[t]["curve"] - np.array(ref[t]["precision"])).max() <= 1e-12, t
    one = E.evaluate(gt_dir, res_dir, threads=1)                   # the pool changes nothing
    for t in res:
        assert np.array_equal(res[t]["curve"], one[t]["curve"])
    assert max(max(v["ap_r40"]) for v in res.values()) > 30.0      # a directory that actually exercises the recall sampling


def test_rotated_overlaps_known_answers():
    """The convex-quad clip that replaces Boost.Geometry, through the C ABI: a single ground truth / detection pair whose
    precision table is 1 exactly when the overlap clears the threshold passed in."""
    from snvc_amd import evaluate as E

    def passes(det_box, thr, metric):
        g = (np.array([[0, 0, 0, 100, 100, 200, 200, 1.5, 1.6, 4.0, 0.0, 1.65, 20.0, 0.0]], dtype=np.float64), np.array([0], np.int32))
        d = (np.array([[0.0, 100, 100, 200, 200] + list(det_box) + [0.9]], dtype=np.float64), np.array([0], np.int32))
        mo = np.full((3, 3), 0.01)
        mo[metric, 0] = thr
        res = E.evaluate_frames([g], [d], min_overlap=mo)
        return res["car" + ("_detection", "_detection_ground", "_detection_3d")[metric]]["curve"][0][0] == 1.0

    same = [1.5, 1.6, 4.0, 0.0, 1.65, 20.0, 0.0]
    assert passes(same, 0.999999, 1) and passes(same, 0.999999, 2)
    quarter_turn = [1.5, 1.6, 4.0, 0.0, 1.65, 20.0, np.pi / 2]       # BEV IoU = 2.56 / (12.8 - 2.56) = 0.25
    assert passes(quarter_turn, 0.2499, 1) and not passes(quarter_turn, 0.2501, 1)
    shifted = [1.5, 1.6, 4.0, 2.0, 1.65, 20.0, 0.0]                  # half the length along x: IoU = 3.2 / 9.6 = 1/3
    assert passes(shifted, 0.3333, 1) and not passes(shifted, 0.3334, 1)
    raised = [1.5, 1.6, 4.0, 0.0, 1.65 - 0.75, 20.0, 0.0]            # half the height: 3D IoU = 1/3, BEV IoU = 1
    assert passes(raised, 0.3333, 2) and not passes(raised, 0.3334, 2) and passes(raised, 0.999999, 1)
    diag = [1.5, 2.0, 2.0, 0.0, 1.65, 20.0, np.pi / 4]               # a 2x2 square turned by 45 degrees inside a 1.6 x 4.0 box
    inter = 4.0 - 2 * (np.sqrt(2.0) - 0.8) ** 2                      # two corner triangles cut off by |z| <= 0.8
    iou = inter / (6.4 + 4.0 - inter)
    assert passes(diag, iou - 1e-9, 1) and not passes(diag, iou + 1e-9, 1)
    touching = [1.5, 1.6, 4.0, 4.0, 1.65, 20.0, 0.0]                 # shares an edge: zero overlap
    assert not passes(touching, 1e-12, 1)


def test_r11_readout_and_empty_inputs():
    from snvc_amd import evaluate as E
    curve = np.linspace(1.0, 0.0, 41)
    assert abs(E.ap_r11(curve) - 100 * curve[::4].mean()) < 1e-4 and abs(E.ap_r40(curve) - 100 * curve[1:].mean()) < 1e-4
    assert E.evaluate_frames([], []) == {}
    empty = (np.zeros((0, 14)), np.zeros(0, np.int32))
    det = (np.array([[0.0, 10, 10, 90, 90, 1.5, 1.6, 4.0, 0.0, 1.65, 20.0, 0.0, 0.5]]), np.array([0], np.int32))
    res = E.evaluate_frames([empty], [det])                        # detections without any ground truth: all-zero tables
    assert sorted(res) == ["car_detection", "car_detection_3d", "car_detection_ground", "car_orientation"]
    assert all(not v["curve"].any() for v in res.values())
