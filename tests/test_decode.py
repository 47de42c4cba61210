"""N4: output decode + KITTI label writer against golden outputs of the imported reference
(tests/golden/make_golden_decode.py -> tests/golden/decode_outputs.npz).  Host math: runs without a GPU; the
device read-out (argmax / range test on the GPU) is checked in the -m gpu test at the bottom."""
import os
import types

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode_outputs.npz")


def decode_case(name):
    """Seeded inputs of one decode case (shared with the golden generator)."""
    r = np.random.default_rng({"argmax": 1, "coordinates": 2, "one_part_only": 3}[name])
    n, parts = 6, (1 if name == "one_part_only" else 9)
    nl, nw = 24, 16
    cfg = types.SimpleNamespace(x_range=(-1.6, 1.6), z_range=(-2.4, 2.4))
    ncf = r.uniform(0.0, 1.0, (n, parts, nl, nw)).astype(np.float32)
    ncf[2, 0, 3, 3] = 2.5          # instance 2 fails the Filter (a value above max_val)
    ncf[4, -1, 0, 0] = -1.5        # instance 4 too (below min_val)
    samples = np.stack([np.array([1.5 + 0.1 * r.random(), 1.6 + 0.1 * r.random(), 3.8 + 0.5 * r.random(),
                                  r.uniform(-10, 10), 1.6 + 0.2 * r.random(), r.uniform(6, 50), r.uniform(-np.pi, np.pi)])
                        for _ in range(n)])
    zs, xs = np.meshgrid(np.linspace(cfg.z_range[0], cfg.z_range[1], nl), np.linspace(cfg.x_range[0], cfg.x_range[1], nw), indexing="ij")
    grid = np.stack([xs.ravel(), r.uniform(-0.5, 0.5, nl * nw), zs.ravel()], axis=1)      # [nl*nw, 3]; y is dropped by the decode
    coordinates = r.uniform(0.05, 0.95, (n, parts, 2)) if name == "coordinates" else None
    meta = {"lp": [os.path.join("kitti", "image_2", f"{i // 2:06d}.png") for i in range(n)],
            "box2d": r.uniform(0, 1200, (n, 4)), "score": r.uniform(0.1, 1.0, n)}
    return dict(cfg=cfg, ncf=ncf, samples=samples, grid=grid, coordinates=coordinates, meta=meta)


@pytest.mark.parametrize("name", ["argmax", "coordinates", "one_part_only"])
def test_decode_vs_reference_golden(name):
    from snvc_amd import decode as D
    G = np.load(GOLDEN)
    c = decode_case(name)
    res = D.ncf_to_update_2d(c["cfg"], c["ncf"], c["samples"].copy(), c["grid"].copy(), D.Filter(), coordinates=c["coordinates"])
    assert np.array_equal(res["keep_flags"], G[f"{name}/keep_flags"]) and res["keep_flags"].sum() == 4
    assert np.array_equal(res["confidence"], G[f"{name}/confidence"])
    for k, v in res["pred"].items():
        exp = G[f"{name}/pred_{k}"]
        got = np.asarray(v, dtype=np.float64)
        assert got.shape == exp.shape, (k, got.shape, exp.shape)
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-9, err_msg=f"{name}/{k}")
    if "all_parts" in res["pred"]:
        record = {}
        D.update_record(record, res, c["meta"])
        lines = []
        for fname in sorted(record):
            lines += [fname] + record[fname]["all_parts"]
        assert "\n".join(lines) == str(G[f"{name}/kitti_lines"])


def test_kitti_writer_files(tmp_path):
    from snvc_amd import decode as D
    G = np.load(GOLDEN)
    got = [D.roty2alpha(x, z, r) for x, z, r in [(1.0, 10.0, 0.3), (-5.0, 20.0, -3.0), (3.0, 8.0, 3.1), (0.0, 5.0, -1.6)]]
    np.testing.assert_allclose(got, G["roty2alpha"], rtol=0, atol=1e-12)
    c = decode_case("argmax")
    res = D.ncf_to_update_2d(c["cfg"], c["ncf"], c["samples"], c["grid"], D.Filter())
    record = {}
    D.update_record(record, res, c["meta"])
    cfg = types.SimpleNamespace(output_dir=str(tmp_path / "out"), pred_type=["all_parts"])
    calib = tmp_path / "data" / "calib"
    calib.mkdir(parents=True)
    for f in ("000000.txt", "000001.txt", "000002.txt", "000007.txt"):
        (calib / f).write_text("")
    D.generate_output(record, cfg, split_file="test.txt", data_path=str(tmp_path / "data"))
    folder = tmp_path / "out" / "all_parts" / "data"
    assert sorted(os.listdir(folder)) == ["000000.txt", "000001.txt", "000002.txt", "000007.txt"]
    assert (folder / "000007.txt").read_text() == ""                       # a frame without predictions
    first = (folder / "000000.txt").read_text().split("\n")
    assert len(first) == 2 and all(len(line.split(" ")) == 16 and line.startswith("Car -1.0 -1.0 ") for line in first)


@pytest.mark.gpu
def test_decode_device_readout_equals_host():
    """max / argmax / range test on the device == the host read-out (same dict)."""
    from snvc_amd import decode as D
    c = decode_case("argmax")
    host = D.ncf_to_update_2d(c["cfg"], c["ncf"], c["samples"].copy(), c["grid"], D.Filter())
    dev = D.ncf_to_update_2d(c["cfg"], torch.from_numpy(c["ncf"]).cuda(), c["samples"].copy(), c["grid"], D.Filter())
    assert np.array_equal(host["keep_flags"], dev["keep_flags"]) and np.array_equal(host["confidence"], dev["confidence"])
    for k in host["pred"]:
        assert np.array_equal(np.asarray(host["pred"][k]), np.asarray(dev["pred"][k]))
