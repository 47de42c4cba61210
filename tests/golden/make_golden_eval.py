"""Generates tests/golden/kitti_eval_outputs.npz by RUNNING THE REFERENCE'S OWN PREBUILT EVALUATOR (this container only).

Run:  python tests/golden/make_golden_eval.py

Pins SURVEY.md 8f row N4's AP evaluator: ``tools/kitti-eval/evaluate_object_3d_offline_r40`` (an x86-64 ELF shipped in the
reference, linked against libstdc++ / libm / libc only; its source ``evaluate_object_3d_offline_r40.cpp`` needs Boost, which
this image does not have, so it cannot be rebuilt here) is run on the seeded synthetic label / result directories of
``tests/kitti_eval_cases.py`` in a scratch directory.  Stored: for every scene and every table the binary writes
(``plot/<class>_<detection|orientation|detection_ground|detection_3d>.txt``: 41 recall points x easy / moderate / hard, 6
decimals) the table itself, and the ``AP:`` lines of its standard output.  Only the binary's OUTPUTS are stored; the
inputs are re-drawn from seeds by the tests.  (gnuplot / ps2pdf are absent: the plots the binary also tries to make fail
harmlessly.)
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.environ.get("SNVC_REFERENCE", "/root/reference")
BINARY = os.path.join(REF, "tools", "kitti-eval", "evaluate_object_3d_offline_r40")

from kitti_eval_cases import SCENES, write_scene  # noqa: E402

out = {}
for name, make in SCENES.items():
    with tempfile.TemporaryDirectory(prefix="snvc_eval_") as tmp:
        gt_dir, res_dir = os.path.join(tmp, "label_2"), os.path.join(tmp, "result")
        write_scene(make(), gt_dir, res_dir, extra_gt=(900001,) if name == "sparse_indices" else ())
        run = subprocess.run([BINARY, gt_dir, res_dir], capture_output=True, text=True, timeout=600, cwd=tmp)
        assert run.returncode == 0, run.stderr
        tables = sorted(f[:-4] for f in os.listdir(os.path.join(res_dir, "plot")) if f.endswith(".txt"))
        for t in tables:
            tab = np.loadtxt(os.path.join(res_dir, "plot", t + ".txt"))
            assert tab.shape == (41, 4)
            out[f"{name}/{t}/curve"] = tab[:, 1:].T.copy()                       # [easy | moderate | hard][41]
        aps = re.findall(r"^(\w+) AP: (\S+) (\S+) (\S+)$", run.stdout, re.M)
        assert sorted(a[0] for a in aps) == tables, (aps, tables)
        for a in aps:
            out[f"{name}/{a[0]}/ap"] = np.array([float(x) for x in a[1:]])
        out[f"{name}/tables"] = np.array(" ".join(tables))
        print(name, {a[0]: a[1:] for a in aps})
path = os.path.join(HERE, "kitti_eval_outputs.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {len(out)} arrays")
