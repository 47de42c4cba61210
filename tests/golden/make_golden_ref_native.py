#!/usr/bin/env python3
"""Golden vectors of `points_in_boxes_cpu` from the REFERENCE'S OWN compiled C++ (oracle/_ref, built by `make -C oracle ref` from
/root/reference/snvc/extension/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168 where it lies).  Run where /root/reference exists:
    make -C oracle ref && python tests/golden/make_golden_ref_native.py
Writes tests/golden/points_in_boxes_cpu_ref.npz: inputs + the reference's outputs (data only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_native as R  # noqa: E402


def cases():
    rng = np.random.default_rng(20261004)
    out = {}
    # (a) random rotated boxes in a KITTI-sized scene, many points near them
    boxes = np.stack([rng.uniform(-20, 20, 64), rng.uniform(0, 60, 64), rng.uniform(-2, 1, 64), rng.uniform(1.2, 5, 64),
                      rng.uniform(0.5, 2.2, 64), rng.uniform(1.2, 2.2, 64), rng.uniform(-np.pi, np.pi, 64)], 1).astype(np.float32)
    centre = boxes[rng.integers(0, 64, 6000), :3]
    pts = (centre + rng.normal(0, 1.4, (6000, 3))).astype(np.float32)
    out["a"] = (pts, boxes)
    # (b) points ON the faces and inside / outside the 1e-2 margin of axis-aligned and quarter-turn boxes
    b = np.array([[0, 0, 0, 2, 4, 2, 0], [10, 0, 0, 2, 4, 2, np.pi / 2], [0, 10, 1, 3, 1, 2, -np.pi / 2], [5, 5, 5, 1, 1, 1, np.pi]], np.float32)
    offs = np.array([0.0, 1e-3, 9e-3, 1e-2, 1.1e-2, 2e-2, -1e-3, -1e-2], np.float32)
    p = []
    for bx in b:
        for ax in range(3):
            for sgn in (-1.0, 1.0):
                for o in offs:
                    q = bx[:3].copy()
                    q[ax] += sgn * (bx[3 + ax] / 2 + o)
                    p.append(q)
                    q2 = bx[:3].copy()      # quarter-turn boxes: the same offsets along the swapped axis
                    q2[(ax + 1) % 3] += sgn * (bx[3 + ax] / 2 + o)
                    p.append(q2)
    out["b"] = (np.array(p, np.float32), b)
    # (c) degenerate: zero-size box, huge heading, empty inputs
    out["c"] = (rng.normal(0, 0.02, (50, 3)).astype(np.float32),
                np.array([[0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 1, 1, 1, 1000.0], [0, 0, 0, 0.01, 0.01, 0.01, -7.5]], np.float32))
    out["d"] = (np.zeros((0, 3), np.float32), boxes[:3])
    out["e"] = (pts[:5], np.zeros((0, 7), np.float32))
    return out


def main():
    if not R.available():
        sys.exit("oracle/_ref/snvc_ref_roiaware.so missing: run `make -C oracle ref`")
    blob = {}
    for k, (pts, boxes) in cases().items():
        blob[f"{k}_pts"], blob[f"{k}_boxes"] = pts, boxes
        blob[f"{k}_flags"] = R.points_in_boxes_cpu(pts, boxes)
        print(k, pts.shape, boxes.shape, "flags set:", int(blob[f"{k}_flags"].sum()))
    path = os.path.join(ROOT, "tests", "golden", "points_in_boxes_cpu_ref.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
