"""Seeded synthetic KITTI label / result directories for the AP evaluator's tests.

Used by ``tests/golden/make_golden_eval.py`` (which runs the reference's prebuilt evaluator on them in the build container
and stores its tables) and by ``tests/test_kitti_eval.py`` (which re-draws the same directories from the same seeds).
Label format: ``type truncated occluded alpha x1 y1 x2 y2 h w l x y z ry [score]`` (KITTI object development kit).
"""
import os

import numpy as np

P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]])
IMG_W, IMG_H = 1242.0, 375.0
DIMS = {"Car": (1.5, 1.6, 3.9), "Van": (2.1, 1.9, 5.0), "Truck": (3.2, 2.6, 9.0), "Pedestrian": (1.75, 0.6, 0.8),
        "Person_sitting": (1.25, 0.6, 0.8), "Cyclist": (1.7, 0.6, 1.8), "Misc": (1.8, 1.5, 3.0), "Tram": (3.5, 2.5, 15.0)}
TYPES = ("Car", "Car", "Car", "Car", "Van", "Truck", "Pedestrian", "Pedestrian", "Person_sitting", "Cyclist", "Cyclist", "Misc")


def project_box(h, w, l, x, y, z, ry):
    """2D box of the 3D box's 8 projected corners (KITTI: y is the BOTTOM centre), clipped to the image; None if behind."""
    c, s = np.cos(ry), np.sin(ry)
    xs = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
    ys = np.array([0, 0, 0, 0, -h, -h, -h, -h], dtype=np.float64)
    zs = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
    cam = np.stack([c * xs + s * zs + x, ys + y, -s * xs + c * zs + z, np.ones(8)])
    if cam[2].min() < 0.5:
        return None
    uvw = P2 @ cam
    u, v = uvw[0] / uvw[2], uvw[1] / uvw[2]
    box = [max(u.min(), 0.0), max(v.min(), 0.0), min(u.max(), IMG_W - 1), min(v.max(), IMG_H - 1)]
    return box if box[2] - box[0] > 1 and box[3] - box[1] > 1 else None


def _alpha(x, z, ry):
    a = ry - np.arctan2(x, z)
    return (a + np.pi) % (2 * np.pi) - np.pi


def _gt_line(t, trunc, occ, box, dims, loc, ry):
    return "%s %.2f %d %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f" % (
        t, trunc, occ, _alpha(loc[0], loc[2], ry), box[0], box[1], box[2], box[3], dims[0], dims[1], dims[2], loc[0], loc[1], loc[2], ry)


def _det_line(t, alpha, box, dims, loc, ry, score):
    return "%s -1 -1 %.4f %.2f %.2f %.2f %.2f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f" % (
        t, alpha, box[0], box[1], box[2], box[3], dims[0], dims[1], dims[2], loc[0], loc[1], loc[2], ry, score)


def random_scene(seed, frames, indices=None, no_aos=False, only_2d=False, classes=None, noise=1.0):
    """{frame index: (gt lines, det lines)}: objects on a ground plane in front of the camera, detections = noisy copies
    (overlaps on both sides of the 0.7 / 0.5 thresholds), duplicates, false positives, small boxes, DontCare regions with
    detections inside them, neighbouring classes (Van, Person_sitting)."""
    r = np.random.default_rng(seed)
    out = {}
    indices = list(range(frames)) if indices is None else list(indices)
    for fi in indices:
        gts, dets = [], []
        for _ in range(int(r.integers(0, 9))):
            t = TYPES[int(r.integers(0, len(TYPES)))] if classes is None else classes[int(r.integers(0, len(classes)))]
            dims = np.array(DIMS[t]) * r.uniform(0.85, 1.15, 3)
            z = r.uniform(5, 70)
            x = r.uniform(-0.45, 0.45) * z
            loc = np.array([x, 1.65 + r.normal(0, 0.1), z])
            ry = r.uniform(-np.pi, np.pi)
            box = project_box(*dims, *loc, ry)
            if box is None:
                continue
            trunc = float(r.choice([0.0, 0.0, 0.0, 0.1, 0.2, 0.4, 0.7]))
            occ = int(r.choice([0, 0, 0, 1, 2, 3]))
            far = z > 55 and r.random() < 0.6
            if far:         # far objects are labelled DontCare (2D box only)
                gts.append("DontCare -1 -1 -10 %.2f %.2f %.2f %.2f -1 -1 -1 -1000 -1000 -1000 -10" % tuple(box))
            else:
                gts.append(_gt_line(t, trunc, occ, box, dims, loc, ry))
            if r.random() < 0.85:       # detected
                det_t = t if t in ("Car", "Pedestrian", "Cyclist") else {"Van": "Car", "Person_sitting": "Pedestrian"}.get(t, "Car")
                for _dup in range(1 + int(r.random() < 0.12)):
                    lvl = noise * float(r.choice([0.02, 0.05, 0.1, 0.2, 0.4]))
                    ddims = dims * (1 + r.normal(0, 0.5 * lvl, 3))
                    dloc = loc + r.normal(0, lvl, 3) * np.array([1.0, 0.3, 1.5])
                    dry = ry + r.normal(0, 0.5 * lvl) + (np.pi if r.random() < 0.08 else 0.0)
                    dbox = project_box(*ddims, *dloc, dry)
                    if dbox is None:
                        continue
                    dbox = [b + r.normal(0, 20 * lvl) for b in dbox]
                    dbox = [min(dbox[0], dbox[2]), min(dbox[1], dbox[3]), max(dbox[0], dbox[2]), max(dbox[1], dbox[3])]
                    score = float(np.round(r.uniform(0.05, 1.0), 3 if r.random() < 0.9 else 1))
                    alpha = _alpha(dloc[0], dloc[2], dry)
                    if only_2d:
                        dets.append(_det_line(det_t, alpha, dbox, (-1, -1, -1), (-1000, -1000, -1000), -10, score))
                    else:
                        dets.append(_det_line(det_t, alpha, dbox, ddims, dloc, dry, score))
        for _ in range(int(r.integers(0, 4))):       # false positives anywhere
            t = ("Car", "Pedestrian", "Cyclist")[int(r.integers(0, 3))] if classes is None else classes[0]
            dims = np.array(DIMS[t]) * r.uniform(0.8, 1.2, 3)
            z = r.uniform(4, 75)
            loc = np.array([r.uniform(-0.45, 0.45) * z, 1.65, z])
            ry = r.uniform(-np.pi, np.pi)
            box = project_box(*dims, *loc, ry)
            if box is None:
                continue
            score = float(np.round(r.uniform(0.0, 0.8), 3))
            if only_2d:
                dets.append(_det_line(t, _alpha(loc[0], loc[2], ry), box, (-1, -1, -1), (-1000, -1000, -1000), -10, score))
            else:
                dets.append(_det_line(t, _alpha(loc[0], loc[2], ry), box, dims, loc, ry, score))
        out[fi] = (gts, dets)
    if no_aos:      # one invalid orientation anywhere switches AOS off for the whole run (:156-157)
        for fi in indices:
            if out[fi][1]:
                f = out[fi][1][0].split()
                f[3] = "-10"
                out[fi][1][0] = " ".join(f)
                break
    return out


def handmade_scene():
    """Degenerate overlaps and tie cases, every number chosen by hand."""
    CAR = "Car 0.00 0 0.00 %s 1.50 1.60 4.00 %.2f 1.65 %.2f %.2f"        # (2D box, x, z, ry)
    g2d = "400.00 150.00 500.00 220.00"
    frames = {}
    # 0: identical box (overlap 1), plus an exact duplicate with the same score (one TP, one FP)
    frames[0] = ([CAR % (g2d, 0.0, 20.0, 0.0)],
                 [(CAR % (g2d, 0.0, 20.0, 0.0)).replace("Car 0.00 0", "Car -1 -1") + " 0.90"] * 2)
    # 1: detection rotated by 90 degrees about the same centre (BEV IoU = 1.6*1.6 / (2*6.4 - 2.56) = 0.25), same 2D box
    frames[1] = ([CAR % (g2d, 2.0, 15.0, 0.0)],
                 [(CAR % (g2d, 2.0, 15.0, 1.57)).replace("Car 0.00 0", "Car -1 -1") + " 0.80"])
    # 2: boxes touching along an edge (intersection of zero area) and a contained box (w, l halved: IoU 0.25)
    frames[2] = ([CAR % (g2d, -3.0, 25.0, 0.0), CAR % ("600.00 150.00 700.00 220.00", 4.0, 25.0, 0.0)],
                 [(CAR % (g2d, 1.0, 25.0, 0.0)).replace("Car 0.00 0", "Car -1 -1") + " 0.70",
                  "Car -1 -1 0.00 600.00 150.00 700.00 220.00 1.50 0.80 2.00 4.00 1.65 25.00 0.00 0.60"])
    # 3: two detections on one ground truth with equal scores and different overlaps; vertical offset kills the 3D overlap
    frames[3] = ([CAR % (g2d, 0.0, 30.0, 0.3)],
                 ["Car -1 -1 0.00 400.00 150.00 500.00 220.00 1.50 1.60 4.00 0.10 1.65 30.05 0.30 0.50",
                  "Car -1 -1 0.00 402.00 151.00 498.00 221.00 1.50 1.60 4.00 0.00 3.30 30.00 0.30 0.50"])
    # 4: a frame with ground truth and no detections at all, 5: detections and no ground truth
    frames[4] = ([CAR % (g2d, 1.0, 12.0, -1.0), "Pedestrian 0.00 0 0.10 700.00 140.00 730.00 230.00 1.80 0.60 0.80 3.00 1.65 14.00 0.20"], [])
    frames[5] = ([], ["Car -1 -1 0.00 100.00 150.00 200.00 220.00 1.50 1.60 4.00 -8.00 1.65 22.00 0.00 0.40",
                      "Pedestrian -1 -1 0.00 700.00 140.00 730.00 230.00 1.80 0.60 0.80 3.00 1.65 14.00 0.20 0.30"])
    # 6: a small (24 px) detection on a small ground truth; a detection inside a DontCare region; a Van matched by a Car detection
    frames[6] = (["Car 0.00 0 0.00 300.00 170.00 330.00 194.00 1.50 1.60 4.00 -6.00 1.65 45.00 0.00",
                  "DontCare -1 -1 -10 800.00 160.00 900.00 200.00 -1 -1 -1 -1000 -1000 -1000 -10",
                  "Van 0.00 0 0.00 50.00 140.00 180.00 240.00 2.10 1.90 5.00 -9.00 1.70 18.00 0.10"],
                 ["Car -1 -1 0.00 300.00 170.00 330.00 194.00 1.50 1.60 4.00 -6.00 1.65 45.00 0.00 0.95",
                  "Car -1 -1 0.00 810.00 162.00 880.00 198.00 1.50 1.60 4.00 9.00 1.65 60.00 0.00 0.85",
                  "Car -1 -1 0.00 52.00 141.00 179.00 239.00 2.00 1.85 4.90 -9.00 1.70 18.00 0.10 0.75"])
    # 7-12: plain matches at decreasing scores so that the recall sampling has something to sample
    for k in range(6):
        z = 10.0 + 4 * k
        frames[7 + k] = ([CAR % (g2d, -2.0 + k, z, 0.1 * k), "Cyclist 0.00 1 0.50 600.00 150.00 640.00 215.00 1.70 0.60 1.80 5.00 1.65 %.2f 0.50" % z],
                         ["Car -1 -1 0.00 401.00 150.00 499.00 221.00 1.52 1.58 3.95 %.2f 1.66 %.2f %.2f %.2f" % (-2.0 + k + 0.05, z + 0.1, 0.1 * k + 0.02, 0.99 - 0.1 * k),
                          "Cyclist -1 -1 0.45 601.00 151.00 640.00 214.00 1.70 0.60 1.80 5.02 1.65 %.2f 0.52 %.2f" % (z + 0.05, 0.9 - 0.12 * k)])
    return frames


SCENES = {
    "mixed": lambda: random_scene(11, 80),
    "no_aos": lambda: random_scene(12, 30, no_aos=True),
    "only_2d_cars": lambda: random_scene(13, 30, only_2d=True, classes=("Car", "Van")),
    "sparse_indices": lambda: random_scene(14, 0, indices=(3, 17, 18, 256, 257, 1999, 2000, 3712, 5000, 6001, 7480, 7517), noise=0.2),
    "coarse": lambda: random_scene(15, 40, noise=2.5),
    "clean": lambda: random_scene(16, 60, noise=0.25),
    "handmade": handmade_scene,
}


def write_scene(scene, gt_dir, result_dir, extra_gt=()):
    """label_2-style directory + a result directory with its ``data`` sub-directory (what the evaluator's command line takes)."""
    os.makedirs(gt_dir, exist_ok=True)
    os.makedirs(os.path.join(result_dir, "data"), exist_ok=True)
    for fi, (gts, dets) in scene.items():
        with open(os.path.join(gt_dir, "%06d.txt" % fi), "w") as fh:
            fh.write("".join(line + "\n" for line in gts))
        with open(os.path.join(result_dir, "data", "%06d.txt" % fi), "w") as fh:
            fh.write("".join(line + "\n" for line in dets))
    for fi in extra_gt:         # labels without a result file are not evaluated (README of tools/kitti-eval)
        with open(os.path.join(gt_dir, "%06d.txt" % fi), "w") as fh:
            fh.write("Car 0.00 0 0.00 100.00 100.00 200.00 200.00 1.50 1.60 4.00 0.00 1.65 20.00 0.00\n")
