"""KITTI object-detection AP / AOS evaluation: what follows ``decode`` + the label writer when
a result directory is scored against ``label_2`` (reference docs/INFERENCE.md:26-34).

Replaces the reference's standalone C++ tool ``tools/kitti-eval/evaluate_object_3d_offline.cpp`` /
``evaluate_object_3d_offline_r40.cpp`` (needs Boost.Geometry): same command-line meaning
(``evaluate(gt_dir, result_dir)``: only frames with a result file are scored, ``:778-793``), same tables -- image /
ground-plane / 3D average precision and the image orientation similarity for car / pedestrian / cyclist at the easy /
moderate / hard filters, 41 recall points -- and the same ``<table> AP: easy moderate hard`` report lines.  The matching
core is ``snvc_kitti_eval`` in ``libsnvc_hip.so`` (csrc/kitti_eval.hip: overlap matrices once per frame, convex-quad
clipping instead of Boost, the 27 sweeps on a thread pool); this module owns what the tool does around it: directory
listing, label parsing, the "which tables" flags (``loadDetections`` ``:131-176``) and the AP read-out (``:719-723``).
Not kept: gnuplot / pdf plots and the mail stub.  Pinned by ``tests/test_kitti_eval.py`` against tables produced by the
reference's own prebuilt binary.
"""
import ctypes
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

CLASS_NAMES = ("car", "pedestrian", "cyclist")                  # :61-65
DIFFICULTIES = ("easy", "moderate", "hard")                     # :37
N_SAMPLE_PTS = 41                                               # :58
# MIN_OVERLAP[metric][class] of the shipped tool (:55): one row per metric (image, ground, 3D), car / pedestrian / cyclist
MIN_OVERLAP = ((0.7, 0.5, 0.5), (0.7, 0.5, 0.5), (0.7, 0.5, 0.5))
_TYPE_CODES = {"car": 0, "pedestrian": 1, "cyclist": 2, "van": 3, "person_sitting": 4, "dontcare": 5}
_SUFFIX = ("_detection", "_detection_ground", "_detection_3d")
GT_COLS, DET_COLS = 14, 13


def _type_code(name: str) -> int:
    return _TYPE_CODES.get(name.lower(), 6)                      # strcasecmp everywhere in the tool


def parse_label_file(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """label_2 file -> (rows [G,14] float64: truncation, occlusion, alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry; type codes [G]).
    Like the tool's ``fscanf`` loop (``:187-198``) the file is read as a token stream, 15 tokens per record."""
    with open(path) as fh:
        tok = fh.read().split()
    n = len(tok) // 15
    rows = np.empty((n, GT_COLS), dtype=np.float64)
    types = np.empty(n, dtype=np.int32)
    for i in range(n):
        f = tok[15 * i:15 * i + 15]
        types[i] = _type_code(f[0])
        rows[i, 0] = float(f[1])
        rows[i, 1] = int(f[2])                                    # %d
        rows[i, 2:] = [float(v) for v in f[3:]]
    return rows, types


def parse_result_file(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """result file -> (rows [M,13] float64: alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry, score; type codes [M]); 16 tokens per
    record, the two after the type are ignored (``:146-149``)."""
    with open(path) as fh:
        tok = fh.read().split()
    n = len(tok) // 16
    rows = np.empty((n, DET_COLS), dtype=np.float64)
    types = np.empty(n, dtype=np.int32)
    for i in range(n):
        f = tok[16 * i:16 * i + 16]
        types[i] = _type_code(f[0])
        rows[i] = [float(v) for v in f[3:]]
    types[types > 2] = 6                                          # only the three evaluated classes mean anything for a detection
    return rows, types


def eval_indices(result_data_dir: str) -> List[int]:
    """Frame numbers of the result files (``getEvalIndices`` ``:778-793``: names of at least 10 characters, number = the
    leading integer of the last 10)."""
    out = []
    for name in sorted(os.listdir(result_data_dir)):
        if len(name) < 10:
            continue
        tail = name[-10:].lstrip()
        k = 1 if tail[:1] in "+-" else 0
        while k < len(tail) and tail[k].isdigit():
            k += 1
        digits = tail[:k]
        out.append(int(digits) if digits not in ("", "+", "-") else 0)
    return out


def _ap(curve: np.ndarray, points: Iterable[int], denom: int) -> float:
    s = np.float32(0)                                             # the tool sums in a float (:719-723)
    for i in points:
        s = np.float32(s + np.float32(curve[i]))
    return float(np.float32(np.float32(s / np.float32(denom)) * np.float32(100)))


def ap_r40(curve) -> float:
    """evaluate_object_3d_offline_r40.cpp:721-723: mean of recall points 1..40, in percent."""
    return _ap(np.asarray(curve), range(1, N_SAMPLE_PTS), 40)


def ap_r11(curve) -> float:
    """evaluate_object_3d_offline.cpp:721-723: mean of recall points 0, 4, ..., 40, in percent."""
    return _ap(np.asarray(curve), range(0, N_SAMPLE_PTS, 4), 11)


def evaluate_frames(gt: Sequence[Tuple[np.ndarray, np.ndarray]], det: Sequence[Tuple[np.ndarray, np.ndarray]],
                    min_overlap=MIN_OVERLAP, threads: int = 0) -> Dict[str, dict]:
    """Scores parsed frames: ``gt[k]`` / ``det[k]`` = (rows, type codes) of frame k (``parse_label_file`` /
    ``parse_result_file`` layouts).  Returns {table name: {"curve": [3 difficulties][41], "ap_r40": [3], "ap_r11": [3]}} for
    the tables the detections enable -- a class is scored on the image when it has a detection with x1 >= 0, on the ground
    plane / in 3D when one has x != -1000 / y != -1000; orientation tables only if no detection carries alpha == -10
    (``:155-170``)."""
    if len(gt) != len(det):
        raise ValueError("one ground-truth entry per detection entry")
    frames = len(gt)
    cat = lambda parts, cols: (np.ascontiguousarray(np.concatenate([p[0].reshape(-1, cols) for p in parts]), dtype=np.float64)      # noqa: E731
                               if parts else np.zeros((0, cols)))
    g_rows, d_rows = cat(gt, GT_COLS), cat(det, DET_COLS)
    g_type = np.ascontiguousarray(np.concatenate([p[1] for p in gt]) if frames else np.zeros(0), dtype=np.int32)
    d_type = np.ascontiguousarray(np.concatenate([p[1] for p in det]) if frames else np.zeros(0), dtype=np.int32)
    g_off = np.zeros(frames + 1, dtype=np.int64)
    d_off = np.zeros(frames + 1, dtype=np.int64)
    g_off[1:] = np.cumsum([len(p[1]) for p in gt])
    d_off[1:] = np.cumsum([len(p[1]) for p in det])
    flags = np.zeros((3, 3), dtype=np.int32)
    for c in range(3):
        sel = d_type == c
        flags[0, c] = bool(np.any(d_rows[sel, 1] >= 0))           # x1
        flags[1, c] = bool(np.any(d_rows[sel, 8] != -1000))       # x
        flags[2, c] = bool(np.any(d_rows[sel, 9] != -1000))       # y
    compute_aos = not bool(np.any(d_rows[:, 0] == -10))
    precision = np.zeros((3, 3, 3, N_SAMPLE_PTS), dtype=np.float64)
    aos = np.zeros((3, 3, N_SAMPLE_PTS), dtype=np.float64)
    mo = np.ascontiguousarray(min_overlap, dtype=np.float64)
    if mo.shape != (3, 3):
        raise ValueError("min_overlap is [3 metrics][3 classes]")
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)             # noqa: E731
    _lib.check(_lib.lib().snvc_kitti_eval(ptr(g_rows), ptr(g_type), ptr(g_off), ptr(d_rows), ptr(d_type), ptr(d_off), frames,
                                          ptr(mo), ptr(flags), int(compute_aos), ptr(precision), ptr(aos), int(threads)),
               "snvc_kitti_eval")
    out = {}

    def table(curves):
        return {"curve": curves.copy(), "ap_r40": [ap_r40(c) for c in curves], "ap_r11": [ap_r11(c) for c in curves]}
    for m in range(3):
        for c in range(3):
            if not flags[m, c]:
                continue
            out[CLASS_NAMES[c] + _SUFFIX[m]] = table(precision[m, c])
            if m == 0 and compute_aos:
                out[CLASS_NAMES[c] + "_orientation"] = table(aos[c])
    return out


def evaluate(gt_dir: str, result_dir: str, write: bool = False, threads: int = 0) -> Dict[str, dict]:
    """``./evaluate_object_3d_offline gt_dir result_dir``: scores ``result_dir/data/*.txt`` against ``gt_dir/<same name>``.
    ``write=True`` also leaves the tool's text outputs: ``result_dir/stats_<class>_<detection|orientation|detection_ground|
    detection_3d>.txt`` (one line of 41 values per difficulty, ``saveStats`` ``:204-219``) and ``result_dir/plot/<table>.txt``
    (recall, easy, moderate, hard per line, ``:713-717``)."""
    data_dir = os.path.join(result_dir, "data")
    gt, det = [], []
    for idx in eval_indices(data_dir):
        name = "%06d.txt" % idx
        gt_path = os.path.join(gt_dir, name)
        if not os.path.exists(gt_path):
            raise FileNotFoundError(f"ERROR: Couldn't read: {name} of ground truth")      # the tool's message (:839-842)
        gt.append(parse_label_file(gt_path))
        det.append(parse_result_file(os.path.join(data_dir, name)))
    res = evaluate_frames(gt, det, threads=threads)
    if write:
        os.makedirs(os.path.join(result_dir, "plot"), exist_ok=True)
        for name, tab in res.items():
            cls, kind = name.split("_", 1)
            with open(os.path.join(result_dir, f"stats_{cls}_{kind}.txt"), "w") as fh:
                for row in tab["curve"]:
                    fh.write("".join("%f " % v for v in row) + "\n")
            with open(os.path.join(result_dir, "plot", name + ".txt"), "w") as fh:
                for i in range(N_SAMPLE_PTS):
                    fh.write("%f %f %f %f\n" % (i / (N_SAMPLE_PTS - 1.0), tab["curve"][0][i], tab["curve"][1][i], tab["curve"][2][i]))
    return res


def report(res: Dict[str, dict], r40: bool = True) -> str:
    """The tool's ``<table> AP: easy moderate hard`` lines, in its order (image tables with their orientation table class by
    class, then ground, then 3D)."""
    key = "ap_r40" if r40 else "ap_r11"
    order = [CLASS_NAMES[c] + s for c in range(3) for s in ("_detection", "_orientation")]
    order += [CLASS_NAMES[c] + "_detection_ground" for c in range(3)] + [CLASS_NAMES[c] + "_detection_3d" for c in range(3)]
    return "\n".join("%s AP: %f %f %f" % ((n,) + tuple(res[n][key])) for n in order if n in res)
