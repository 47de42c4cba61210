"""CPU oracle for the KITTI object AP / AOS evaluator -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, function by function and in the reference's own loop order, ``tools/kitti-eval/evaluate_object_3d_offline_r40.cpp``
(the R40 variant of ``evaluate_object_3d_offline.cpp``; the two differ only in the AP read-out, ``:719-723``).  Plain Python
loops: small cases only.  The product is ``snvc_amd/evaluate.py`` over ``csrc/kitti_eval.hip`` -- a different design
(overlap matrices computed once, C++); this file is its checker.

Pinned: ``tests/golden/make_golden_eval.py`` runs the reference's OWN prebuilt binary
(``tools/kitti-eval/evaluate_object_3d_offline_r40``, links only libstdc++ / libm / libc) on seeded synthetic label /
result directories in the build container and stores its precision / AOS tables and AP lines in
``tests/golden/kitti_eval_outputs.npz``; ``tests/test_kitti_eval.py`` requires this restatement to reproduce them.

Third-party arithmetic: the rotated-box overlaps go through Boost.Geometry (``intersection`` / ``union_`` / ``area`` of two
4-corner polygons, ``:267-344``), which is not in /root/reference and not in this image (version unknown: whatever the
binary was linked against).  Restated from its published meaning: the intersection of two convex polygons by
Sutherland-Hodgman clipping, area by the shoelace formula, union area = area(a) + area(b) - area(intersection) when the
intersection is non-empty (two convex polygons that overlap have a hole-free union of exactly that area).
"""
import math
import os

import numpy as np

N_SAMPLE_PTS = 41                                        # :58
MIN_HEIGHT = (40, 25, 25)                                # :43
MAX_OCCLUSION = (0, 1, 2)                                # :44
MAX_TRUNCATION = (0.15, 0.3, 0.5)                        # :45
CLASS_NAMES = ("car", "pedestrian", "cyclist")           # :61-65
MIN_OVERLAP = ((0.7, 0.5, 0.5), (0.7, 0.5, 0.5), (0.7, 0.5, 0.5))    # :55  [metric][class]
IMAGE, GROUND, BOX3D = 0, 1, 2                           # :40
NO_DETECTION = -10000000.0                               # :463


class Box:
    """tGroundtruth / tDetection (:83-123) in one record."""
    __slots__ = ("type", "x1", "y1", "x2", "y2", "alpha", "truncation", "occlusion", "h", "w", "l", "t1", "t2", "t3", "ry", "thresh")


def load_detections(path, state):
    """loadDetections :131-176.  ``state`` = dict(compute_aos, eval_image, eval_ground, eval_3d)."""
    dets = []
    with open(path) as fh:
        tok = fh.read().split()
    i = 0
    while i + 16 <= len(tok):                            # fscanf of 16 fields per record (:146-149)
        f = tok[i:i + 16]
        i += 16
        d = Box()
        d.type = f[0]
        d.alpha, d.x1, d.y1, d.x2, d.y2, d.h, d.w, d.l, d.t1, d.t2, d.t3, d.ry, d.thresh = (float(v) for v in f[3:16])
        dets.append(d)
        if d.alpha == -10:
            state["compute_aos"] = False                 # :156-157
        for c in range(3):                               # :160-170
            if d.type.lower() == CLASS_NAMES[c]:
                if d.x1 >= 0:
                    state["eval_image"][c] = True
                if d.t1 != -1000:
                    state["eval_ground"][c] = True
                if d.t2 != -1000:
                    state["eval_3d"][c] = True
                break
    return dets


def load_groundtruth(path):
    """loadGroundtruth :178-202."""
    gts = []
    with open(path) as fh:
        tok = fh.read().split()
    i = 0
    while i + 15 <= len(tok):
        f = tok[i:i + 15]
        i += 15
        g = Box()
        g.type = f[0]
        g.truncation = float(f[1])
        g.occlusion = int(f[2])
        g.alpha, g.x1, g.y1, g.x2, g.y2, g.h, g.w, g.l, g.t1, g.t2, g.t3, g.ry = (float(v) for v in f[3:15])
        gts.append(g)
    return gts


def _div(a, b):
    """C double division (0/0 = NaN, x/0 = +-inf): comparisons with NaN are then false, as in the reference."""
    with np.errstate(all="ignore"):
        return float(np.float64(a) / np.float64(b))


def image_box_overlap(a, b, criterion=-1):
    """imageBoxOverlap :227-265."""
    x1, y1 = max(a.x1, b.x1), max(a.y1, b.y1)
    x2, y2 = min(a.x2, b.x2), min(a.y2, b.y2)
    w, h = x2 - x1, y2 - y1
    if w <= 0 or h <= 0:
        return 0.0
    inter = w * h
    a_area = (a.x2 - a.x1) * (a.y2 - a.y1)
    b_area = (b.x2 - b.x1) * (b.y2 - b.y1)
    if criterion == -1:
        return _div(inter, a_area + b_area - inter)
    if criterion == 0:
        return _div(inter, a_area)
    return _div(inter, b_area)


def to_polygon(g):
    """toPolygon :268-291: corners (l/2,w/2), (l/2,-w/2), (-l/2,-w/2), (-l/2,w/2) rotated by [[cos, sin], [-sin, cos]], + (t1, t3)."""
    c, s = math.cos(g.ry), math.sin(g.ry)
    pts = []
    for lx, wz in ((g.l / 2, g.w / 2), (g.l / 2, -g.w / 2), (-g.l / 2, -g.w / 2), (-g.l / 2, g.w / 2)):
        pts.append((c * lx + s * wz + g.t1, -s * lx + c * wz + g.t3))
    return pts


def _signed_area(p):
    return 0.5 * sum(p[i][0] * p[(i + 1) % len(p)][1] - p[(i + 1) % len(p)][0] * p[i][1] for i in range(len(p)))


def polygon_area(p):
    return abs(_signed_area(p)) if len(p) >= 3 else 0.0


def convex_intersection_area(a, b):
    """Area of the intersection of two convex polygons (Sutherland-Hodgman: clip ``a`` by every edge of ``b``)."""
    if _signed_area(a) < 0:
        a = a[::-1]
    if _signed_area(b) < 0:
        b = b[::-1]
    out = list(a)
    for i in range(len(b)):
        if not out:
            break
        (x0, y0), (x1, y1) = b[i], b[(i + 1) % len(b)]
        side = lambda p: (x1 - x0) * (p[1] - y0) - (y1 - y0) * (p[0] - x0)                  # noqa: E731  >= 0: inside (left of the edge)
        src, out = out, []
        for j in range(len(src)):
            p, q = src[j], src[(j + 1) % len(src)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp >= 0) != (sq >= 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return polygon_area(out)


def ground_box_overlap(d, g, criterion=-1):
    """groundBoxOverlap :294-314."""
    gp, dp = to_polygon(g), to_polygon(d)
    inter = convex_intersection_area(gp, dp)
    if criterion == -1:
        return _div(inter, polygon_area(gp) + polygon_area(dp) - inter)      # area(union_.front())
    if criterion == 0:
        return _div(inter, polygon_area(dp))
    return _div(inter, polygon_area(gp))


def box3d_overlap(d, g, criterion=-1):
    """box3DOverlap :317-344."""
    inter_area = convex_intersection_area(to_polygon(g), to_polygon(d))
    ymax = min(d.t2, g.t2)
    ymin = max(d.t2 - d.h, g.t2 - g.h)
    inter_vol = inter_area * max(0.0, ymax - ymin)
    det_vol = d.h * d.l * d.w
    gt_vol = g.h * g.l * g.w
    if criterion == -1:
        return _div(inter_vol, det_vol + gt_vol - inter_vol)
    if criterion == 0:
        return _div(inter_vol, det_vol)
    return _div(inter_vol, gt_vol)


OVERLAPS = (image_box_overlap, ground_box_overlap, box3d_overlap)


def get_thresholds(v, n_groundtruth):
    """getThresholds :346-379."""
    t = []
    v = sorted(v, reverse=True)
    current_recall = 0.0
    for i in range(len(v)):
        l_recall = _div(i + 1, n_groundtruth)
        r_recall = _div(i + 2, n_groundtruth) if i < len(v) - 1 else l_recall
        if (r_recall - current_recall) < (current_recall - l_recall) and i < len(v) - 1:
            continue
        t.append(v[i])
        current_recall += 1.0 / (N_SAMPLE_PTS - 1.0)
    return t


def clean_data(cls, gt, det, difficulty):
    """cleanData :381-454 -> (ignored_gt, dontcare, ignored_det, n_gt of this frame)."""
    ignored_gt, dc, ignored_det, n_gt = [], [], [], 0
    name = CLASS_NAMES[cls]
    for g in gt:
        height = g.y2 - g.y1
        ty = g.type.lower()
        if ty == name:
            valid = 1
        elif name == "pedestrian" and ty == "person_sitting":
            valid = 0
        elif name == "car" and ty == "van":
            valid = 0
        else:
            valid = -1
        ignore = g.occlusion > MAX_OCCLUSION[difficulty] or g.truncation > MAX_TRUNCATION[difficulty] or height < MIN_HEIGHT[difficulty]
        if valid == 1 and not ignore:
            ignored_gt.append(0)
            n_gt += 1
        elif valid == 0 or (ignore and valid == 1):
            ignored_gt.append(1)
        else:
            ignored_gt.append(-1)
    for g in gt:
        if g.type.lower() == "dontcare":
            dc.append(g)
    for d in det:
        valid = 1 if d.type.lower() == name else -1
        height = int(abs(d.y1 - d.y2))                   # int32_t height = fabs(...) (:444): truncated
        if height < MIN_HEIGHT[difficulty]:
            ignored_det.append(1)
        elif valid == 1:
            ignored_det.append(0)
        else:
            ignored_det.append(-1)
    return ignored_gt, dc, ignored_det, n_gt


def compute_statistics(cls, gt, det, dc, ignored_gt, ignored_det, compute_fp, metric, compute_aos=False, thresh=0.0):
    """computeStatistics :456-616 -> dict(tp, fp, fn, similarity, v)."""
    boxoverlap = OVERLAPS[metric]
    min_ov = MIN_OVERLAP[metric][cls]
    tp = fp = fn = 0
    similarity = 0.0
    v, delta = [], []
    assigned = [False] * len(det)
    ign_thr = [compute_fp and d.thresh < thresh for d in det]            # :471-474
    for i, g in enumerate(gt):
        if ignored_gt[i] == -1:
            continue
        det_idx, valid_detection, max_overlap, assigned_ignored_det = -1, NO_DETECTION, 0.0, False
        for j, d in enumerate(det):
            if ignored_det[j] == -1 or assigned[j] or ign_thr[j]:
                continue
            overlap = boxoverlap(d, g, -1)
            if not compute_fp and overlap > min_ov and d.thresh > valid_detection:                      # :506-509
                det_idx, valid_detection = j, d.thresh
            elif compute_fp and overlap > min_ov and (overlap > max_overlap or assigned_ignored_det) and ignored_det[j] == 0:   # :513-518
                max_overlap, det_idx, valid_detection, assigned_ignored_det = overlap, j, 1.0, False
            elif compute_fp and overlap > min_ov and valid_detection == NO_DETECTION and ignored_det[j] == 1:        # :519-523
                det_idx, valid_detection, assigned_ignored_det = j, 1.0, True
        if valid_detection == NO_DETECTION and ignored_gt[i] == 0:       # :531-533
            fn += 1
        elif valid_detection != NO_DETECTION and (ignored_gt[i] == 1 or ignored_det[det_idx] == 1):    # :536-537
            assigned[det_idx] = True
        elif valid_detection != NO_DETECTION:                            # :540-554
            tp += 1
            v.append(det[det_idx].thresh)
            if compute_aos:
                delta.append(g.alpha - det[det_idx].alpha)
            assigned[det_idx] = True
    if compute_fp:
        for i in range(len(det)):                                        # :561-566
            if not (assigned[i] or ignored_det[i] == -1 or ignored_det[i] == 1 or ign_thr[i]):
                fp += 1
        nstuff = 0
        for g in dc:                                                     # :570-588
            for j, d in enumerate(det):
                if assigned[j] or ignored_det[j] in (-1, 1) or ign_thr[j]:
                    continue
                if boxoverlap(d, g, 0) > min_ov:
                    assigned[j] = True
                    nstuff += 1
        fp -= nstuff
        if compute_aos:                                                  # :594-613
            tmp = [0.0] * fp + [(1.0 + math.cos(x)) / 2.0 for x in delta]
            assert len(tmp) == fp + tp and len(delta) == tp
            similarity = sum(tmp, 0.0) if (tp > 0 or fp > 0) else -1.0     # std::accumulate: a sequential double sum
    return {"tp": tp, "fp": fp, "fn": fn, "similarity": similarity, "v": v}


def _max_from(vals, i):
    """*max_element(begin + i, end) with operator< (NaN never replaces, a leading NaN stays)."""
    best = vals[i]
    for x in vals[i + 1:]:
        if best < x:
            best = x
    return best


def eval_class(cls, groundtruth, detections, compute_aos, difficulty, metric):
    """eval_class :622-706 -> (precision[41], aos[41] or None)."""
    n_gt = 0
    v, ign_gt, ign_det, dontcare = [], [], [], []
    for gt, det in zip(groundtruth, detections):
        i_gt, dc, i_det, n = clean_data(cls, gt, det, difficulty)
        n_gt += n
        ign_gt.append(i_gt)
        ign_det.append(i_det)
        dontcare.append(dc)
        v += compute_statistics(cls, gt, det, dc, i_gt, i_det, False, metric)["v"]
    thresholds = get_thresholds(v, n_gt)
    pr = [dict(tp=0, fp=0, fn=0, similarity=0.0) for _ in thresholds]
    for f, (gt, det) in enumerate(zip(groundtruth, detections)):
        for t, th in enumerate(thresholds):
            tmp = compute_statistics(cls, gt, det, dontcare[f], ign_gt[f], ign_det[f], True, metric, compute_aos, th)
            pr[t]["tp"] += tmp["tp"]
            pr[t]["fp"] += tmp["fp"]
            pr[t]["fn"] += tmp["fn"]
            if tmp["similarity"] != -1:
                pr[t]["similarity"] += tmp["similarity"]
    precision = [0.0] * N_SAMPLE_PTS
    aos = [0.0] * N_SAMPLE_PTS if compute_aos else None
    for i in range(len(thresholds)):
        precision[i] = _div(pr[i]["tp"], pr[i]["tp"] + pr[i]["fp"])
        if compute_aos:
            aos[i] = _div(pr[i]["similarity"], pr[i]["tp"] + pr[i]["fp"])
    for i in range(len(thresholds)):
        precision[i] = _max_from(precision, i)
        if compute_aos:
            aos[i] = _max_from(aos, i)
    return precision, aos


def average_precision_r40(vals):
    """saveAndPlotPlots :719-723: a FLOAT sum of points 1..40, / 40 * 100."""
    s = np.float32(0)
    for x in vals[1:]:
        s = np.float32(s + np.float32(x))
    return float(np.float32(np.float32(s / np.float32(40)) * np.float32(100)))


def get_eval_indices(result_data_dir):
    """getEvalIndices :778-793 (names shorter than 10 characters are skipped; the last 10 are parsed with atoi)."""
    out = []
    for name in os.listdir(result_data_dir):
        if len(name) < 10:
            continue
        digits = ""
        for ch in name[-10:].lstrip():
            if ch.isdigit() or (not digits and ch in "+-"):
                digits += ch
            else:
                break
        out.append(int(digits) if digits not in ("", "+", "-") else 0)
    return out


def evaluate(gt_dir, result_dir):
    """eval :795-915 without the plots / mail: {"car_detection": {"precision": [3][41], "ap": [3]}, "car_orientation": ...,
    "car_detection_ground": ..., "car_detection_3d": ..., ...} for every class / metric the result files enable."""
    state = {"compute_aos": True, "eval_image": [False] * 3, "eval_ground": [False] * 3, "eval_3d": [False] * 3}
    groundtruth, detections = [], []
    for idx in get_eval_indices(os.path.join(result_dir, "data")):
        name = "%06d.txt" % idx
        groundtruth.append(load_groundtruth(os.path.join(gt_dir, name)))
        detections.append(load_detections(os.path.join(result_dir, "data", name), state))
    out = {}
    for metric, flags, suffix in ((IMAGE, state["eval_image"], "_detection"), (GROUND, state["eval_ground"], "_detection_ground"),
                                  (BOX3D, state["eval_3d"], "_detection_3d")):
        aos_on = state["compute_aos"] and metric == IMAGE                # :876-877
        for c in range(3):
            if not flags[c]:
                continue
            rows = [eval_class(c, groundtruth, detections, aos_on, diff, metric) for diff in range(3)]
            out[CLASS_NAMES[c] + suffix] = {"precision": [r[0] for r in rows], "ap": [average_precision_r40(r[0]) for r in rows]}
            if aos_on:
                out[CLASS_NAMES[c] + "_orientation"] = {"precision": [r[1] for r in rows], "ap": [average_precision_r40(r[1]) for r in rows]}
    return out
