// KITTI object AP / AOS evaluator core -- host code of libsnvc_hip.so, no device work.
//
// What it replaces: the reference's standalone C++ tool tools/kitti-eval/evaluate_object_3d_offline(_r40).cpp
// (:227-344 overlaps on the image / on the ground plane / in 3D, :346-379 recall thresholds, :381-454 cleanData,
// :456-616 computeStatistics, :622-706 eval_class).  Same results (tests/test_kitti_eval.py: tables of the reference's
// own prebuilt binary), different construction:
//   * the tool recomputes every detection x ground-truth overlap inside computeStatistics, i.e. once per recall
//     threshold (41) x difficulty (3) x class (3) -- through Boost.Geometry's general polygon overlay for the rotated
//     boxes.  Here the three overlap matrices of a frame (and detection x DontCare) are computed ONCE, with a
//     closed-form clip of two convex quadrilaterals (no Boost), and every (metric, class, difficulty) sweep reads them;
//   * the 27 (metric, class, difficulty) sweeps are independent and run on a small thread pool;
//   * per sweep the per-frame ignore flags are computed once and the threshold loop touches only integers.
// The matching rules themselves (greedy per ground truth in file order, highest score for the recall pass, largest
// overlap for the precision pass, height-ignored detections, DontCare absorption) are the benchmark's definition and
// are kept decision for decision.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kPts = 41;
constexpr double kNoDetection = -10000000.0;
constexpr int kMinHeight[3] = {40, 25, 25};
constexpr int kMaxOcclusion[3] = {0, 1, 2};
constexpr double kMaxTruncation[3] = {0.15, 0.3, 0.5};

// gt row: truncation, occlusion, alpha, x1, y1, x2, y2, h, w, l, t1, t2, t3, ry
enum { G_TRUNC, G_OCC, G_ALPHA, G_X1, G_Y1, G_X2, G_Y2, G_H, G_W, G_L, G_T1, G_T2, G_T3, G_RY, G_COLS };
// det row: alpha, x1, y1, x2, y2, h, w, l, t1, t2, t3, ry, score
enum { D_ALPHA, D_X1, D_Y1, D_X2, D_Y2, D_H, D_W, D_L, D_T1, D_T2, D_T3, D_RY, D_SCORE, D_COLS };

struct Pt { double x, y; };

// corners of a ground-plane box, the tool's order and rotation (:268-291)
inline void bev_corners(double l, double w, double t1, double t3, double ry, Pt out[4]) {
    const double c = std::cos(ry), s = std::sin(ry);
    const double lx[4] = {l / 2, l / 2, -l / 2, -l / 2}, wz[4] = {w / 2, -w / 2, -w / 2, w / 2};
    for (int i = 0; i < 4; ++i) out[i] = {c * lx[i] + s * wz[i] + t1, -s * lx[i] + c * wz[i] + t3};
}

inline double signed_area(const Pt *p, int n) {
    double a = 0;
    for (int i = 0; i < n; ++i) {
        const Pt &u = p[i], &v = p[(i + 1) % n];
        a += u.x * v.y - v.x * u.y;
    }
    return 0.5 * a;
}

// area of the intersection of two convex quadrilaterals: a clipped by the four half-planes of b
double quad_intersection_area(const Pt *a_in, const Pt *b_in) {
    Pt a[4], b[4];
    const bool fa = signed_area(a_in, 4) < 0, fb = signed_area(b_in, 4) < 0;
    for (int i = 0; i < 4; ++i) {
        a[i] = a_in[fa ? 3 - i : i];
        b[i] = b_in[fb ? 3 - i : i];
    }
    Pt buf0[16], buf1[16];
    Pt *src = buf0, *dst = buf1;
    int n = 4;
    std::memcpy(src, a, sizeof(a));
    for (int e = 0; e < 4 && n > 0; ++e) {
        const Pt &p0 = b[e], &p1 = b[(e + 1) & 3];
        const double ex = p1.x - p0.x, ey = p1.y - p0.y;
        int m = 0;
        for (int j = 0; j < n; ++j) {
            const Pt &p = src[j], &q = src[j + 1 == n ? 0 : j + 1];
            const double sp = ex * (p.y - p0.y) - ey * (p.x - p0.x), sq = ex * (q.y - p0.y) - ey * (q.x - p0.x);
            if (sp >= 0) dst[m++] = p;
            if ((sp >= 0) != (sq >= 0)) {
                const double t = sp / (sp - sq);
                dst[m++] = {p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)};
            }
        }
        std::swap(src, dst);
        n = m;
    }
    return n >= 3 ? std::fabs(signed_area(src, n)) : 0.0;
}

struct Frame {
    int ng, nd, ndc;                 // ground truths, detections, DontCare regions (a subset of the ground truths, file order)
    const double *gt, *det;
    const int32_t *gt_type, *det_type;
    std::vector<int> dc;             // indices of the DontCare rows
    std::vector<double> ov[3];       // [metric][det * ng + gt], criterion -1 (both areas)
    std::vector<double> ovdc[3];     // [metric][dc * nd + det], criterion 0 (the detection's area)
    std::vector<int> det_height;     // int32_t height = fabs(y1 - y2): truncated (:444)
};

inline double image_overlap(const double *d, const double *g, int dx1, int gx1, bool det_area_only) {
    const double x1 = std::max(d[dx1], g[gx1]), y1 = std::max(d[dx1 + 1], g[gx1 + 1]);
    const double x2 = std::min(d[dx1 + 2], g[gx1 + 2]), y2 = std::min(d[dx1 + 3], g[gx1 + 3]);
    const double w = x2 - x1, h = y2 - y1;
    if (w <= 0 || h <= 0) return 0;
    const double inter = w * h;
    const double a_area = (d[dx1 + 2] - d[dx1]) * (d[dx1 + 3] - d[dx1 + 1]);
    const double b_area = (g[gx1 + 2] - g[gx1]) * (g[gx1 + 3] - g[gx1 + 1]);
    return det_area_only ? inter / a_area : inter / (a_area + b_area - inter);
}

void frame_overlaps(Frame &f) {
    std::vector<Pt> gc(4 * (size_t)f.ng), dc(4 * (size_t)f.nd);
    std::vector<double> garea(f.ng), darea(f.nd);
    for (int i = 0; i < f.ng; ++i) {
        const double *g = f.gt + (size_t)i * G_COLS;
        bev_corners(g[G_L], g[G_W], g[G_T1], g[G_T3], g[G_RY], &gc[4 * (size_t)i]);
        garea[i] = std::fabs(signed_area(&gc[4 * (size_t)i], 4));
    }
    for (int j = 0; j < f.nd; ++j) {
        const double *d = f.det + (size_t)j * D_COLS;
        bev_corners(d[D_L], d[D_W], d[D_T1], d[D_T3], d[D_RY], &dc[4 * (size_t)j]);
        darea[j] = std::fabs(signed_area(&dc[4 * (size_t)j], 4));
        f.det_height[j] = (int)std::fabs(d[D_Y1] - d[D_Y2]);
    }
    for (int m = 0; m < 3; ++m) {
        f.ov[m].assign((size_t)f.nd * f.ng, 0.0);
        f.ovdc[m].assign((size_t)f.ndc * f.nd, 0.0);
    }
    for (int j = 0; j < f.nd; ++j) {
        const double *d = f.det + (size_t)j * D_COLS;
        for (int i = 0; i < f.ng; ++i) {
            const double *g = f.gt + (size_t)i * G_COLS;
            const size_t k = (size_t)j * f.ng + i;
            f.ov[0][k] = image_overlap(d, g, D_X1, G_X1, false);
            const double inter = quad_intersection_area(&gc[4 * (size_t)i], &dc[4 * (size_t)j]);
            f.ov[1][k] = inter / (garea[i] + darea[j] - inter);        // the union of two overlapping convex polygons
            const double ymax = std::min(d[D_T2], g[G_T2]), ymin = std::max(d[D_T2] - d[D_H], g[G_T2] - g[G_H]);
            const double vol = inter * std::max(0.0, ymax - ymin);
            f.ov[2][k] = vol / (d[D_H] * d[D_L] * d[D_W] + g[G_H] * g[G_L] * g[G_W] - vol);
        }
    }
    for (int c = 0; c < f.ndc; ++c) {
        const int i = f.dc[c];
        const double *g = f.gt + (size_t)i * G_COLS;
        for (int j = 0; j < f.nd; ++j) {
            const double *d = f.det + (size_t)j * D_COLS;
            const size_t k = (size_t)c * f.nd + j;
            f.ovdc[0][k] = image_overlap(d, g, D_X1, G_X1, true);
            const double inter = quad_intersection_area(&gc[4 * (size_t)i], &dc[4 * (size_t)j]);
            f.ovdc[1][k] = inter / darea[j];
            const double ymax = std::min(d[D_T2], g[G_T2]), ymin = std::max(d[D_T2] - d[D_H], g[G_T2] - g[G_H]);
            f.ovdc[2][k] = inter * std::max(0.0, ymax - ymin) / (d[D_H] * d[D_L] * d[D_W]);
        }
    }
}

// ignore flags of one frame for (class, difficulty): 0 = counts, 1 = ignored (may absorb a match), -1 = other class
struct Flags {
    std::vector<int8_t> gt, det;
};

int frame_flags(const Frame &f, int cls, int diff, Flags &out) {
    out.gt.resize(f.ng);
    out.det.resize(f.nd);
    int n_gt = 0;
    for (int i = 0; i < f.ng; ++i) {
        const double *g = f.gt + (size_t)i * G_COLS;
        const int t = f.gt_type[i];
        // neighbouring classes: Van for Car, Person_sitting for Pedestrian (:398-401)
        const int valid = t == cls ? 1 : ((cls == 0 && t == SNVC_KITTI_VAN) || (cls == 1 && t == SNVC_KITTI_PERSON_SITTING)) ? 0 : -1;
        const bool ignore = (int)g[G_OCC] > kMaxOcclusion[diff] || g[G_TRUNC] > kMaxTruncation[diff] ||
                            (g[G_Y2] - g[G_Y1]) < kMinHeight[diff];
        if (valid == 1 && !ignore) {
            out.gt[i] = 0;
            ++n_gt;
        } else if (valid == 0 || (ignore && valid == 1)) {
            out.gt[i] = 1;
        } else {
            out.gt[i] = -1;
        }
    }
    for (int j = 0; j < f.nd; ++j)
        out.det[j] = f.det_height[j] < kMinHeight[diff] ? 1 : (f.det_type[j] == cls ? 0 : -1);
    return n_gt;
}

struct Counts {
    int64_t tp = 0, fp = 0, fn = 0;
    double similarity = 0;
    bool has_similarity = false;
};

// One frame at one score threshold (the precision pass), or the recall pass (collect = true: scores of the matches).
void match_frame(const Frame &f, const Flags &fl, int metric, double min_ov, bool recall_pass, double thresh, bool want_aos,
                 std::vector<uint8_t> &assigned, std::vector<double> *scores, Counts &out) {
    const std::vector<double> &ov = f.ov[metric];
    assigned.assign(f.nd, 0);
    int64_t tp = 0, fn = 0;
    double sim = 0;
    auto skip = [&](int j) {
        return fl.det[j] == -1 || assigned[j] || (!recall_pass && f.det[(size_t)j * D_COLS + D_SCORE] < thresh);
    };
    for (int i = 0; i < f.ng; ++i) {
        if (fl.gt[i] == -1) continue;
        int pick = -1;
        double best_score = kNoDetection, best_ov = 0;
        bool found = false, pick_is_ignored = false;
        for (int j = 0; j < f.nd; ++j) {
            if (skip(j)) continue;
            const double o = ov[(size_t)j * f.ng + i];
            if (!(o > min_ov)) continue;
            const double sc = f.det[(size_t)j * D_COLS + D_SCORE];
            if (recall_pass) {
                if (sc > best_score) {
                    pick = j;
                    best_score = sc;
                    found = true;
                }
            } else if (fl.det[j] == 0) {
                if (o > best_ov || pick_is_ignored) {
                    best_ov = o;
                    pick = j;
                    found = true;
                    pick_is_ignored = false;
                }
            } else if (!found) {       // a height-ignored detection is only taken while nothing else has been
                pick = j;
                found = true;
                pick_is_ignored = true;
            }
        }
        if (!found) {
            if (fl.gt[i] == 0) ++fn;
        } else if (fl.gt[i] == 1 || fl.det[pick] == 1) {
            assigned[pick] = 1;
        } else {
            ++tp;
            if (scores) scores->push_back(f.det[(size_t)pick * D_COLS + D_SCORE]);
            if (want_aos) sim += (1.0 + std::cos(f.gt[(size_t)i * G_COLS + G_ALPHA] - f.det[(size_t)pick * D_COLS + D_ALPHA])) / 2.0;
            assigned[pick] = 1;
        }
    }
    out.tp = tp;
    out.fn = fn;
    if (recall_pass) return;
    int64_t fp = 0;
    for (int j = 0; j < f.nd; ++j)
        if (!assigned[j] && fl.det[j] == 0 && !(f.det[(size_t)j * D_COLS + D_SCORE] < thresh)) ++fp;
    for (int c = 0; c < f.ndc; ++c)
        for (int j = 0; j < f.nd; ++j) {
            if (assigned[j] || fl.det[j] != 0 || f.det[(size_t)j * D_COLS + D_SCORE] < thresh) continue;
            if (f.ovdc[metric][(size_t)c * f.nd + j] > min_ov) {
                assigned[j] = 1;
                --fp;
            }
        }
    out.fp = fp;
    out.has_similarity = want_aos && (tp > 0 || fp > 0);
    out.similarity = sim;          // false positives add 0
}

void sweep(const std::vector<Frame> &frames, int metric, int cls, int diff, double min_ov, bool want_aos, double *precision,
           double *aos) {
    std::vector<Flags> flags(frames.size());
    std::vector<double> scores;
    std::vector<uint8_t> assigned;
    int64_t n_gt = 0;
    for (size_t k = 0; k < frames.size(); ++k) {
        n_gt += frame_flags(frames[k], cls, diff, flags[k]);
        Counts c;
        match_frame(frames[k], flags[k], metric, min_ov, true, 0, false, assigned, &scores, c);
    }
    // score thresholds at (approximately) equally spaced recall (:346-379)
    std::sort(scores.begin(), scores.end(), std::greater<double>());
    std::vector<double> th;
    double current = 0;
    const double ngt = (double)n_gt;
    for (size_t i = 0; i < scores.size(); ++i) {
        const bool last = i + 1 == scores.size();
        const double l_recall = (double)(i + 1) / ngt, r_recall = last ? l_recall : (double)(i + 2) / ngt;
        if ((r_recall - current) < (current - l_recall) && !last) continue;
        th.push_back(scores[i]);
        current += 1.0 / (kPts - 1.0);
    }
    std::vector<Counts> pr(th.size());
    for (size_t k = 0; k < frames.size(); ++k)
        for (size_t t = 0; t < th.size(); ++t) {
            Counts c;
            match_frame(frames[k], flags[k], metric, min_ov, false, th[t], want_aos, assigned, nullptr, c);
            pr[t].tp += c.tp;
            pr[t].fp += c.fp;
            pr[t].fn += c.fn;
            if (c.has_similarity) pr[t].similarity += c.similarity;
        }
    for (int i = 0; i < kPts; ++i) {
        precision[i] = 0;
        if (aos) aos[i] = 0;
    }
    for (size_t i = 0; i < th.size(); ++i) {
        precision[i] = (double)pr[i].tp / (double)(pr[i].tp + pr[i].fp);
        if (aos) aos[i] = pr[i].similarity / (double)(pr[i].tp + pr[i].fp);
    }
    // the curve is made monotone: p[i] = max over p[i..40], taken in ascending i (the entries right of i are still the
    // raw ones); std::max_element's `<` lets a NaN at position i stand and never picks one up from the right
    auto envelope = [&](double *p) {
        for (size_t i = 0; i < th.size(); ++i) {
            double best = p[i];
            for (int j = (int)i + 1; j < kPts; ++j)
                if (best < p[j]) best = p[j];
            p[i] = best;
        }
    };
    envelope(precision);
    if (aos) envelope(aos);
}

}  // namespace

int snvc_kitti_eval(const double *gt, const int32_t *gt_type, const int64_t *gt_offsets, const double *det,
                    const int32_t *det_type, const int64_t *det_offsets, int64_t frames, const double *min_overlap,
                    const int32_t *evaluate, int compute_aos, double *precision, double *aos, int threads) {
    using namespace snvc;
    if (frames < 0 || !gt_offsets || !det_offsets || !min_overlap || !evaluate || !precision)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_kitti_eval: null pointer or negative frame count");
    if (compute_aos && !aos) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_kitti_eval: compute_aos needs the aos table");
    for (int64_t k = 0; k < frames; ++k)
        if (gt_offsets[k + 1] < gt_offsets[k] || det_offsets[k + 1] < det_offsets[k] || gt_offsets[0] != 0 || det_offsets[0] != 0)
            return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_kitti_eval: offsets must start at 0 and not decrease");
    if (frames > 0 && ((gt_offsets[frames] > 0 && (!gt || !gt_type)) || (det_offsets[frames] > 0 && (!det || !det_type))))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_kitti_eval: null box table");
    std::vector<Frame> fr((size_t)frames);
    for (int64_t k = 0; k < frames; ++k) {
        Frame &f = fr[(size_t)k];
        f.ng = (int)(gt_offsets[k + 1] - gt_offsets[k]);
        f.nd = (int)(det_offsets[k + 1] - det_offsets[k]);
        f.gt = gt + gt_offsets[k] * G_COLS;
        f.det = det + det_offsets[k] * D_COLS;
        f.gt_type = gt_type + gt_offsets[k];
        f.det_type = det_type + det_offsets[k];
        for (int i = 0; i < f.ng; ++i)
            if (f.gt_type[i] == SNVC_KITTI_DONTCARE) f.dc.push_back(i);
        f.ndc = (int)f.dc.size();
        f.det_height.resize(f.nd);
    }
    int nthreads = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 32) nthreads = 32;
    {   // overlap matrices: frames are independent
        std::atomic<int64_t> next{0};
        auto work = [&] {
            for (int64_t k; (k = next.fetch_add(1)) < frames;) frame_overlaps(fr[(size_t)k]);
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads && t < frames; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    struct Job { int metric, cls, diff; };
    std::vector<Job> jobs;
    for (int m = 0; m < 3; ++m)
        for (int c = 0; c < 3; ++c)
            if (evaluate[m * 3 + c])
                for (int d = 0; d < 3; ++d) jobs.push_back({m, c, d});
    for (int i = 0; i < 3 * 3 * 3 * kPts; ++i) precision[i] = 0;
    if (aos)
        for (int i = 0; i < 3 * 3 * kPts; ++i) aos[i] = 0;
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t k; (k = next.fetch_add(1)) < jobs.size();) {
            const Job &j = jobs[k];
            const bool want_aos = compute_aos && j.metric == 0;          // orientation is scored with the image boxes only (:876-877)
            sweep(fr, j.metric, j.cls, j.diff, min_overlap[j.metric * 3 + j.cls], want_aos,
                  precision + ((j.metric * 3 + j.cls) * 3 + j.diff) * kPts, want_aos ? aos + (j.cls * 3 + j.diff) * kPts : nullptr);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads && t < (int)jobs.size(); ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return SNVC_OK;
}
