/*
 * oracle/roiaware_pool3d_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's RoI-aware point pooling:
 *   membership + voxel index : snvc/extension/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-75
 *   per-voxel point lists    : roiaware_pool3d_kernel.cu:78-108
 *   max / avg pooling        : roiaware_pool3d_kernel.cu:111-190
 *   backward                 : roiaware_pool3d_kernel.cu:236-286
 *   points_in_boxes (gpu)    : roiaware_pool3d_kernel.cu:313-336
 *   points_in_boxes (cpu)    : snvc/extension/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168
 *
 * PARITY UNPINNED BY THE REFERENCE: no tests / golden vectors exist upstream
 * and the CUDA sources cannot be built here.  Pinned by hand-derived
 * known-answer tests (tests/test_oracle_roiaware.py).
 *
 * One deliberate, documented choice: the box rotation uses
 * (float)cos((double)angle) / (float)sin((double)angle), i.e. the correctly
 * rounded float value, on both this oracle and the HIP kernel.  The reference
 * calls device cosf/sinf (2-ulp error bound), which neither a CPU libm nor
 * another GPU libm reproduces bit-for-bit; the correctly rounded value is the
 * one result every implementation can agree on, and it is what makes the
 * voxel indices bit-comparable across CPU and GPU.
 *
 * Compiled with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

static inline void to_box_frame(float sx, float sy, float angle, float *lx, float *ly)
{
    /* roiaware_pool3d_kernel.cu:16-20 (rotate by -heading) */
    const float cosa = (float)cos((double)(-angle)), sina = (float)sin((double)(-angle));
    *lx = sx * cosa + sy * (-sina);
    *ly = sx * sina + sy * cosa;
}

static inline int point_in_box(const float *pt, const float *box, float margin,
                               float *lx, float *ly)
{
    /* roiaware_pool3d_kernel.cu:23-36; margin 1e-5 there, 1e-2 in roiaware_pool3d.cpp:131 */
    const float x = pt[0], y = pt[1], z = pt[2];
    const float cx = box[0], cy = box[1], cz = box[2];
    const float dx = box[3], dy = box[4], dz = box[5], rz = box[6];
    if ((double)fabsf(z - cz) > (double)dz / 2.0) return 0;
    to_box_frame(x - cx, y - cy, rz, lx, ly);
    return ((double)fabsf(*lx) < (double)dx / 2.0 + (double)margin) &
           ((double)fabsf(*ly) < (double)dy / 2.0 + (double)margin);
}

static inline unsigned clamp_index(float v, int out)
{
    /* roiaware_pool3d_kernel.cu:64-70: int() truncation, stored unsigned, then
     * min(max(idx, 0), out - 1) evaluated on the unsigned value. */
    unsigned idx = (unsigned)(int)v;
    const unsigned hi = (unsigned)(out - 1);
    return idx < hi ? idx : hi;
}

/* generate_pts_mask_for_box3d, roiaware_pool3d_kernel.cu:39-75.
 * mask [B,P] int32: -1 outside, else (x<<16)+(y<<8)+z. */
ORACLE_API void oracle_roiaware_mask(const float *rois, const float *pts, int32_t *mask,
                                     int B, int P, int ox, int oy, int oz)
{
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < P; ++p) {
            const float *box = rois + 7 * b, *pt = pts + 3 * p;
            float lx = 0, ly = 0;
            int32_t code = -1;
            if (point_in_box(pt, box, 1e-5f, &lx, &ly)) {
                const float lz = pt[2] - box[2];
                const float dx = box[3], dy = box[4], dz = box[5];
                const float xr = dx / ox, yr = dy / oy, zr = dz / oz;
                const unsigned xi = clamp_index((lx + dx / 2) / xr, ox);
                const unsigned yi = clamp_index((ly + dy / 2) / yr, oy);
                const unsigned zi = clamp_index((lz + dz / 2) / zr, oz);
                code = (int32_t)((xi << 16) + (yi << 8) + zi);
            }
            mask[(int64_t)b * P + p] = code;
        }
}

/* roiaware_pool3d_launcher, roiaware_pool3d_kernel.cu:193-233.
 * Caller zero-fills argmax / pts_idx_of_voxels / pooled like
 * roiaware_pool3d_utils.py:124-126.  pool_method 0 = max, 1 = avg. */
ORACLE_API void oracle_roiaware_pool3d_forward(
    const float *rois, const float *pts, const float *feat,
    int32_t *argmax, int32_t *pts_idx_of_voxels, float *pooled,
    int B, int P, int C, int max_pts, int ox, int oy, int oz, int pool_method,
    int32_t *mask_scratch /* [B,P] */)
{
    oracle_roiaware_mask(rois, pts, mask_scratch, B, P, ox, oy, oz);
    const int cap = max_pts - 1; /* slot 0 holds the count, :86 */
    for (int b = 0; b < B; ++b) {
        int32_t *lists = pts_idx_of_voxels + (int64_t)b * ox * oy * oz * max_pts;
        for (int k = 0; k < P; ++k) { /* ascending point order, :89-107 */
            const int32_t code = mask_scratch[(int64_t)b * P + k];
            if (code == -1) continue;
            const unsigned xi = ((unsigned)code >> 16) & 0xFF, yi = ((unsigned)code >> 8) & 0xFF,
                           zi = (unsigned)code & 0xFF;
            int32_t *cell = lists + ((int64_t)(xi * oy + yi) * oz + zi) * max_pts;
            if (cell[0] < cap) { cell[cell[0] + 1] = k; cell[0]++; }
        }
    }
    const int64_t nvox = (int64_t)ox * oy * oz;
    for (int b = 0; b < B; ++b)
        for (int64_t v = 0; v < nvox; ++v) {
            const int32_t *cell = pts_idx_of_voxels + ((int64_t)b * nvox + v) * max_pts;
            const int total = cell[0];
            for (int c = 0; c < C; ++c) {
                const int64_t o = ((int64_t)b * nvox + v) * C + c;
                if (pool_method == 0) { /* :111-157 */
                    int32_t best = -1;
                    float best_val = -INFINITY; /* float(-1e50) overflows to -inf, :137 */
                    for (int k = 1; k <= total; ++k) {
                        const float f = feat[(int64_t)cell[k] * C + c];
                        if (f > best_val) { best_val = f; best = cell[k]; }
                    }
                    if (best != -1) pooled[o] = best_val;
                    argmax[o] = best;
                } else { /* :160-190 */
                    float sum = 0;
                    for (int k = 1; k <= total; ++k) sum += feat[(int64_t)cell[k] * C + c];
                    if (total > 0) pooled[o] = sum / total;
                }
            }
        }
}

/* roiaware_pool3d_backward_launcher, roiaware_pool3d_kernel.cu:236-310.
 * grad_in [P,C] is accumulated into (caller zero-fills, roiaware_pool3d_utils.py:142).
 * Sequential (box, voxel, channel) order; the reference's atomics leave it undefined. */
ORACLE_API void oracle_roiaware_pool3d_backward(
    const int32_t *pts_idx_of_voxels, const int32_t *argmax, const float *grad_out,
    float *grad_in, int B, int C, int max_pts, int ox, int oy, int oz, int pool_method)
{
    const int64_t nvox = (int64_t)ox * oy * oz;
    for (int b = 0; b < B; ++b)
        for (int64_t v = 0; v < nvox; ++v)
            for (int c = 0; c < C; ++c) {
                const int64_t o = ((int64_t)b * nvox + v) * C + c;
                if (pool_method == 0) {
                    if (argmax[o] == -1) continue;
                    grad_in[(int64_t)argmax[o] * C + c] += grad_out[o] * 1;
                } else {
                    const int32_t *cell = pts_idx_of_voxels + ((int64_t)b * nvox + v) * max_pts;
                    const int total = cell[0];
                    const float share = 1 / fmaxf((float)total, 1.0f);
                    for (int k = 1; k <= total; ++k)
                        grad_in[(int64_t)cell[k] * C + c] += grad_out[o] * share;
                }
            }
}

/* points_in_boxes_kernel, roiaware_pool3d_kernel.cu:313-336: first containing box, else
 * the caller's prefill (-1, roiaware_pool3d_utils.py:81).  boxes [Bs,T,7], pts [Bs,M,3]. */
ORACLE_API void oracle_points_in_boxes_gpu(const float *boxes, const float *pts, int32_t *out,
                                           int Bs, int T, int M)
{
    for (int s = 0; s < Bs; ++s)
        for (int m = 0; m < M; ++m) {
            float lx = 0, ly = 0;
            for (int k = 0; k < T; ++k)
                if (point_in_box(pts + ((int64_t)s * M + m) * 3, boxes + ((int64_t)s * T + k) * 7,
                                 1e-5f, &lx, &ly)) {
                    out[(int64_t)s * M + m] = k;
                    break;
                }
        }
}

/* points_in_boxes_cpu, roiaware_pool3d.cpp:137-168: 0/1 flag per (box, point), margin 1e-2. */
ORACLE_API void oracle_points_in_boxes_cpu(const float *boxes, const float *pts, int32_t *out,
                                           int T, int M)
{
    float lx = 0, ly = 0;
    for (int i = 0; i < T; ++i)
        for (int j = 0; j < M; ++j)
            out[(int64_t)i * M + j] = point_in_box(pts + 3 * j, boxes + 7 * i, 1e-2f, &lx, &ly);
}
