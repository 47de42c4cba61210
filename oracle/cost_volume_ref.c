/*
 * oracle/cost_volume_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's plane-sweep concat cost volume
 * (the reference has no CPU path for this op: BuildCostVolume.cpp:26,41 raise
 * "Not implemented on the CPU"; the CUDA source is the specification).
 *
 *   forward   : snvc/extension/build_cost_volume/src/BuildCostVolume_cuda.cu:15-98,208-256
 *   backward  : snvc/extension/build_cost_volume/src/BuildCostVolume_cuda.cu:101-205,259-303
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests or golden
 * vectors for this op and the CUDA-only sources cannot be built here (nvcc and
 * the removed THC headers are absent).  The restatement is pinned instead by
 * hand-derived known-answer tests (tests/test_oracle_cost_volume.py, SURVEY.md
 * section 8c items i-vii) and by an independent torch grid_sample expression.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  The product path never does.
 *
 * Compiled with -ffp-contract=off so that every product and sum below is
 * rounded individually, in the left-to-right order the reference writes them.
 * Backward sums are accumulated in a fixed (d, w, tap) order -- the reference
 * uses float atomics whose order is undefined, so any order is admissible; the
 * HIP kernel uses this same order, which makes the two bit-identical.
 *
 * Indexing is 64-bit throughout (the reference's 32-bit index math overflows
 * once N*2C*D*H*W >= 2^31, BuildCostVolume_cuda.cu:70-82).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* One x-sample of the right row.  Mirrors the clamp rules of
 * bilinear_interpolate (BuildCostVolume_cuda.cu:15-61) for the only case the
 * forward kernel can reach: y integral (ly = 0) and x already gated to
 * [0, img_w - 1] (BuildCostVolume_cuda.cu:88).  The row below (y_high) is
 * still loaded and multiplied by a zero weight exactly as the reference does
 * (BuildCostVolume_cuda.cu:54-58), so a non-finite value there leaks through
 * in the same way. */
#define DEFINE_FOR_TYPE(T, SUFFIX)                                                          \
static T sample_row_##SUFFIX(const T *plane, int64_t img_h, int64_t img_w,                  \
                             int64_t iy, T x)                                               \
{                                                                                           \
    T y = (T)iy;                                                                            \
    if (y < (T)-1.0 || y > (T)img_h || x < (T)-1.0 || x > (T)img_w) return (T)0;           \
    if (y <= 0) y = 0;                                                                      \
    if (x <= 0) x = 0;                                                                      \
    int64_t y_lo = (int64_t)y, x_lo = (int64_t)x, y_hi, x_hi;                               \
    if (y_lo >= img_h - 1) { y_hi = y_lo = img_h - 1; y = (T)y_lo; } else y_hi = y_lo + 1;  \
    if (x_lo >= img_w - 1) { x_hi = x_lo = img_w - 1; x = (T)x_lo; } else x_hi = x_lo + 1;  \
    T ly = y - (T)y_lo, lx = x - (T)x_lo;                                                   \
    T hy = (T)1.0 - ly, hx = (T)1.0 - lx;                                                   \
    T v1 = plane[y_lo * img_w + x_lo], v2 = plane[y_lo * img_w + x_hi];                     \
    T v3 = plane[y_hi * img_w + x_lo], v4 = plane[y_hi * img_w + x_hi];                     \
    T w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;                               \
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;                                           \
}                                                                                           \
                                                                                            \
/* weights + tap positions, BuildCostVolume_cuda.cu:101-150 */                              \
static void sample_weights_##SUFFIX(int64_t img_h, int64_t img_w, int64_t iy, T x,          \
                                    T w[4], int64_t *x_lo_, int64_t *x_hi_,                 \
                                    int64_t *y_lo_, int64_t *y_hi_)                         \
{                                                                                           \
    T y = (T)iy;                                                                            \
    if (y < (T)-1.0 || y > (T)img_h || x < (T)-1.0 || x > (T)img_w) {                       \
        w[0] = w[1] = w[2] = w[3] = 0; *x_lo_ = *x_hi_ = *y_lo_ = *y_hi_ = -1; return;      \
    }                                                                                       \
    if (y <= 0) y = 0;                                                                      \
    if (x <= 0) x = 0;                                                                      \
    int64_t y_lo = (int64_t)y, x_lo = (int64_t)x, y_hi, x_hi;                               \
    if (y_lo >= img_h - 1) { y_hi = y_lo = img_h - 1; y = (T)y_lo; } else y_hi = y_lo + 1;  \
    if (x_lo >= img_w - 1) { x_hi = x_lo = img_w - 1; x = (T)x_lo; } else x_hi = x_lo + 1;  \
    T ly = y - (T)y_lo, lx = x - (T)x_lo;                                                   \
    T hy = (T)1.0 - ly, hx = (T)1.0 - lx;                                                   \
    w[0] = hy * hx; w[1] = hy * lx; w[2] = ly * hx; w[3] = ly * lx;                         \
    *x_lo_ = x_lo; *x_hi_ = x_hi; *y_lo_ = y_lo; *y_hi_ = y_hi;                             \
}                                                                                           \
                                                                                            \
/* BuildCostVolumeForward, BuildCostVolume_cuda.cu:63-98.                                   \
 * left,right [N,C,Hi,Wi]; shift [N,D]; out [N,2C,D,Hi/ds,Wi/ds].  Returns 0, or -1 if     \
 * the shapes are unusable (ds < 1, or Hi/Wi not multiples of ds: the reference silently    \
 * uses H*ds / W*ds as strides, BuildCostVolume_cuda.cu:78-79, which is only meaningful     \
 * then). */                                                                                \
ORACLE_API int oracle_cost_volume_forward_##SUFFIX(                                         \
    const T *left, const T *right, const T *shift, T *out,                                  \
    int64_t N, int64_t C, int64_t Hi, int64_t Wi, int64_t D, int64_t ds)                    \
{                                                                                           \
    if (ds < 1 || Hi % ds || Wi % ds) return -1;                                            \
    const int64_t H = Hi / ds, W = Wi / ds;                                                 \
    const int64_t img_h = H * ds, img_w = W * ds;                                           \
    /* (n, c) planes are independent: OpenMP over them changes no bit of the result */     \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                \
    for (int64_t n = 0; n < N; ++n)                                                         \
    for (int64_t c = 0; c < C; ++c) {                                                       \
        const T *lplane = left + (n * C + c) * img_h * img_w;                               \
        const T *rplane = right + (n * C + c) * img_h * img_w;                              \
        for (int64_t d = 0; d < D; ++d) {                                                   \
            const T neg_shift = -shift[n * D + d];                                          \
            T *oL = out + ((n * 2 * C + c) * D + d) * H * W;                                \
            T *oR = oL + C * D * H * W;                                                     \
            for (int64_t h = 0; h < H; ++h)                                                 \
            for (int64_t w = 0; w < W; ++w) {                                               \
                const int64_t iw = w * ds, ih = h * ds;                                     \
                oL[h * W + w] = lplane[ih * img_w + iw];                                    \
                const T x = (T)iw + neg_shift;                                              \
                if (x >= (T)0.0 && x <= (T)(img_w - 1))                                     \
                    oR[h * W + w] = sample_row_##SUFFIX(rplane, img_h, img_w, ih, x);       \
                else                                                                        \
                    oR[h * W + w] = (T)0.0;                                                 \
            }                                                                               \
        }                                                                                   \
    }                                                                                       \
    return 0;                                                                               \
}                                                                                           \
                                                                                            \
/* BuildCostVolumeBackwardFeature, BuildCostVolume_cuda.cu:152-205.                         \
 * grad [N,2C,D,H,W] -> gL,gR [N,C,H*ds,W*ds] (zero-filled here, like at::zeros :270-271). \
 * Accumulation order: d ascending, then h, w ascending, then tap 1..4. */                  \
ORACLE_API int oracle_cost_volume_backward_##SUFFIX(                                        \
    const T *grad, const T *shift, T *gL, T *gR,                                            \
    int64_t N, int64_t C, int64_t H, int64_t W, int64_t D, int64_t ds)                      \
{                                                                                           \
    if (ds < 1) return -1;                                                                  \
    const int64_t img_h = H * ds, img_w = W * ds;                                           \
    memset(gL, 0, sizeof(T) * (size_t)(N * C * img_h * img_w));                             \
    memset(gR, 0, sizeof(T) * (size_t)(N * C * img_h * img_w));                             \
    /* each (n, c) pair owns its two gradient planes; the order inside a plane is fixed */  \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                \
    for (int64_t n = 0; n < N; ++n)                                                         \
    for (int64_t c = 0; c < C; ++c) {                                                       \
        T *glp = gL + (n * C + c) * img_h * img_w;                                          \
        T *grp = gR + (n * C + c) * img_h * img_w;                                          \
        for (int64_t d = 0; d < D; ++d) {                                                   \
            const T neg_shift = -shift[n * D + d];                                          \
            const T *gl = grad + ((n * 2 * C + c) * D + d) * H * W;                         \
            const T *gr = gl + C * D * H * W;                                               \
            for (int64_t h = 0; h < H; ++h)                                                 \
            for (int64_t w = 0; w < W; ++w) {                                               \
                const int64_t iw = w * ds, ih = h * ds;                                     \
                glp[ih * img_w + iw] += gl[h * W + w];                                      \
                const T x = (T)iw + neg_shift;                                              \
                if (x >= (T)0.0 && x <= (T)(img_w - 1)) {                                   \
                    T wt[4]; int64_t x_lo, x_hi, y_lo, y_hi;                                \
                    sample_weights_##SUFFIX(img_h, img_w, ih, x, wt, &x_lo, &x_hi,          \
                                            &y_lo, &y_hi);                                  \
                    const T g = gr[h * W + w];                                              \
                    const T g1 = g * wt[0], g2 = g * wt[1], g3 = g * wt[2], g4 = g * wt[3]; \
                    if ((double)wt[0] >= 1e-10) grp[y_lo * img_w + x_lo] += g1;                  \
                    if ((double)wt[1] >= 1e-10) grp[y_lo * img_w + x_hi] += g2;                  \
                    if ((double)wt[2] >= 1e-10) grp[y_hi * img_w + x_lo] += g3;                  \
                    if ((double)wt[3] >= 1e-10) grp[y_hi * img_w + x_hi] += g4;                  \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
    }                                                                                       \
    return 0;                                                                               \
}

DEFINE_FOR_TYPE(float, f32)
DEFINE_FOR_TYPE(double, f64)
