// Element-wise companions of the fp16-storage conv family (conv3d_f16.hip): layout conversion between the
// reference's contiguous NCDHW fp32 tensors and the C8 half layout [N][C/8][S][8] (S = D*H*W), and the two
// element-wise steps of predict_3d_heatmaps in that layout (snvc/models/vernier.py:433 and :289,436-438).
// All of them are single-pass HBM streams: one 16-byte piece per thread on the C8 side, rows of one channel
// coalesced across the wave on the NCDHW side.
#include "common.hpp"

namespace snvc {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256)
ncdhw_f32_to_c8_kernel(const float *__restrict__ x, _Float16 *__restrict__ y, int C, int64_t S, int64_t x_bs, int64_t y_bs) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int g = blockIdx.y;
    const int64_t n = blockIdx.z;
    const float *xp = x + n * x_bs + (int64_t)g * 8 * S + s;
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (g * 8 + e < C) ? (_Float16)xp[(int64_t)e * S] : (_Float16)0.0f;
    *reinterpret_cast<h8 *>(y + n * y_bs + ((int64_t)g * S + s) * 8) = o;
}

__global__ void __launch_bounds__(256)
c8_to_ncdhw_f32_kernel(const _Float16 *__restrict__ x, float *__restrict__ y, int C, int64_t S, int64_t x_bs, int64_t y_bs) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int g = blockIdx.y;
    const int64_t n = blockIdx.z;
    const h8 v = *reinterpret_cast<const h8 *>(x + n * x_bs + ((int64_t)g * S + s) * 8);
    float *yp = y + n * y_bs + (int64_t)g * 8 * S + s;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (g * 8 + e < C) yp[(int64_t)e * S] = (float)v[e];
}

// split mode (conv3d_f16.hip, F16Cfg::PL == 2): value * mul (a power of two) = hi + lo, two C8 half planes
__global__ void __launch_bounds__(256)
ncdhw_f32_to_c8_split_kernel(const float *__restrict__ x, _Float16 *__restrict__ yh, _Float16 *__restrict__ yl, int C, int64_t S,
                             int64_t x_bs, int64_t y_bs, float mul, const float *__restrict__ mul_dev) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    if (mul_dev) mul = *mul_dev;             // the scale chosen on the device (no host round trip): a power of two
    const int g = blockIdx.y;
    const int64_t n = blockIdx.z;
    const float *xp = x + n * x_bs + (int64_t)g * 8 * S + s;
    h8 o, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = (g * 8 + e < C) ? xp[(int64_t)e * S] * mul : 0.0f;
        o[e] = (_Float16)v;
        ol[e] = (_Float16)(v - (float)o[e]);
    }
    *reinterpret_cast<h8 *>(yh + n * y_bs + ((int64_t)g * S + s) * 8) = o;
    *reinterpret_cast<h8 *>(yl + n * y_bs + ((int64_t)g * S + s) * 8) = ol;
}

// GroupNorm layers in split mode (r5): the layer's raw fp32 NCDHW result -> the split C8 pair of  act(scale * raw + shift [+ res]) [+ res],
// scale / shift per (sample, channel) from the statistics of `raw` (snvc_norm_stats), the residual a split pair in its own units
// (res_mul = 2^-e_res).  Result * out_mul (2^e) is clamped to half's range and flagged like every split-mode epilogue.
__global__ void __launch_bounds__(256)
ncdhw_affine_to_c8_split_kernel(const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                                const _Float16 *__restrict__ rh, const _Float16 *__restrict__ rl, _Float16 *__restrict__ yh,
                                _Float16 *__restrict__ yl, int *__restrict__ overflow, int C, int64_t S, int64_t x_bs, int64_t y_bs,
                                int64_t r_bs, int per_sample, int flags, float out_mul, float res_mul) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int g = blockIdx.y;
    const int64_t n = blockIdx.z;
    const float *xp = x + n * x_bs + (int64_t)g * 8 * S + s;
    const float *sc = scale + (per_sample ? n * C : 0) + g * 8, *sh = shift + (per_sample ? n * C : 0) + g * 8;
    const bool relu = (flags & SNVC_EPI_RELU) != 0, pre = (flags & SNVC_EPI_ADD_PRE) != 0, post = (flags & SNVC_EPI_ADD_POST) != 0;
    h8 rhi = {}, rlo = {};
    if (rh) {
        rhi = *reinterpret_cast<const h8 *>(rh + n * r_bs + ((int64_t)g * S + s) * 8);
        rlo = *reinterpret_cast<const h8 *>(rl + n * r_bs + ((int64_t)g * S + s) * 8);
    }
    h8 o, ol;
    float vmax = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = 0.0f;
        if (g * 8 + e < C) {
            v = __builtin_fmaf(xp[(int64_t)e * S], sc[e], sh[e]);
            const float r = rh ? ((float)rhi[e] + (float)rlo[e]) * res_mul : 0.0f;
            if (pre) v += r;
            if (relu) v = v > 0.0f ? v : 0.0f;
            if (post) v += r;
            v *= out_mul;
        }
        const float c = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
        vmax = __builtin_fmaxf(vmax, __builtin_fabsf(c));
        o[e] = (_Float16)c;
        ol[e] = (_Float16)(c - (float)o[e]);
    }
    *reinterpret_cast<h8 *>(yh + n * y_bs + ((int64_t)g * S + s) * 8) = o;
    *reinterpret_cast<h8 *>(yl + n * y_bs + ((int64_t)g * S + s) * 8) = ol;
    if (vmax >= 65504.0f && overflow) atomicOr(overflow, 1);
}

__global__ void __launch_bounds__(256)
c8_split_to_ncdhw_f32_kernel(const _Float16 *__restrict__ xh, const _Float16 *__restrict__ xl, float *__restrict__ y, int C, int64_t S,
                             int64_t x_bs, int64_t y_bs, float mul) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int g = blockIdx.y;
    const int64_t n = blockIdx.z;
    const h8 v = *reinterpret_cast<const h8 *>(xh + n * x_bs + ((int64_t)g * S + s) * 8);
    const h8 l = *reinterpret_cast<const h8 *>(xl + n * x_bs + ((int64_t)g * S + s) * 8);
    float *yp = y + n * y_bs + (int64_t)g * 8 * S + s;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (g * 8 + e < C) yp[(int64_t)e * S] = ((float)v[e] + (float)l[e]) * mul;
}

// out[n][g][s][:] = feat[n][g][s][:] * occ[n][s]   (occ: fp32 plane; product in fp32, rounded once)
__global__ void __launch_bounds__(256)
mul_broadcast_c8_kernel(const _Float16 *__restrict__ feat, const float *__restrict__ occ, _Float16 *__restrict__ out,
                        int64_t S, int64_t f_bs, int64_t o_bs) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int g = blockIdx.y;
    const int64_t n = blockIdx.z;
    const h8 v = *reinterpret_cast<const h8 *>(feat + n * f_bs + ((int64_t)g * S + s) * 8);
    const float w = occ[n * S + s];
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)v[e] * w);
    *reinterpret_cast<h8 *>(out + n * o_bs + ((int64_t)g * S + s) * 8) = o;
}

// split mode: out = (hi + lo) * occ, re-split (the product of the pair's VALUE with the occupancy, not of its halves)
__global__ void __launch_bounds__(256)
mul_broadcast_c8_split_kernel(const _Float16 *__restrict__ fh, const _Float16 *__restrict__ fl, const float *__restrict__ occ,
                              _Float16 *__restrict__ oh, _Float16 *__restrict__ ol, int64_t S, int64_t f_bs, int64_t o_bs) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int g = blockIdx.y;
    const int64_t n = blockIdx.z;
    const h8 a = *reinterpret_cast<const h8 *>(fh + n * f_bs + ((int64_t)g * S + s) * 8);
    const h8 b = *reinterpret_cast<const h8 *>(fl + n * f_bs + ((int64_t)g * S + s) * 8);
    const float w = occ[n * S + s];
    h8 o, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = ((float)a[e] + (float)b[e]) * w;
        o[e] = (_Float16)v;
        l[e] = (_Float16)(v - (float)o[e]);
    }
    *reinterpret_cast<h8 *>(oh + n * o_bs + ((int64_t)g * S + s) * 8) = o;
    *reinterpret_cast<h8 *>(ol + n * o_bs + ((int64_t)g * S + s) * 8) = l;
}

// AvgPool3d((4,1,1)) of a C8 tensor [N][G][D][HW][8] into the fp32 NCDHW tensor [N][C][D/4][HW] the 2D neck reads
__global__ void __launch_bounds__(256)
avgpool_depth4_c8_kernel(const _Float16 *__restrict__ x, float *__restrict__ y, int C, int D, int64_t HW, int64_t x_bs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int D4 = D / 4;
    if (i >= (int64_t)D4 * HW) return;
    const int64_t hw = i % HW;
    const int d4 = (int)(i / HW);
    const int g = blockIdx.y;
    const int64_t n = blockIdx.z;
    const _Float16 *xp = x + n * x_bs + (((int64_t)g * D + 4 * d4) * HW + hw) * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const h8 v = *reinterpret_cast<const h8 *>(xp + (int64_t)k * HW * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
    }
    float *yp = y + ((n * C + (int64_t)g * 8) * D4 + d4) * HW + hw;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (g * 8 + e < C) yp[(int64_t)e * D4 * HW] = acc[e] * 0.25f;
}

inline bool grid_ok(int64_t S, int64_t G, int64_t N) { return ceil_div<int64_t>(S, 256) < ((int64_t)1 << 31) && G <= 65535 && N <= 65535; }

}  // namespace
}  // namespace snvc

extern "C" {

int snvc_f16_from_ncdhw(const float *x, void *y, int64_t N, int64_t C, int64_t S, int64_t x_batch_stride,
                        int64_t y_batch_stride, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || S < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_from_ncdhw: bad sizes");
    if (N == 0 || S == 0) return SNVC_OK;
    const int64_t G = ceil_div<int64_t>(C, 8);
    if (!x || !y || (reinterpret_cast<uintptr_t>(y) & 15)) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_from_ncdhw: null or unaligned pointer");
    if (!grid_ok(S, G, N)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_from_ncdhw: tensor too large");
    dim3 grid((unsigned)ceil_div<int64_t>(S, 256), (unsigned)G, (unsigned)N);
    ncdhw_f32_to_c8_kernel<<<grid, 256, 0, as_stream(stream)>>>(x, reinterpret_cast<_Float16 *>(y), (int)C, S,
                                                               x_batch_stride ? x_batch_stride : C * S,
                                                               y_batch_stride ? y_batch_stride : G * 8 * S);
    return check_launch("snvc_f16_from_ncdhw");
}

int snvc_f16_to_ncdhw(const void *x, float *y, int64_t N, int64_t C, int64_t S, int64_t x_batch_stride,
                      int64_t y_batch_stride, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || S < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_to_ncdhw: bad sizes");
    if (N == 0 || S == 0) return SNVC_OK;
    const int64_t G = ceil_div<int64_t>(C, 8);
    if (!x || !y || (reinterpret_cast<uintptr_t>(x) & 15)) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_to_ncdhw: null or unaligned pointer");
    if (!grid_ok(S, G, N)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_to_ncdhw: tensor too large");
    dim3 grid((unsigned)ceil_div<int64_t>(S, 256), (unsigned)G, (unsigned)N);
    c8_to_ncdhw_f32_kernel<<<grid, 256, 0, as_stream(stream)>>>(reinterpret_cast<const _Float16 *>(x), y, (int)C, S,
                                                               x_batch_stride ? x_batch_stride : G * 8 * S,
                                                               y_batch_stride ? y_batch_stride : C * S);
    return check_launch("snvc_f16_to_ncdhw");
}

int snvc_f16x3_from_ncdhw(const float *x, void *y_hi, void *y_lo, int64_t N, int64_t C, int64_t S, int64_t x_batch_stride,
                          int64_t y_batch_stride, float mul, const float *mul_dev, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || S < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_from_ncdhw: bad sizes");
    if (N == 0 || S == 0) return SNVC_OK;
    const int64_t G = ceil_div<int64_t>(C, 8);
    if (!x || !y_hi || !y_lo || ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 15))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_from_ncdhw: null or unaligned pointer");
    if (!grid_ok(S, G, N)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_from_ncdhw: tensor too large");
    dim3 grid((unsigned)ceil_div<int64_t>(S, 256), (unsigned)G, (unsigned)N);
    ncdhw_f32_to_c8_split_kernel<<<grid, 256, 0, as_stream(stream)>>>(x, reinterpret_cast<_Float16 *>(y_hi), reinterpret_cast<_Float16 *>(y_lo),
                                                                     (int)C, S, x_batch_stride ? x_batch_stride : C * S,
                                                                     y_batch_stride ? y_batch_stride : 2 * G * 8 * S, mul, mul_dev);
    return check_launch("snvc_f16x3_from_ncdhw");
}

int snvc_f16x3_affine_from_ncdhw(const float *x, const float *scale, const float *shift, const void *res_hi, const void *res_lo,
                                 void *y_hi, void *y_lo, int *overflow, int64_t N, int64_t C, int64_t S, int64_t x_batch_stride,
                                 int64_t y_batch_stride, int64_t res_batch_stride, int per_sample, int flags, float out_mul,
                                 float res_mul, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || S < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_affine_from_ncdhw: bad sizes");
    if (flags & ~(SNVC_EPI_RELU | SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_affine_from_ncdhw: RELU / ADD_PRE / ADD_POST only");
    if ((flags & SNVC_EPI_ADD_PRE) && (flags & SNVC_EPI_ADD_POST)) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_affine_from_ncdhw: one residual position");
    if (N == 0 || S == 0) return SNVC_OK;
    const int64_t G = ceil_div<int64_t>(C, 8);
    if (!x || !scale || !shift || !y_hi || !y_lo || ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 15))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_affine_from_ncdhw: null or unaligned pointer");
    const bool want_res = (flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST)) != 0;
    if (want_res != (res_hi != nullptr) || (res_hi == nullptr) != (res_lo == nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_affine_from_ncdhw: a residual pair exactly when ADD_PRE / ADD_POST is set");
    if (res_hi && ((reinterpret_cast<uintptr_t>(res_hi) | reinterpret_cast<uintptr_t>(res_lo)) & 15))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_affine_from_ncdhw: unaligned residual");
    if (!grid_ok(S, G, N)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_affine_from_ncdhw: tensor too large");
    dim3 grid((unsigned)ceil_div<int64_t>(S, 256), (unsigned)G, (unsigned)N);
    ncdhw_affine_to_c8_split_kernel<<<grid, 256, 0, as_stream(stream)>>>(
        x, scale, shift, reinterpret_cast<const _Float16 *>(res_hi), reinterpret_cast<const _Float16 *>(res_lo), reinterpret_cast<_Float16 *>(y_hi),
        reinterpret_cast<_Float16 *>(y_lo), overflow, (int)C, S, x_batch_stride ? x_batch_stride : C * S,
        y_batch_stride ? y_batch_stride : 2 * G * 8 * S, res_batch_stride ? res_batch_stride : 2 * G * 8 * S, per_sample, flags, out_mul, res_mul);
    return check_launch("snvc_f16x3_affine_from_ncdhw");
}

int snvc_f16x3_to_ncdhw(const void *x_hi, const void *x_lo, float *y, int64_t N, int64_t C, int64_t S, int64_t x_batch_stride,
                        int64_t y_batch_stride, float mul, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || S < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_to_ncdhw: bad sizes");
    if (N == 0 || S == 0) return SNVC_OK;
    const int64_t G = ceil_div<int64_t>(C, 8);
    if (!x_hi || !x_lo || !y || ((reinterpret_cast<uintptr_t>(x_hi) | reinterpret_cast<uintptr_t>(x_lo)) & 15))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_to_ncdhw: null or unaligned pointer");
    if (!grid_ok(S, G, N)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_to_ncdhw: tensor too large");
    dim3 grid((unsigned)ceil_div<int64_t>(S, 256), (unsigned)G, (unsigned)N);
    c8_split_to_ncdhw_f32_kernel<<<grid, 256, 0, as_stream(stream)>>>(reinterpret_cast<const _Float16 *>(x_hi), reinterpret_cast<const _Float16 *>(x_lo),
                                                                     y, (int)C, S, x_batch_stride ? x_batch_stride : 2 * G * 8 * S,
                                                                     y_batch_stride ? y_batch_stride : C * S, mul);
    return check_launch("snvc_f16x3_to_ncdhw");
}

int snvc_f16_mul_broadcast(const void *feat, const float *occ, void *out, int64_t N, int64_t C, int64_t S,
                           int64_t feat_batch_stride, int64_t out_batch_stride, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || C % 8 || S < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_mul_broadcast: bad sizes (C % 8 == 0)");
    if (N == 0 || S == 0) return SNVC_OK;
    if (!feat || !occ || !out || ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out)) & 15))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_mul_broadcast: null or unaligned pointer");
    if (!grid_ok(S, C / 8, N)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_mul_broadcast: tensor too large");
    dim3 grid((unsigned)ceil_div<int64_t>(S, 256), (unsigned)(C / 8), (unsigned)N);
    mul_broadcast_c8_kernel<<<grid, 256, 0, as_stream(stream)>>>(reinterpret_cast<const _Float16 *>(feat), occ,
                                                                reinterpret_cast<_Float16 *>(out), S,
                                                                feat_batch_stride ? feat_batch_stride : C * S,
                                                                out_batch_stride ? out_batch_stride : C * S);
    return check_launch("snvc_f16_mul_broadcast");
}

int snvc_f16x3_mul_broadcast(const void *feat_hi, const void *feat_lo, const float *occ, void *out_hi, void *out_lo, int64_t N,
                             int64_t C, int64_t S, int64_t feat_batch_stride, int64_t out_batch_stride, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || C % 8 || S < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_mul_broadcast: bad sizes (C % 8 == 0)");
    if (N == 0 || S == 0) return SNVC_OK;
    if (!feat_hi || !feat_lo || !occ || !out_hi || !out_lo ||
        ((reinterpret_cast<uintptr_t>(feat_hi) | reinterpret_cast<uintptr_t>(feat_lo) | reinterpret_cast<uintptr_t>(out_hi) |
          reinterpret_cast<uintptr_t>(out_lo)) & 15))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_mul_broadcast: null or unaligned pointer");
    if (!grid_ok(S, C / 8, N)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_mul_broadcast: tensor too large");
    dim3 grid((unsigned)ceil_div<int64_t>(S, 256), (unsigned)(C / 8), (unsigned)N);
    mul_broadcast_c8_split_kernel<<<grid, 256, 0, as_stream(stream)>>>(
        reinterpret_cast<const _Float16 *>(feat_hi), reinterpret_cast<const _Float16 *>(feat_lo), occ, reinterpret_cast<_Float16 *>(out_hi),
        reinterpret_cast<_Float16 *>(out_lo), S, feat_batch_stride ? feat_batch_stride : 2 * C * S, out_batch_stride ? out_batch_stride : 2 * C * S);
    return check_launch("snvc_f16x3_mul_broadcast");
}

int snvc_f16_avgpool_depth4(const void *x, float *y, int64_t N, int64_t C, int64_t D, int64_t HW, int64_t x_batch_stride,
                            void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || D < 0 || HW < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_avgpool_depth4: bad sizes");
    const int64_t G = ceil_div<int64_t>(C, 8), out = (D / 4) * HW;
    if (N == 0 || out == 0) return SNVC_OK;
    if (!x || !y || (reinterpret_cast<uintptr_t>(x) & 15)) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_avgpool_depth4: null or unaligned pointer");
    if (!grid_ok(out, G, N)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_avgpool_depth4: tensor too large");
    dim3 grid((unsigned)ceil_div<int64_t>(out, 256), (unsigned)G, (unsigned)N);
    avgpool_depth4_c8_kernel<<<grid, 256, 0, as_stream(stream)>>>(reinterpret_cast<const _Float16 *>(x), y, (int)C, (int)D, HW,
                                                                 x_batch_stride ? x_batch_stride : G * 8 * D * HW);
    return check_launch("snvc_f16_avgpool_depth4");
}

}  // extern "C"
