// HBM-bound glue steps of the 3D trunk and the
// normalisation statistics for GroupNorm / train-mode BatchNorm (row a4).
// All kernels stream with 16-byte accesses when the row length allows it and fall back to
// scalar lanes otherwise; none of them re-reads its input.
#include "common.hpp"
#include "elementwise_internal.hpp"

#include <cfloat>
#include <cstdlib>
#include <initializer_list>

namespace snvc {
namespace {

#pragma clang fp contract(off)

__device__ __forceinline__ float epilogue(float v, float res, int flags) {
    if (flags & SNVC_EPI_ADD_PRE) v = v + res;
    if (flags & SNVC_EPI_RELU) v = v > 0.0f ? v : 0.0f;  // NaN -> 0 differs from torch only for NaN inputs
    if (flags & SNVC_EPI_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    if (flags & SNVC_EPI_ADD_POST) v = v + res;
    return v;
}

// out[n,c,s] = feat[n,c,s] * occ[n,s]
__global__ void __launch_bounds__(256)
mul_broadcast_kernel(const float *__restrict__ feat, const float *__restrict__ occ,
                     float *__restrict__ out, int64_t C, int64_t S, int64_t out_bs) {
    const int64_t n = blockIdx.z, c = blockIdx.y;
    const float *f = feat + (n * C + c) * S;
    const float *o = occ + n * S;
    float *y = out + n * out_bs + c * S;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = ((S & 3) == 0) && ((((uintptr_t)f | (uintptr_t)o | (uintptr_t)y) & 15) == 0);
    if (vec) {
        const int64_t S4 = S >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S4; i += stride) {
            const float4 a = reinterpret_cast<const float4 *>(f)[i];
            const float4 b = reinterpret_cast<const float4 *>(o)[i];
            reinterpret_cast<float4 *>(y)[i] = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += stride) y[i] = f[i] * o[i];
    }
}

// AvgPool3d((4,1,1)): y[p, dq, i] = ((x[4dq]+x[4dq+1])+x[4dq+2])+x[4dq+3]) / 4, p = n*C + c
__global__ void __launch_bounds__(256)
avgpool_depth4_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t D, int64_t HW) {
    const int64_t p = blockIdx.z, dq = blockIdx.y, Dq = D / 4;
    const float *a = x + (p * D + dq * 4) * HW;
    float *o = y + (p * Dq + dq) * HW;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = ((HW & 3) == 0) && ((((uintptr_t)a | (uintptr_t)o) & 15) == 0);
    if (vec) {
        const int64_t n4 = HW >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const float4 v0 = reinterpret_cast<const float4 *>(a)[i];
            const float4 v1 = reinterpret_cast<const float4 *>(a + HW)[i];
            const float4 v2 = reinterpret_cast<const float4 *>(a + 2 * HW)[i];
            const float4 v3 = reinterpret_cast<const float4 *>(a + 3 * HW)[i];
            float4 r;
            r.x = (((v0.x + v1.x) + v2.x) + v3.x) / 4.0f;
            r.y = (((v0.y + v1.y) + v2.y) + v3.y) / 4.0f;
            r.z = (((v0.z + v1.z) + v2.z) + v3.z) / 4.0f;
            r.w = (((v0.w + v1.w) + v2.w) + v3.w) / 4.0f;
            reinterpret_cast<float4 *>(o)[i] = r;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += stride)
            o[i] = (((a[i] + a[HW + i]) + a[2 * HW + i]) + a[3 * HW + i]) / 4.0f;
    }
}

// adjoint of avgpool_depth4_kernel: gx[p, 4 dq + j, i] = gy[p, dq, i] / 4 for j = 0..3; depth planes beyond 4 * (D / 4) (the floor of
// AvgPool3d without ceil_mode drops them) get zero
__global__ void __launch_bounds__(256)
avgpool_depth4_bwd_kernel(const float *__restrict__ gy, float *__restrict__ gx, int64_t D, int64_t HW) {
    const int64_t p = blockIdx.z, d = blockIdx.y, Dq = D / 4;
    const bool live = d < 4 * Dq;
    const float *a = gy + (p * Dq + (live ? d / 4 : 0)) * HW;
    float *o = gx + (p * D + d) * HW;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += stride) o[i] = live ? a[i] / 4.0f : 0.0f;
}

// out[n,i] = sum_d x[n,d,i] * depth[d], d ascending (torch.sum over dim 1 of a product)
__global__ void __launch_bounds__(256)
disparity_regression_kernel(const float *__restrict__ x, const float *__restrict__ depth,
                            float *__restrict__ out, int64_t D, int64_t HW) {
    const int64_t n = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float *p = x + n * D * HW + i;
    float acc = 0.0f;
    for (int64_t d = 0; d < D; ++d) acc = acc + p[d * HW] * depth[d];
    out[n * HW + i] = acc;
}

// Row-wise first-maximum argmax, numpy semantics (a NaN is treated as the maximum and the
// first NaN wins).  One 256-thread workgroup per row: strided scan, wave shuffle reduction,
// then a 4-entry LDS combine.
__device__ __forceinline__ bool better(float va, int64_t ia, float vb, int64_t ib) {
    // true if (va, ia) should replace (vb, ib)
    const bool a_nan = va != va, b_nan = vb != vb;
    if (a_nan || b_nan) {
        if (a_nan && b_nan) return ia < ib;
        return a_nan;
    }
    if (va > vb) return true;
    if (va < vb) return false;
    return ia < ib;
}

__global__ void __launch_bounds__(256)
argmax_rows_kernel(const float *__restrict__ x, int64_t *__restrict__ idx, float *__restrict__ val,
                   int64_t L) {
    const int64_t r = blockIdx.x;
    const float *row = x + r * L;
    float bv = -INFINITY;
    int64_t bi = INT64_MAX;
    for (int64_t i = threadIdx.x; i < L; i += blockDim.x) {
        const float v = row[i];
        if (bi == INT64_MAX || better(v, i, bv, bi)) { bv = v; bi = i; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(bv, off, 64);
        const int64_t oi = __shfl_down(bi, off, 64);
        if (oi != INT64_MAX && (bi == INT64_MAX || better(ov, oi, bv, bi))) { bv = ov; bi = oi; }
    }
    __shared__ float sv[4];
    __shared__ int64_t si[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k)
            if (si[k] != INT64_MAX && (bi == INT64_MAX || better(sv[k], si[k], bv, bi))) { bv = sv[k]; bi = si[k]; }
        idx[r] = bi;
        if (val) val[r] = bv;
    }
}

// ---------------------------------------------------------------------------- norm statistics
// Pass 1: partial (sum, sum of squares) in fp64 per (row, split); a "row" is one contiguous
// run of `len` floats: (n, group) for GroupNorm; (n, c) for BatchNorm (combined over n later).
// VEC = 4: 16-byte loads and four independent fp64 accumulator pairs per thread (rows and strides of whole float4: the
// scalar form, one dependent fp64 chain per thread, read 2.4 TB/s on the 736 MB layers of cfg4); VEC = 1 otherwise.
template <int VEC>
__global__ void __launch_bounds__(256)
norm_partial_kernel(const float *__restrict__ x, double *__restrict__ partial, int64_t rows_per_n,
                    int64_t len, int64_t x_bs, int splits) {
    typedef float fv __attribute__((ext_vector_type(VEC)));
    const int64_t row = blockIdx.y;
    const int split = blockIdx.x;
    const int64_t n = row / rows_per_n, g = row % rows_per_n;
    const fv *p = reinterpret_cast<const fv *>(x + n * x_bs + g * len);
    const int64_t lenv = len / VEC;
    const int64_t chunk = ceil_div<int64_t>(lenv, splits);
    const int64_t lo = split * chunk, hi = (lo + chunk < lenv) ? lo + chunk : lenv;
    double sv[VEC], ssv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sv[j] = 0.0; ssv[j] = 0.0; }
    // r6: two loads in flight per thread (32 splits x rows workgroups = 16 waves per CU: with one 16-byte load each the pass moved
    // 4.1 TB/s on cfg4's 736 MB layers); the order of the additions into each accumulator is unchanged
    int64_t i = lo + threadIdx.x;
    for (; i + blockDim.x < hi; i += 2 * blockDim.x) {
        const fv q0 = p[i], q1 = p[i + blockDim.x];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const double v = (double)q0[j];
            sv[j] += v;
            ssv[j] += v * v;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const double v = (double)q1[j];
            sv[j] += v;
            ssv[j] += v * v;
        }
    }
    for (; i < hi; i += blockDim.x) {
        const fv q = p[i];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const double v = (double)q[j];
            sv[j] += v;
            ssv[j] += v * v;
        }
    }
    double s = sv[0], ss = ssv[0];
    if constexpr (VEC == 4) { s = (sv[0] + sv[1]) + (sv[2] + sv[3]); ss = (ssv[0] + ssv[1]) + (ssv[2] + ssv[3]); }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        ss += __shfl_down(ss, off, 64);
    }
    __shared__ double sh[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh[2 * wave] = s; sh[2 * wave + 1] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { s += sh[2 * k]; ss += sh[2 * k + 1]; }
        partial[(row * splits + split) * 2 + 0] = s;
        partial[(row * splits + split) * 2 + 1] = ss;
    }
}

// Pass 2: one thread per (n or 0, channel): fold partials -> mean/var of its group -> scale/shift.
__global__ void norm_finalize_kernel(const double *__restrict__ partial, const float *__restrict__ gamma,
                                     const float *__restrict__ beta, float *__restrict__ scale,
                                     float *__restrict__ shift, float *__restrict__ mean_out,
                                     float *__restrict__ var_out, int64_t N, int64_t C, int64_t S,
                                     int64_t groups, int per_sample, int splits, float eps) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t outer = per_sample ? N : 1;
    if (t >= outer * C) return;
    const int64_t n = t / C, c = t % C;
    const int64_t cpg = C / groups, g = c / cpg;
    double s = 0.0, ss = 0.0, count;
    if (per_sample) {
        const int64_t row = n * groups + g;
        for (int k = 0; k < splits; ++k) { s += partial[(row * splits + k) * 2]; ss += partial[(row * splits + k) * 2 + 1]; }
        count = (double)cpg * (double)S;
    } else {  // batch statistics: groups == C, fold over the batch
        for (int64_t b = 0; b < N; ++b) {
            const int64_t row = b * C + c;
            for (int k = 0; k < splits; ++k) { s += partial[(row * splits + k) * 2]; ss += partial[(row * splits + k) * 2 + 1]; }
        }
        count = (double)N * (double)S;
    }
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
    scale[t] = (float)(rstd * ga);
    shift[t] = (float)(be - mean * rstd * ga);
    if (c % cpg == 0) {
        if (mean_out) mean_out[n * groups + g] = (float)mean;
        if (var_out) var_out[n * groups + g] = (float)var;
    }
}

// |v| as bits: for non-negative floats the unsigned order is the float order (a NaN sorts above infinity); one atomicMax per wave
__device__ __forceinline__ unsigned amax_bits4(unsigned m, float a, float b, float c, float d) {
    const unsigned b0 = __float_as_uint(a) & 0x7fffffffu, b1 = __float_as_uint(b) & 0x7fffffffu;
    const unsigned b2 = __float_as_uint(c) & 0x7fffffffu, b3 = __float_as_uint(d) & 0x7fffffffu;
    const unsigned m01 = b0 > b1 ? b0 : b1, m23 = b2 > b3 ? b2 : b3;
    const unsigned mm = m01 > m23 ? m01 : m23;
    return m > mm ? m : mm;
}
__device__ __forceinline__ void amax_publish(unsigned *amax, unsigned m) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, off);
        m = m > o ? m : o;
    }
    // SNVC_AMAX_SLOTS words, the workgroup picks one: thousands of atomics on ONE address serialise in the L2 (measured: +0.06 ms per
    // launch of these passes at cfg2 size with a single word); the consumer takes the maximum of the slots
    if ((threadIdx.x & 63) == 0 && m) atomicMax(amax + ((blockIdx.x + 7u * blockIdx.y + 13u * blockIdx.z) & (SNVC_AMAX_SLOTS - 1)), m);
}

__global__ void __launch_bounds__(256)
affine_act_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                  const float *__restrict__ shift, const float *__restrict__ res,
                  float *__restrict__ y, int64_t C, int64_t S, int64_t x_bs, int64_t y_bs,
                  int64_t r_bs, int per_sample, int flags, unsigned *__restrict__ amax) {
    const int64_t n = blockIdx.z, c = blockIdx.y;
    unsigned mx = 0;        // bits of max|y| over this thread's elements (r6: the split-operand weight gradient scales by it)
    const float sc = scale ? scale[(per_sample ? n * C : 0) + c] : 1.0f;
    const float sh = shift ? shift[(per_sample ? n * C : 0) + c] : 0.0f;
    const float *a = x + n * x_bs + c * S;
    const float *r = res ? res + n * r_bs + c * S : nullptr;
    float *o = y + n * y_bs + c * S;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = ((S & 3) == 0) && ((((uintptr_t)a | (uintptr_t)o | (uintptr_t)r) & 15) == 0);
    if (vec) {
        const int64_t S4 = S >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S4; i += stride) {
            const float4 v = reinterpret_cast<const float4 *>(a)[i];
            float4 q = make_float4(0, 0, 0, 0);
            if (r) q = reinterpret_cast<const float4 *>(r)[i];
            float4 w;
            w.x = epilogue(v.x * sc + sh, q.x, flags);
            w.y = epilogue(v.y * sc + sh, q.y, flags);
            w.z = epilogue(v.z * sc + sh, q.z, flags);
            w.w = epilogue(v.w * sc + sh, q.w, flags);
            reinterpret_cast<float4 *>(o)[i] = w;
            mx = amax_bits4(mx, w.x, w.y, w.z, w.w);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += stride) {
            const float w = epilogue(a[i] * sc + sh, r ? r[i] : 0.0f, flags);
            o[i] = w;
            mx = amax_bits4(mx, w, w, w, w);
        }
    }
    if (amax) amax_publish(amax, mx);
}


// ---------------------------------------------------------------------------- backward of the fused epilogue
// Forward (per element, channel c):  v = raw*sc + sh;  [v += res];  y = act(v);  [y += res]
// g = dL/dv = gy * act'(v).  Training BatchNorm / GroupNorm additionally need, per statistics
// group, sum(g) and sum(g*raw) -- accumulated here in fp64 exactly like the forward statistics.
__device__ __forceinline__ float act_grad(float raw, float gy, float res, float sc, float sh, int flags) {
    float v = raw * sc + sh;
    if (flags & SNVC_EPI_ADD_PRE) v = v + res;
    float g = gy;
    if (flags & SNVC_EPI_RELU) g = v > 0.0f ? g : 0.0f;
    if (flags & SNVC_EPI_SIGMOID) { const float s = 1.0f / (1.0f + expf(-v)); g = g * (s * (1.0f - s)); }
    return g;
}

template <int VEC>     // as norm_partial_kernel: 16-byte loads and four independent accumulator pairs when the rows allow it
__global__ void __launch_bounds__(256)
act_bwd_partial_kernel(const float *__restrict__ raw, const float *__restrict__ gy, const float *__restrict__ res,
                       const float *__restrict__ scale, const float *__restrict__ shift, double *__restrict__ partial,
                       int64_t C, int64_t S, int64_t raw_bs, int64_t gy_bs, int64_t r_bs, int per_sample, int flags,
                       int splits, unsigned *__restrict__ amax_gy) {
    typedef float fv __attribute__((ext_vector_type(VEC)));
    unsigned mx = 0;                           // r6: bits of max|gy| (the split twin of draw is scaled by a bound built from it)
    const int64_t row = blockIdx.y;            // n*C + c
    const int split = blockIdx.x;
    const int64_t n = row / C, c = row % C;
    const float sc = scale ? scale[(per_sample ? n * C : 0) + c] : 1.0f;
    const float sh = shift ? shift[(per_sample ? n * C : 0) + c] : 0.0f;
    const fv *a = reinterpret_cast<const fv *>(raw + n * raw_bs + c * S), *b = reinterpret_cast<const fv *>(gy + n * gy_bs + c * S);
    const fv *r = res ? reinterpret_cast<const fv *>(res + n * r_bs + c * S) : nullptr;
    const int64_t Sv = S / VEC;
    const int64_t chunk = ceil_div<int64_t>(Sv, splits);
    const int64_t lo = split * chunk, hi = (lo + chunk < Sv) ? lo + chunk : Sv;
    double s0v[VEC], s1v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s0v[j] = 0.0; s1v[j] = 0.0; }
    auto step = [&](const fv av, const fv bv, const fv rv) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float aj = av[j], bj = bv[j], rj = r ? rv[j] : 0.0f;
            const float g = act_grad(aj, bj, rj, sc, sh, flags);
            s0v[j] += (double)g;
            s1v[j] += (double)g * (double)aj;
        }
        if (amax_gy) {
            if constexpr (VEC == 4) mx = amax_bits4(mx, bv[0], bv[1], bv[2], bv[3]);
            else mx = amax_bits4(mx, bv[0], bv[0], bv[0], bv[0]);
        }
    };
    // r6: two iterations' loads in flight (see norm_partial_kernel); accumulation order unchanged
    int64_t i = lo + threadIdx.x;
    for (; i + blockDim.x < hi; i += 2 * blockDim.x) {
        const fv a0 = a[i], b0 = b[i], a1 = a[i + blockDim.x], b1 = b[i + blockDim.x];
        fv r0 = a0, r1 = a1;
        if (r) { r0 = r[i]; r1 = r[i + blockDim.x]; }
        step(a0, b0, r0);
        step(a1, b1, r1);
    }
    for (; i < hi; i += blockDim.x) {
        const fv av = a[i], bv = b[i];
        fv rv = av;
        if (r) rv = r[i];
        step(av, bv, rv);
    }
    if (amax_gy) amax_publish(amax_gy, mx);
    double s0 = s0v[0], s1 = s1v[0];
    if constexpr (VEC == 4) { s0 = (s0v[0] + s0v[1]) + (s0v[2] + s0v[3]); s1 = (s1v[0] + s1v[1]) + (s1v[2] + s1v[3]); }
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
    }
    __shared__ double shm[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { shm[2 * wave] = s0; shm[2 * wave + 1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { s0 += shm[2 * k]; s1 += shm[2 * k + 1]; }
        partial[(row * splits + split) * 2 + 0] = s0;
        partial[(row * splits + split) * 2 + 1] = s1;
    }
}

// sums[row][2] = fold of the splits (row = n*C + c), fixed order
__global__ void act_bwd_fold_kernel(const double *__restrict__ partial, double *__restrict__ sums, int64_t rows,
                                    int splits) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < splits; ++k) { s0 += partial[(row * splits + k) * 2]; s1 += partial[(row * splits + k) * 2 + 1]; }
    sums[row * 2] = s0;
    sums[row * 2 + 1] = s1;
}

// draw = A*g + B*raw + Cc per (n?, c);  g_out (optional) = g
__global__ void __launch_bounds__(256)
act_bwd_apply_kernel(const float *__restrict__ raw, const float *__restrict__ gy, const float *__restrict__ res,
                     const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ A,
                     const float *__restrict__ B, const float *__restrict__ Cc, float *__restrict__ draw,
                     float *__restrict__ g_out, int64_t C, int64_t S, int64_t raw_bs, int64_t gy_bs, int64_t r_bs,
                     int per_sample, int flags, unsigned *__restrict__ amax) {
    const int64_t n = blockIdx.z, c = blockIdx.y;
    const int64_t pc = (per_sample ? n * C : 0) + c;
    const float sc = scale ? scale[pc] : 1.0f, sh = shift ? shift[pc] : 0.0f;
    const float ca = A[pc], cb = B ? B[pc] : 0.0f, cc = Cc ? Cc[pc] : 0.0f;
    const float *a = raw + n * raw_bs + c * S, *b = gy + n * gy_bs + c * S;
    const float *r = res ? res + n * r_bs + c * S : nullptr;
    float *o = draw + (n * C + c) * S;
    float *go = g_out ? g_out + (n * C + c) * S : nullptr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned mx = 0;
    // r6: 16-byte pieces when the rows allow it (the pass moved 3.7 TB/s with 4-byte accesses: 1.7 ms of the cfg4 step)
    const bool vec = ((S & 3) == 0) && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)o | (uintptr_t)r | (uintptr_t)go) & 15) == 0);
    if (vec) {
        const int64_t S4 = S >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S4; i += stride) {
            const float4 x = reinterpret_cast<const float4 *>(a)[i], gy4 = reinterpret_cast<const float4 *>(b)[i];
            float4 q = make_float4(0, 0, 0, 0);
            if (r) q = reinterpret_cast<const float4 *>(r)[i];
            float4 g, dr;
            g.x = act_grad(x.x, gy4.x, q.x, sc, sh, flags); g.y = act_grad(x.y, gy4.y, q.y, sc, sh, flags);
            g.z = act_grad(x.z, gy4.z, q.z, sc, sh, flags); g.w = act_grad(x.w, gy4.w, q.w, sc, sh, flags);
            dr.x = ca * g.x + cb * x.x + cc; dr.y = ca * g.y + cb * x.y + cc;
            dr.z = ca * g.z + cb * x.z + cc; dr.w = ca * g.w + cb * x.w + cc;
            reinterpret_cast<float4 *>(o)[i] = dr;
            mx = amax_bits4(mx, dr.x, dr.y, dr.z, dr.w);
            if (go) reinterpret_cast<float4 *>(go)[i] = g;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += stride) {
            const float x = a[i];
            const float g = act_grad(x, b[i], r ? r[i] : 0.0f, sc, sh, flags);
            const float dr = ca * g + cb * x + cc;
            o[i] = dr;
            mx = amax_bits4(mx, dr, dr, dr, dr);
            if (go) go[i] = g;
        }
    }
    if (amax) amax_publish(amax, mx);
}

// ---------------------------------------------------------------------------- r6: passes that also write a split C8 twin
// The training step's half- / quarter-resolution and transposed layers run their forward and data-gradient convolutions on the split
// kernels (csrc/conv3d_f16.hip), which read [N][2 (hi | lo)][C/8][S][8] half pairs.  The pass that produces the tensor in float32
// NCDHW (for the weight gradient, the statistics and autograd) writes that pair as well: +4 bytes per element on a pass that moves 8
// or 12, instead of a layout pass of its own (8 bytes per element, measured 0.22-0.25 ms at half resolution).  One thread = 4
// consecutive voxels x the 8 channels of a group; twin value = v * mul[0] (a power of two from an upper bound of max|v|, see
// split_scale_bound_kernel: no clamp needed), hi = half(v * mul), lo = half(v * mul - hi).
typedef _Float16 tw_h8 __attribute__((ext_vector_type(8)));

// A lane holds the 4 pieces (16 bytes each) of its 4 voxels: stored directly, every store instruction would touch 64 lanes x 16 bytes at
// a 64-byte stride (measured: the full-resolution apply pass at 3.4 TB/s instead of 5.3).  The wave transposes through its own 8 KB of
// LDS instead -- piece (4 lane + k) written, piece (64 j + lane) read back -- so that a store instruction covers 1 KB of consecutive
// bytes.  `full`: all 64 lanes of the wave have voxels (the tail wave stores directly).
__device__ __forceinline__ void twin_store(_Float16 *__restrict__ th, _Float16 *__restrict__ tl, int64_t piece, const float (&w)[8][4], float m,
                                           tw_h8 *__restrict__ lds_wave, bool full) {
    const int lane = threadIdx.x & 63;
    tw_h8 hi[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = w[e][k] * m;
            hi[k][e] = (_Float16)f;
            lo[k][e] = (_Float16)(f - (float)hi[k][e]);
        }
    }
    if (!full) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            *reinterpret_cast<tw_h8 *>(th + (piece + k) * 8) = hi[k];
            *reinterpret_cast<tw_h8 *>(tl + (piece + k) * 8) = lo[k];
        }
        return;
    }
    // rotate the piece slot by the lane's quad so that the four 16-byte writes of four neighbouring lanes fall into different banks
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lds_wave[4 * lane + ((k + (lane >> 2)) & 3)] = hi[k];
        lds_wave[256 + 4 * lane + ((k + (lane >> 2)) & 3)] = lo[k];
    }
    const int64_t base = piece - 4 * lane;          // the wave's first piece
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = 64 * j + lane;                // piece p = voxel-lane p >> 2, slot k = p & 3
        const int src = (p & ~3) + (((p & 3) + ((p >> 2) >> 2)) & 3);
        *reinterpret_cast<tw_h8 *>(th + (base + p) * 8) = lds_wave[src];
        *reinterpret_cast<tw_h8 *>(tl + (base + p) * 8) = lds_wave[256 + src];
    }
}

__global__ void __launch_bounds__(256)
affine_act_twin_kernel(const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                       const float *__restrict__ res, float *__restrict__ y, _Float16 *__restrict__ t_hi, _Float16 *__restrict__ t_lo,
                       const float *__restrict__ mul, int64_t C, int64_t S, int64_t x_bs, int64_t y_bs, int64_t r_bs, int64_t t_bs,
                       int per_sample, int flags, unsigned *__restrict__ amax) {
    const int64_t n = blockIdx.z, g = blockIdx.y, c0 = 8 * g;
    const float m = mul[0];
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = scale ? scale[(per_sample ? n * C : 0) + c0 + e] : 1.0f;
        sh[e] = shift ? shift[(per_sample ? n * C : 0) + c0 + e] : 0.0f;
    }
    const float *a = x + n * x_bs + c0 * S;
    const float *r = res ? res + n * r_bs + c0 * S : nullptr;
    float *o = y + n * y_bs + c0 * S;
    _Float16 *th = t_hi + n * t_bs + g * S * 8, *tl = t_lo + n * t_bs + g * S * 8;
    const int64_t S4 = S >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    unsigned mx = 0;
    __shared__ tw_h8 twin_lds[4][512];
    tw_h8 *lds_wave = twin_lds[threadIdx.x >> 6];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S4; i += stride) {
        const bool full = (i - (threadIdx.x & 63)) + 64 <= S4;
        float4 v[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = reinterpret_cast<const float4 *>(a + e * S)[i];
        if (r) {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = reinterpret_cast<const float4 *>(r + e * S)[i];
        }
        float w[8][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float4 qq = r ? q[e] : make_float4(0, 0, 0, 0);
            w[e][0] = epilogue(v[e].x * sc[e] + sh[e], qq.x, flags);
            w[e][1] = epilogue(v[e].y * sc[e] + sh[e], qq.y, flags);
            w[e][2] = epilogue(v[e].z * sc[e] + sh[e], qq.z, flags);
            w[e][3] = epilogue(v[e].w * sc[e] + sh[e], qq.w, flags);
            reinterpret_cast<float4 *>(o + e * S)[i] = make_float4(w[e][0], w[e][1], w[e][2], w[e][3]);
            mx = amax_bits4(mx, w[e][0], w[e][1], w[e][2], w[e][3]);
        }
        twin_store(th, tl, 4 * i, w, m, lds_wave, full);
    }
    if (amax) amax_publish(amax, mx);
}

// draw = A*g + B*raw + Cc per (n?, c) as act_bwd_apply_kernel, plus draw's split twin (what the data-gradient convolution reads)
__global__ void __launch_bounds__(256)
act_bwd_apply_twin_kernel(const float *__restrict__ raw, const float *__restrict__ gy, const float *__restrict__ res,
                          const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ A,
                          const float *__restrict__ B, const float *__restrict__ Cc, float *__restrict__ draw,
                          float *__restrict__ g_out, _Float16 *__restrict__ t_hi, _Float16 *__restrict__ t_lo,
                          const float *__restrict__ mul, int64_t C, int64_t S, int64_t raw_bs, int64_t gy_bs, int64_t r_bs, int64_t t_bs,
                          int per_sample, int flags, unsigned *__restrict__ amax) {
    const int64_t n = blockIdx.z, g = blockIdx.y, c0 = 8 * g;
    const int64_t pc = (per_sample ? n * C : 0) + c0;
    const float m = mul[0];
    float sc[8], sh[8], ca[8], cb[8], cc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = scale ? scale[pc + e] : 1.0f;
        sh[e] = shift ? shift[pc + e] : 0.0f;
        ca[e] = A[pc + e];
        cb[e] = B ? B[pc + e] : 0.0f;
        cc[e] = Cc ? Cc[pc + e] : 0.0f;
    }
    const float *a = raw + n * raw_bs + c0 * S, *b = gy + n * gy_bs + c0 * S;
    const float *r = res ? res + n * r_bs + c0 * S : nullptr;
    float *o = draw + (n * C + c0) * S;
    float *go = g_out ? g_out + (n * C + c0) * S : nullptr;
    _Float16 *th = t_hi + n * t_bs + g * S * 8, *tl = t_lo + n * t_bs + g * S * 8;
    const int64_t S4 = S >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    unsigned mx = 0;
    __shared__ tw_h8 twin_lds[4][512];
    tw_h8 *lds_wave = twin_lds[threadIdx.x >> 6];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S4; i += stride) {
        const bool full = (i - (threadIdx.x & 63)) + 64 <= S4;
        float4 xv[8], gv[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xv[e] = reinterpret_cast<const float4 *>(a + e * S)[i];
            gv[e] = reinterpret_cast<const float4 *>(b + e * S)[i];
        }
        if (r) {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = reinterpret_cast<const float4 *>(r + e * S)[i];
        }
        float w[8][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float4 qq = r ? q[e] : make_float4(0, 0, 0, 0);
            float4 gg;
            gg.x = act_grad(xv[e].x, gv[e].x, qq.x, sc[e], sh[e], flags); gg.y = act_grad(xv[e].y, gv[e].y, qq.y, sc[e], sh[e], flags);
            gg.z = act_grad(xv[e].z, gv[e].z, qq.z, sc[e], sh[e], flags); gg.w = act_grad(xv[e].w, gv[e].w, qq.w, sc[e], sh[e], flags);
            w[e][0] = ca[e] * gg.x + cb[e] * xv[e].x + cc[e]; w[e][1] = ca[e] * gg.y + cb[e] * xv[e].y + cc[e];
            w[e][2] = ca[e] * gg.z + cb[e] * xv[e].z + cc[e]; w[e][3] = ca[e] * gg.w + cb[e] * xv[e].w + cc[e];
            reinterpret_cast<float4 *>(o + e * S)[i] = make_float4(w[e][0], w[e][1], w[e][2], w[e][3]);
            mx = amax_bits4(mx, w[e][0], w[e][1], w[e][2], w[e][3]);
            if (go) reinterpret_cast<float4 *>(go + e * S)[i] = gg;
        }
        twin_store(th, tl, 4 * i, w, m, lds_wave, full);
    }
    if (amax) amax_publish(amax, mx);
}

// The scale of a twin from an UPPER BOUND of max|v| known before the pass runs (one workgroup):
//   bound = max over rows r of ( |a[r]| * P + |b[r]| * L[r % C] * X + |c[r]| ) + R ,
// P / X / R the maxima of SNVC_AMAX_SLOTS-word amax arrays (a NULL array counts 0; a NULL vector counts 0 for a, c and 1 for b, L),
// mul = the power of two that puts bound into [2^13, 2^14) (1 for a zero or non-finite bound).
//   forward  y = act(scale * raw + shift [+ res]) [+ res]:  b = scale, L = the L1 norms of the layer's filters, X = max|x| (so that
//            L * X bounds |raw|), c = shift, R = max|res|;
//   backward draw = A * g + B * raw + Cc:  a = A, P = max|gy|, b = B, L, X as forward, c = Cc.
// A loose bound costs nothing until the pair's lo part leaves half's subnormal range: 2^-25 of the bound absolute, below float32's own
// rounding for a bound up to 2^10 times the true maximum.
__device__ __forceinline__ float slots_max(const unsigned *__restrict__ slots) {      // one wave: a slot per lane
    if (!slots) return 0.0f;
    unsigned m = slots[threadIdx.x & (SNVC_AMAX_SLOTS - 1)];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, off);
        m = m > o ? m : o;
    }
    return __uint_as_float(m);
}

__global__ void __launch_bounds__(256)
split_scale_bound_kernel(const float *__restrict__ a, const unsigned *__restrict__ P, const float *__restrict__ b,
                         const float *__restrict__ L, const unsigned *__restrict__ X, const float *__restrict__ c,
                         const unsigned *__restrict__ R, int rows, int C, float *__restrict__ mul_out) {
    const float Pv = slots_max(P), Xv = X ? slots_max(X) : 1.0f, Rv = slots_max(R);
    float m = 0.0f;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const float v = (a ? fabsf(a[r]) * Pv : 0.0f) + (b ? fabsf(b[r]) : 1.0f) * (L ? L[r % C] : 1.0f) * Xv + (c ? fabsf(c[r]) : 0.0f);
        m = v > m || v != v ? v : m;          // a NaN sticks
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(m, off);
        m = (o > m || o != o) ? o : m;
    }
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) m = (sm[i] > m || sm[i] != sm[i]) ? sm[i] : m;
        const float bound = m + Rv;
        float mul = 1.0f;
        if (bound > 0.0f && bound < INFINITY) {
            int e;
            frexpf(bound, &e);               // bound = f * 2^e, f in [0.5, 1)
            int k = 14 - e;
            k = k < -24 ? -24 : (k > 40 ? 40 : k);
            mul = ldexpf(1.0f, k);
        }
        mul_out[0] = mul;
    }
}

// Train-mode BatchNorm backward coefficients (see snvc_bn_backward_coefs): one thread per channel, fp64.
__global__ void bn_bwd_coefs_kernel(const double *__restrict__ sums, const float *__restrict__ mean, const float *__restrict__ var,
                                    const float *__restrict__ gamma, float *__restrict__ coef_g, float *__restrict__ coef_raw,
                                    float *__restrict__ coef_const, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                    int64_t N, int64_t C, double count, double eps) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sg = 0.0, sgr = 0.0;
    for (int64_t n = 0; n < N; ++n) { sg += sums[(n * C + c) * 2]; sgr += sums[(n * C + c) * 2 + 1]; }
    const double mu = (double)mean[c], rstd = 1.0 / sqrt((double)var[c] + eps), gam = gamma ? (double)gamma[c] : 1.0;
    const double sgx = rstd * (sgr - mu * sg);
    const double a = gam * rstd, b = -gam * rstd * rstd * sgx / count, cc = -gam * rstd * sg / count - b * mu;
    coef_g[c] = (float)a;
    coef_raw[c] = (float)b;
    coef_const[c] = (float)cc;
    if (dgamma) dgamma[c] = (float)sgx;
    if (dbeta) dbeta[c] = (float)sg;
}

// nn.BatchNorm's train-mode bookkeeping in ONE launch (r6: it was four small torch launches per layer and step -- the counter, two
// lerp_ and the unbiasing product -- 36 launches of ~4.7 us in the cfg4 step): running <- lerp(running, batch, momentum) with the
// batch variance unbiased first, num_batches_tracked += 1.  lerp as torch evaluates it (start + w * diff for |w| < 0.5, else end - diff * (1 - w)).
__global__ void bn_track_kernel(float *__restrict__ running_mean, float *__restrict__ running_var, int64_t *__restrict__ nbt,
                                const float *__restrict__ mean, const float *__restrict__ var, int C, float momentum, float unbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) nbt[0] += 1;
    if (c >= C) return;
    auto lerp = [&](float a, float b) {
        const float diff = b - a;
        return fabsf(momentum) < 0.5f ? __builtin_fmaf(momentum, diff, a) : __builtin_fmaf(-diff, 1.0f - momentum, b);      // contracted, as torch's build
    };
    running_mean[c] = lerp(running_mean[c], mean[c]);
    running_var[c] = lerp(running_var[c], var[c] * unbias);
}

constexpr int kNormSplits = 32;

inline unsigned stream_blocks(int64_t items, int64_t outer) {
    // enough workgroups to fill 256 CUs x 8 without over-subscribing tiny rows
    int64_t b = ceil_div<int64_t>(items, 256 * 4);
    const int64_t cap = ceil_div<int64_t>(2048, outer > 0 ? outer : 1);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}


// ConvTranspose2d(k3, s2, p1, op1) = Conv2d(k3, s1, p1) with the flipped, channel-transposed kernel over the
// zero-stuffed input u[2i, 2j] = x[i, j] (u is [2H, 2W]; every other element 0).  The 2D BEV neck's four
// up-sampling layers (snvc/models/submodule.py:291-314) run that way on the depth-1 conv kernel; this is the
// stuffing pass: planes [R][H][W] -> [R][2H][2W], fully written.
__global__ void __launch_bounds__(256)
zero_stuff2x_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ow = (int)(i % (2 * W));
    const int oh = (int)((i / (2 * W)) % (2 * H));
    const int64_t r = i / ((int64_t)4 * H * W);
    y[i] = ((ow | oh) & 1) ? 0.0f : x[(r * H + (oh >> 1)) * W + (ow >> 1)];
}

}  // namespace
}  // namespace snvc

namespace snvc {
void launch_norm_finalize(const double *partial, const float *gamma, const float *beta, float *scale, float *shift, float *mean,
                          float *var, int64_t N, int64_t C, int64_t S, int splits, float eps, hipStream_t st) {
    norm_finalize_kernel<<<dim3((unsigned)ceil_div<int64_t>(C, 128)), 128, 0, st>>>(partial, gamma, beta, scale, shift, mean, var, N, C, S, C, 0,
                                                                                   splits, eps);
}
}  // namespace snvc

extern "C" {

int snvc_zero_stuff2x(const float *x, float *y, int64_t R, int64_t H, int64_t W, void *stream) {
    using namespace snvc;
    if (R < 0 || H < 0 || W < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_zero_stuff2x: negative size");
    const int64_t total = R * 4 * H * W;
    if (total == 0) return SNVC_OK;
    if (!x || !y) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_zero_stuff2x: null pointer");
    if (ceil_div<int64_t>(total, 256) >= ((int64_t)1 << 31) || 2 * W >= ((int64_t)1 << 30) || 2 * H >= ((int64_t)1 << 30))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_zero_stuff2x: tensor too large");
    zero_stuff2x_kernel<<<(unsigned)ceil_div<int64_t>(total, 256), 256, 0, as_stream(stream)>>>(x, y, (int)H, (int)W, total);
    return check_launch("snvc_zero_stuff2x");
}

int snvc_mul_broadcast(const float *feat, const float *occ, float *out, int64_t N, int64_t C,
                       int64_t S, int64_t out_batch_stride, void *stream) {
    using namespace snvc;
    if (N < 0 || C < 0 || S < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_mul_broadcast: negative size");
    if (N == 0 || C == 0 || S == 0) return SNVC_OK;
    if (!feat || !occ || !out) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_mul_broadcast: null pointer");
    if (C > 65535 || N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_mul_broadcast: C or N > 65535");
    if (out_batch_stride == 0) out_batch_stride = C * S;
    dim3 grid(stream_blocks(S / 4 + 1, N * C), (unsigned)C, (unsigned)N);
    mul_broadcast_kernel<<<grid, 256, 0, as_stream(stream)>>>(feat, occ, out, C, S, out_batch_stride);
    return check_launch("snvc_mul_broadcast");
}

int snvc_avgpool_depth4(const float *x, float *y, int64_t N, int64_t C, int64_t D, int64_t HW,
                        void *stream) {
    using namespace snvc;
    if (N < 0 || C < 0 || D < 0 || HW < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_avgpool_depth4: negative size");
    const int64_t Dq = D / 4;  // floor, like AvgPool3d without ceil_mode
    if (N * C == 0 || Dq == 0 || HW == 0) return SNVC_OK;
    if (!x || !y) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_avgpool_depth4: null pointer");
    if (Dq > 65535 || N * C > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_avgpool_depth4: grid too large");
    dim3 grid(stream_blocks(HW / 4 + 1, N * C * Dq), (unsigned)Dq, (unsigned)(N * C));
    avgpool_depth4_kernel<<<grid, 256, 0, as_stream(stream)>>>(x, y, D, HW);
    return check_launch("snvc_avgpool_depth4");
}

int snvc_avgpool_depth4_backward(const float *grad_y, float *grad_x, int64_t N, int64_t C, int64_t D, int64_t HW, void *stream) {
    using namespace snvc;
    if (N < 0 || C < 0 || D < 0 || HW < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_avgpool_depth4_backward: negative size");
    if (N * C == 0 || D == 0 || HW == 0) return SNVC_OK;
    if (!grad_x || (D / 4 > 0 && !grad_y)) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_avgpool_depth4_backward: null pointer");
    if (D > 65535 || N * C > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_avgpool_depth4_backward: grid too large");
    dim3 grid(stream_blocks(HW, N * C * D), (unsigned)D, (unsigned)(N * C));
    avgpool_depth4_bwd_kernel<<<grid, 256, 0, as_stream(stream)>>>(grad_y, grad_x, D, HW);
    return check_launch("snvc_avgpool_depth4_backward");
}

int snvc_disparity_regression(const float *x, const float *depth, float *out, int64_t N, int64_t D,
                              int64_t HW, void *stream) {
    using namespace snvc;
    if (N < 0 || D < 0 || HW < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_disparity_regression: negative size");
    if (N == 0 || HW == 0) return SNVC_OK;
    if (!out || (D > 0 && (!x || !depth))) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_disparity_regression: null pointer");
    if (N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_disparity_regression: N > 65535");
    dim3 grid((unsigned)ceil_div<int64_t>(HW, 256), (unsigned)N);
    disparity_regression_kernel<<<grid, 256, 0, as_stream(stream)>>>(x, depth, out, D, HW);
    return check_launch("snvc_disparity_regression");
}

int snvc_argmax_rows(const float *x, int64_t *idx, float *val, int64_t R, int64_t L, void *stream) {
    using namespace snvc;
    if (R < 0 || L < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_argmax_rows: negative size");
    if (R == 0) return SNVC_OK;
    if (L == 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_argmax_rows: attempt to get argmax of an empty sequence");
    if (!x || !idx) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_argmax_rows: null pointer");
    argmax_rows_kernel<<<dim3((unsigned)R), 256, 0, as_stream(stream)>>>(x, idx, val, L);
    return check_launch("snvc_argmax_rows");
}

int64_t snvc_norm_workspace_bytes(int64_t N, int64_t C, int64_t groups) {
    const int64_t rows = N * (groups > C ? groups : C);
    return rows * snvc::kNormSplits * 2 * (int64_t)sizeof(double);
}

int snvc_norm_stats(const float *x, const float *gamma, const float *beta, float *scale, float *shift,
                    float *mean_out, float *var_out, void *workspace, int64_t N, int64_t C, int64_t S,
                    int64_t x_batch_stride, int64_t groups, int per_sample, float eps, void *stream) {
    using namespace snvc;
    if (N <= 0 || C <= 0 || S <= 0 || groups <= 0 || C % groups)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_norm_stats: sizes must be positive and C divisible by groups");
    if (!per_sample && groups != C)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_norm_stats: batch statistics need groups == C");
    if (!x || !scale || !shift || !workspace) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_norm_stats: null pointer");
    if (x_batch_stride == 0) x_batch_stride = C * S;
    const int64_t rows_per_n = groups, len = (C / groups) * S, rows = N * rows_per_n;
    if (rows > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_norm_stats: N*groups > 65535");
    dim3 grid(kNormSplits, (unsigned)rows);
    if (len % 4 == 0 && x_batch_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        norm_partial_kernel<4><<<grid, 256, 0, as_stream(stream)>>>(x, (double *)workspace, rows_per_n, len, x_batch_stride, kNormSplits);
    else
        norm_partial_kernel<1><<<grid, 256, 0, as_stream(stream)>>>(x, (double *)workspace, rows_per_n, len, x_batch_stride, kNormSplits);
    int rc = check_launch("snvc_norm_stats(partial)");
    if (rc) return rc;
    const int64_t t = (per_sample ? N : 1) * C;
    norm_finalize_kernel<<<dim3((unsigned)ceil_div<int64_t>(t, 128)), 128, 0, as_stream(stream)>>>(
        (const double *)workspace, gamma, beta, scale, shift, mean_out, var_out, N, C, S, groups, per_sample,
        kNormSplits, eps);
    return check_launch("snvc_norm_stats(finalize)");
}

int64_t snvc_act_backward_workspace_bytes(int64_t N, int64_t C) {
    return N * C * snvc::kNormSplits * 2 * (int64_t)sizeof(double);
}

int snvc_act_backward_reduce(const float *raw, const float *gy, const float *residual, const float *scale,
                             const float *shift, double *sums, void *workspace, int64_t N, int64_t C, int64_t S,
                             int64_t raw_batch_stride, int64_t gy_batch_stride, int64_t res_batch_stride,
                             int per_sample, int flags, void *stream) {
    return snvc_act_backward_reduce_amax(raw, gy, residual, scale, shift, sums, workspace, N, C, S, raw_batch_stride, gy_batch_stride,
                                         res_batch_stride, per_sample, flags, nullptr, stream);
}

int snvc_act_backward_reduce_amax(const float *raw, const float *gy, const float *residual, const float *scale,
                                  const float *shift, double *sums, void *workspace, int64_t N, int64_t C, int64_t S,
                                  int64_t raw_batch_stride, int64_t gy_batch_stride, int64_t res_batch_stride,
                                  int per_sample, int flags, uint32_t *amax_gy, void *stream) {
    using namespace snvc;
    if (N <= 0 || C <= 0 || S <= 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_act_backward_reduce: sizes must be positive");
    if (!raw || !gy || !sums || !workspace) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_act_backward_reduce: null pointer");
    if ((flags & SNVC_EPI_ADD_PRE) && !residual) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_act_backward_reduce: ADD_PRE needs the residual");
    if (N * C > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_act_backward_reduce: N*C > 65535");
    // the residual only enters through the activation's derivative: without one (conv6: bn(conv) + x) it is not read (r6: 736 MB per pass at cfg4)
    if (!(flags & SNVC_EPI_ADD_PRE) || !(flags & (SNVC_EPI_RELU | SNVC_EPI_SIGMOID))) residual = nullptr;
    if (raw_batch_stride == 0) raw_batch_stride = C * S;
    if (gy_batch_stride == 0) gy_batch_stride = C * S;
    if (res_batch_stride == 0) res_batch_stride = C * S;
    dim3 grid(kNormSplits, (unsigned)(N * C));
    const bool v4 = S % 4 == 0 && raw_batch_stride % 4 == 0 && gy_batch_stride % 4 == 0 && res_batch_stride % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(raw) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
    if (v4)
        act_bwd_partial_kernel<4><<<grid, 256, 0, as_stream(stream)>>>(raw, gy, residual, scale, shift, (double *)workspace, C, S,
                                                                       raw_batch_stride, gy_batch_stride, res_batch_stride,
                                                                       per_sample, flags, kNormSplits, amax_gy);
    else
        act_bwd_partial_kernel<1><<<grid, 256, 0, as_stream(stream)>>>(raw, gy, residual, scale, shift, (double *)workspace, C, S,
                                                                       raw_batch_stride, gy_batch_stride, res_batch_stride,
                                                                       per_sample, flags, kNormSplits, amax_gy);
    int rc = check_launch("snvc_act_backward_reduce(partial)");
    if (rc) return rc;
    act_bwd_fold_kernel<<<dim3((unsigned)ceil_div<int64_t>(N * C, 128)), 128, 0, as_stream(stream)>>>(
        (const double *)workspace, sums, N * C, kNormSplits);
    return check_launch("snvc_act_backward_reduce(fold)");
}

int snvc_act_backward_apply(const float *raw, const float *gy, const float *residual, const float *scale,
                            const float *shift, const float *coef_g, const float *coef_raw, const float *coef_const,
                            float *draw, float *g_out, int64_t N, int64_t C, int64_t S, int64_t raw_batch_stride,
                            int64_t gy_batch_stride, int64_t res_batch_stride, int per_sample, int flags, void *stream) {
    return snvc_act_backward_apply_amax(raw, gy, residual, scale, shift, coef_g, coef_raw, coef_const, draw, g_out, N, C, S,
                                        raw_batch_stride, gy_batch_stride, res_batch_stride, per_sample, flags, nullptr, stream);
}

int snvc_act_backward_apply_amax(const float *raw, const float *gy, const float *residual, const float *scale,
                                 const float *shift, const float *coef_g, const float *coef_raw, const float *coef_const,
                                 float *draw, float *g_out, int64_t N, int64_t C, int64_t S, int64_t raw_batch_stride,
                                 int64_t gy_batch_stride, int64_t res_batch_stride, int per_sample, int flags, uint32_t *amax,
                                 void *stream) {
    using namespace snvc;
    if (N < 0 || C < 0 || S < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_act_backward_apply: negative size");
    if (N == 0 || C == 0 || S == 0) return SNVC_OK;
    if (!raw || !gy || !coef_g || !draw) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_act_backward_apply: null pointer");
    if ((flags & SNVC_EPI_ADD_PRE) && !residual) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_act_backward_apply: ADD_PRE needs the residual");
    if (C > 65535 || N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_act_backward_apply: C or N > 65535");
    // the residual only enters through the activation's derivative: without one (conv6: bn(conv) + x) it is not read (r6: 736 MB per pass at cfg4)
    if (!(flags & SNVC_EPI_ADD_PRE) || !(flags & (SNVC_EPI_RELU | SNVC_EPI_SIGMOID))) residual = nullptr;
    if (raw_batch_stride == 0) raw_batch_stride = C * S;
    if (gy_batch_stride == 0) gy_batch_stride = C * S;
    if (res_batch_stride == 0) res_batch_stride = C * S;
    dim3 grid(stream_blocks(S / 4 + 1, N * C), (unsigned)C, (unsigned)N);
    act_bwd_apply_kernel<<<grid, 256, 0, as_stream(stream)>>>(raw, gy, residual, scale, shift, coef_g, coef_raw, coef_const,
                                                              draw, g_out, C, S, raw_batch_stride, gy_batch_stride,
                                                              res_batch_stride, per_sample, flags, amax);
    return check_launch("snvc_act_backward_apply");
}

int snvc_bn_backward_coefs(const double *sums, const float *mean, const float *var, const float *gamma, float *coef_g,
                           float *coef_raw, float *coef_const, float *dgamma, float *dbeta, int64_t N, int64_t C, double count,
                           double eps, void *stream) {
    using namespace snvc;
    if (N <= 0 || C <= 0 || !(count > 0.0)) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_bn_backward_coefs: sizes must be positive");
    if (!sums || !mean || !var || !coef_g || !coef_raw || !coef_const)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_bn_backward_coefs: null pointer");
    bn_bwd_coefs_kernel<<<dim3((unsigned)ceil_div<int64_t>(C, 64)), 64, 0, as_stream(stream)>>>(
        sums, mean, var, gamma, coef_g, coef_raw, coef_const, dgamma, dbeta, N, C, count, eps);
    return check_launch("snvc_bn_backward_coefs");
}

int snvc_affine_act(const float *x, const float *scale, const float *shift, const float *residual,
                    float *y, int64_t N, int64_t C, int64_t S, int64_t x_batch_stride,
                    int64_t y_batch_stride, int64_t res_batch_stride, int per_sample, int flags,
                    void *stream) {
    return snvc_affine_act_amax(x, scale, shift, residual, y, N, C, S, x_batch_stride, y_batch_stride, res_batch_stride, per_sample,
                                flags, nullptr, stream);
}

int snvc_affine_act_amax(const float *x, const float *scale, const float *shift, const float *residual,
                         float *y, int64_t N, int64_t C, int64_t S, int64_t x_batch_stride,
                         int64_t y_batch_stride, int64_t res_batch_stride, int per_sample, int flags,
                         uint32_t *amax, void *stream) {
    using namespace snvc;
    if (N < 0 || C < 0 || S < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_affine_act: negative size");
    if (N == 0 || C == 0 || S == 0) return SNVC_OK;
    if (!x || !y) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_affine_act: null pointer");
    if ((flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST)) && !residual)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_affine_act: residual flag without residual pointer");
    if (C > 65535 || N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_affine_act: C or N > 65535");
    if (!(flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST))) residual = nullptr;
    if (x_batch_stride == 0) x_batch_stride = C * S;
    if (y_batch_stride == 0) y_batch_stride = C * S;
    if (res_batch_stride == 0) res_batch_stride = C * S;
    dim3 grid(stream_blocks(S / 4 + 1, N * C), (unsigned)C, (unsigned)N);
    affine_act_kernel<<<grid, 256, 0, as_stream(stream)>>>(x, scale, shift, residual, y, C, S, x_batch_stride,
                                                           y_batch_stride, res_batch_stride, per_sample, flags, amax);
    return check_launch("snvc_affine_act");
}

// 2048 workgroups in all (measured at cfg4's layer sizes: 1024-8192 and contiguous chunks per workgroup instead of a grid stride make
// no difference, profiles/r6/kernel_experiments_r6.txt item 19)
static unsigned twin_blocks(int64_t s4, int64_t outer) {
    int64_t b = snvc::ceil_div<int64_t>(s4, 256);
    const int64_t cap = snvc::ceil_div<int64_t>(2048, outer > 0 ? outer : 1);
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

static int twin_check(const char *who, int64_t N, int64_t C, int64_t S, const void *t_hi, const void *t_lo, const float *mul,
                      int64_t twin_batch_stride, std::initializer_list<const void *> f32, std::initializer_list<int64_t> strides) {
    using namespace snvc;
    if (N <= 0 || C <= 0 || S <= 0) { set_error("%s: sizes must be positive", who); return SNVC_ERR_INVALID_ARGUMENT; }
    if (C % 8 || S % 4) { set_error("%s: C must be a multiple of 8 and the voxel count of 4", who); return SNVC_ERR_UNSUPPORTED; }
    if (!t_hi || !t_lo || !mul) { set_error("%s: null twin pointer", who); return SNVC_ERR_INVALID_ARGUMENT; }
    if (C / 8 > 65535 || N > 65535) { set_error("%s: C/8 or N > 65535", who); return SNVC_ERR_UNSUPPORTED; }
    uintptr_t bits = reinterpret_cast<uintptr_t>(t_hi) | reinterpret_cast<uintptr_t>(t_lo);
    for (const void *q : f32) bits |= reinterpret_cast<uintptr_t>(q);
    if (bits & 15) { set_error("%s: tensors must be 16-byte aligned", who); return SNVC_ERR_INVALID_ARGUMENT; }
    for (int64_t st : strides)
        if (st % 4) { set_error("%s: batch strides must be multiples of 4 elements", who); return SNVC_ERR_INVALID_ARGUMENT; }
    if (twin_batch_stride % 8) { set_error("%s: the twin's batch stride must be a multiple of 8", who); return SNVC_ERR_INVALID_ARGUMENT; }
    return SNVC_OK;
}

int snvc_affine_act_twin(const float *x, const float *scale, const float *shift, const float *residual, float *y, void *twin_hi,
                         void *twin_lo, const float *twin_mul, int64_t N, int64_t C, int64_t S, int64_t x_batch_stride,
                         int64_t y_batch_stride, int64_t res_batch_stride, int64_t twin_batch_stride, int per_sample, int flags,
                         uint32_t *amax, void *stream) {
    using namespace snvc;
    if (!x || !y) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_affine_act_twin: null pointer");
    if ((flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST)) && !residual)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_affine_act_twin: residual flag without residual pointer");
    if (!(flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST))) residual = nullptr;
    if (x_batch_stride == 0) x_batch_stride = C * S;
    if (y_batch_stride == 0) y_batch_stride = C * S;
    if (res_batch_stride == 0) res_batch_stride = C * S;
    if (twin_batch_stride == 0) twin_batch_stride = 2 * C * S;
    int rc = twin_check("snvc_affine_act_twin", N, C, S, twin_hi, twin_lo, twin_mul, twin_batch_stride, {x, y, residual},
                        {x_batch_stride, y_batch_stride, res_batch_stride});
    if (rc) return rc;
    dim3 grid(twin_blocks(S / 4, N * C / 8), (unsigned)(C / 8), (unsigned)N);
    affine_act_twin_kernel<<<grid, 256, 0, as_stream(stream)>>>(x, scale, shift, residual, y, reinterpret_cast<_Float16 *>(twin_hi),
                                                                reinterpret_cast<_Float16 *>(twin_lo), twin_mul, C, S, x_batch_stride,
                                                                y_batch_stride, res_batch_stride, twin_batch_stride, per_sample, flags, amax);
    return check_launch("snvc_affine_act_twin");
}

int snvc_act_backward_apply_twin(const float *raw, const float *gy, const float *residual, const float *scale, const float *shift,
                                 const float *coef_g, const float *coef_raw, const float *coef_const, float *draw, float *g_out,
                                 void *twin_hi, void *twin_lo, const float *twin_mul, int64_t N, int64_t C, int64_t S,
                                 int64_t raw_batch_stride, int64_t gy_batch_stride, int64_t res_batch_stride, int64_t twin_batch_stride,
                                 int per_sample, int flags, uint32_t *amax, void *stream) {
    using namespace snvc;
    if (!raw || !gy || !coef_g || !draw) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_act_backward_apply_twin: null pointer");
    if ((flags & SNVC_EPI_ADD_PRE) && !residual) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_act_backward_apply_twin: ADD_PRE needs the residual");
    // the residual only enters through the activation's derivative: without one (conv6: bn(conv) + x) it is not read (r6: 736 MB per pass at cfg4)
    if (!(flags & SNVC_EPI_ADD_PRE) || !(flags & (SNVC_EPI_RELU | SNVC_EPI_SIGMOID))) residual = nullptr;
    if (raw_batch_stride == 0) raw_batch_stride = C * S;
    if (gy_batch_stride == 0) gy_batch_stride = C * S;
    if (res_batch_stride == 0) res_batch_stride = C * S;
    if (twin_batch_stride == 0) twin_batch_stride = 2 * C * S;
    int rc = twin_check("snvc_act_backward_apply_twin", N, C, S, twin_hi, twin_lo, twin_mul, twin_batch_stride, {raw, gy, residual, draw, g_out},
                        {raw_batch_stride, gy_batch_stride, res_batch_stride});
    if (rc) return rc;
    dim3 grid(twin_blocks(S / 4, N * C / 8), (unsigned)(C / 8), (unsigned)N);
    act_bwd_apply_twin_kernel<<<grid, 256, 0, as_stream(stream)>>>(raw, gy, residual, scale, shift, coef_g, coef_raw, coef_const, draw, g_out,
                                                                   reinterpret_cast<_Float16 *>(twin_hi), reinterpret_cast<_Float16 *>(twin_lo),
                                                                   twin_mul, C, S, raw_batch_stride, gy_batch_stride, res_batch_stride,
                                                                   twin_batch_stride, per_sample, flags, amax);
    return check_launch("snvc_act_backward_apply_twin");
}

int snvc_bn_track(float *running_mean, float *running_var, int64_t *num_batches_tracked, const float *mean, const float *var, int64_t C,
                  float momentum, float unbias, void *stream) {
    using namespace snvc;
    if (C <= 0 || !running_mean || !running_var || !mean || !var) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_bn_track: null pointer or empty");
    bn_track_kernel<<<dim3((unsigned)ceil_div<int64_t>(C, 64)), 64, 0, as_stream(stream)>>>(running_mean, running_var, num_batches_tracked, mean, var,
                                                                                      (int)C, momentum, unbias);
    return check_launch("snvc_bn_track");
}

int snvc_split_scale_bound(const float *a, const uint32_t *amax_p, const float *b, const float *l1, const uint32_t *amax_x, const float *c,
                           const uint32_t *amax_r, int64_t rows, int64_t C, float *mul_out, void *stream) {
    using namespace snvc;
    if (rows <= 0 || C <= 0 || rows % C || rows > (1 << 24)) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_split_scale_bound: rows must be a positive multiple of C");
    if (!mul_out) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_split_scale_bound: null pointer");
    split_scale_bound_kernel<<<1, 256, 0, as_stream(stream)>>>(a, amax_p, b, l1, amax_x, c, amax_r, (int)rows, (int)C, mul_out);
    return check_launch("snvc_split_scale_bound");
}

}  // extern "C"
