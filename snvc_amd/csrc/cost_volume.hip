// Plane-sweep concat cost volume for gfx950.
//
// Semantics follow the reference kernels
//   forward  snvc/extension/build_cost_volume/src/BuildCostVolume_cuda.cu:15-98
//   backward snvc/extension/build_cost_volume/src/BuildCostVolume_cuda.cu:101-205
// but the decomposition is new:
//   * forward is a pure HBM-write stream: one workgroup walks whole (n, c, d) output planes
//     (H*W contiguous floats), plane decode on the scalar unit, 16-byte stores, and the
//     small left/right feature planes are re-read from L2;
//   * backward is a deterministic GATHER (one thread per input pixel, d ascending) instead
//     of the reference's float atomics with D-way contention, and writes every output
//     element, so no zero-fill pass is needed.
// Every product / sum is rounded on its own (fp contract off), in the order the reference
// writes them, which makes the results bit-identical to oracle/cost_volume_ref.c.
#include "common.hpp"

namespace snvc {
namespace {

#pragma clang fp contract(off)

// bilinear_interpolate (BuildCostVolume_cuda.cu:15-61) specialised to what the forward can
// reach: y = ih integral, x gated to [0, img_w - 1].  The y_high row is still read and
// multiplied by its zero weight, like the reference, so non-finite inputs propagate alike.
template <typename T>
__device__ __forceinline__ T sample_right(const T *__restrict__ plane, int img_h, int img_w,
                                          int ih, T x) {
    int y_lo = ih, y_hi;
    if (y_lo >= img_h - 1) { y_hi = y_lo = img_h - 1; } else { y_hi = y_lo + 1; }
    int x_lo = (int)x, x_hi;
    if (x_lo >= img_w - 1) { x_hi = x_lo = img_w - 1; x = (T)x_lo; } else { x_hi = x_lo + 1; }
    const T ly = (T)0, lx = x - (T)x_lo;
    const T hy = (T)1 - ly, hx = (T)1 - lx;
    const T v1 = plane[(int64_t)y_lo * img_w + x_lo], v2 = plane[(int64_t)y_lo * img_w + x_hi];
    const T v3 = plane[(int64_t)y_hi * img_w + x_lo], v4 = plane[(int64_t)y_hi * img_w + x_hi];
    const T w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

template <typename T>
__device__ __forceinline__ T right_value(const T *__restrict__ rplane, int img_h, int img_w,
                                         int ih, int iw, T neg_shift) {
    const T x = (T)iw + neg_shift;  // BuildCostVolume_cuda.cu:88,91
    if (x >= (T)0 && x <= (T)(img_w - 1)) return sample_right(rplane, img_h, img_w, ih, x);
    return (T)0;
}

// Generic forward: any dtype, any downsample.  One thread per output element of a plane.
template <typename T>
__global__ void __launch_bounds__(256)
cost_volume_fwd_generic(const T *__restrict__ left, const T *__restrict__ right,
                        const T *__restrict__ shift, T *__restrict__ out, int C, int D, int H,
                        int W, int ds, int64_t planes) {
    const int img_h = H * ds, img_w = W * ds;
    const int hw = H * W;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e < hw;
    const int h = live ? e / W : 0, w = live ? e - h * W : 0;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {  // p = (n*2C + c2)*D + d, uniform
        const int d = (int)(p % D);
        const int64_t nc2 = p / D;
        const int c2 = (int)(nc2 % (2 * C));
        const int64_t n = nc2 / (2 * C);
        if (!live) continue;
        T v;
        if (c2 < C) {
            v = left[((n * C + c2) * img_h + (int64_t)h * ds) * img_w + (int64_t)w * ds];
        } else {
            const T *rplane = right + (n * C + (c2 - C)) * (int64_t)img_h * img_w;
            v = right_value(rplane, img_h, img_w, h * ds, w * ds, -shift[n * D + d]);
        }
        out[p * hw + e] = v;
    }
}

// Fast forward: fp32, downsample 1, W % 4 == 0 -> one 16-byte store per lane.
__global__ void __launch_bounds__(256)
cost_volume_fwd_f32x4(const float *__restrict__ left, const float *__restrict__ right,
                      const float *__restrict__ shift, float *__restrict__ out, int C, int D,
                      int H, int W, int64_t planes) {
    const int hw4 = (H * W) >> 2;
    const int e4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e4 < hw4;
    const int e = e4 << 2;
    const int h = live ? e / W : 0, w = live ? e - h * W : 0;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {
        const int d = (int)(p % D);
        const int64_t nc2 = p / D;
        const int c2 = (int)(nc2 % (2 * C));
        const int64_t n = nc2 / (2 * C);
        if (!live) continue;
        float4 v;
        if (c2 < C) {
            v = *reinterpret_cast<const float4 *>(left + (n * C + c2) * (int64_t)H * W + e);
        } else {
            const float *rplane = right + (n * C + (c2 - C)) * (int64_t)H * W;
            const float ns = -shift[n * D + d];
            v.x = right_value(rplane, H, W, h, w + 0, ns);
            v.y = right_value(rplane, H, W, h, w + 1, ns);
            v.z = right_value(rplane, H, W, h, w + 2, ns);
            v.w = right_value(rplane, H, W, h, w + 3, ns);
        }
        *reinterpret_cast<float4 *>(out + p * (int64_t)H * W + e) = v;
    }
}

// fp64, downsample 1, W % 2 == 0 -> one 16-byte store per lane (r6: the generic kernel's 8-byte stores and per-element index
// arithmetic moved 2.2 TB/s on the 2.96 GB cfg2 volume).  Same per-element arithmetic as right_value<double>.
__global__ void __launch_bounds__(256)
cost_volume_fwd_f64x2(const double *__restrict__ left, const double *__restrict__ right,
                      const double *__restrict__ shift, double *__restrict__ out, int C, int D,
                      int H, int W, int64_t planes) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int hw2 = (H * W) >> 1;
    const int e2 = blockIdx.x * blockDim.x + threadIdx.x;
    if (e2 >= hw2) return;
    const int e = e2 << 1;
    const int h = e / W, w = e - h * W;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {
        const int d = (int)(p % D);
        const int64_t nc2 = p / D;
        const int c2 = (int)(nc2 % (2 * C));
        const int64_t n = nc2 / (2 * C);
        d2 v;
        if (c2 < C) {
            v = *reinterpret_cast<const d2 *>(left + (n * C + c2) * (int64_t)H * W + e);
        } else {
            const double *rplane = right + (n * C + (c2 - C)) * (int64_t)H * W;
            const double ns = -shift[n * D + d];
            v[0] = right_value(rplane, H, W, h, w + 0, ns);
            v[1] = right_value(rplane, H, W, h, w + 1, ns);
        }
        __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(out + p * (int64_t)H * W + e));
    }
}

// Branch-free twin of right_value() for an LDS slab of `rows` (1 or 2) rows of width img_w: the gate
// becomes a select, so the four elements of a float4 issue their LDS reads together.  For lanes
// that pass the gate the arithmetic is operation-for-operation the same as sample_right().
__device__ __forceinline__ float right_value_lds(const float *__restrict__ slab, int rows, int img_w, int iw,
                                                 float neg_shift) {
    const float x0 = (float)iw + neg_shift;
    const bool ok = x0 >= 0.0f && x0 <= (float)(img_w - 1);
    float x = ok ? x0 : 0.0f;
    int x_lo = (int)x, x_hi;
    if (x_lo >= img_w - 1) { x_hi = x_lo = img_w - 1; x = (float)x_lo; } else { x_hi = x_lo + 1; }
    const int y_hi = rows - 1;                       // row below (same row on the last image row)
    const float ly = 0.0f, lx = x - (float)x_lo;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float v1 = slab[x_lo], v2 = slab[x_hi];
    const float v3 = slab[y_hi * img_w + x_lo], v4 = slab[y_hi * img_w + x_hi];
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    const float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
    return ok ? val : 0.0f;
}

// The same value where the caller has established (for the whole wave) that x0 < img_w - 1, so the right-edge clamp of
// sample_right() cannot trigger: what is left of the reference's arithmetic, operation for operation, is
//   x_lo = floor(x) (x >= 0), lx = x - x_lo, hx = 1 - lx, w1 = 1*hx = hx, w2 = 1*lx = lx, w3 = 0*hx = +0, w4 = 0*lx = +0
// (hx in (0, 1] and lx in [0, 1) are finite and non-negative, so the two zero weights are +0 exactly), then the four
// products and three sums in the reference's order -- the zero-weight taps are still loaded and multiplied, so
// non-finite inputs and signed zeros come out alike.  16 VALU operations per element instead of 27: the forward was
// VALU-bound (4.5 TB/s of stores against 6.9 TB/s of a plain fill).
__device__ __forceinline__ float right_value_lds_inner(const float *__restrict__ slab, int rowoff, float x0) {
    const bool ok = x0 >= 0.0f;
    const float x = ok ? x0 : 0.0f;
    const float xf = __builtin_floorf(x);
    const float lx = x - xf, hx = 1.0f - lx;
    const float *p = slab + (int)xf;
    const float v1 = p[0], v2 = p[1], v3 = p[rowoff], v4 = p[rowoff + 1];
    const float val = hx * v1 + lx * v2 + 0.0f * v3 + 0.0f * v4;
    return ok ? val : 0.0f;
}

// Fast forward v2: fp32, downsample 1, W % 4 == 0.  A workgroup owns RB consecutive rows of one
// (n, c) feature plane: the right rows (+ the row below, whose zero-weight taps the reference still
// loads) are staged ONCE in LDS, the left float4 of every thread stays in a register, and the
// workgroup then walks a range of disparity planes writing, per plane, RB*W contiguous floats of
// the left half and of the right half.  Gathers hit LDS instead of L1/L2, the shift is a scalar
// load, and all vector-memory traffic is the two 16-byte stores per thread per plane.
template <bool WITH_LEFT>
__global__ void __launch_bounds__(512)
cost_volume_fwd_rows(const float *__restrict__ left, const float *__restrict__ right,
                     const float *__restrict__ shift, float *__restrict__ out, int C, int D, int H, int W,
                     int RB, int hblocks, int dchunk) {
    extern __shared__ __attribute__((aligned(16))) float rrows[];   // [(RB+1)][W]
    const int W4 = W >> 2;
    const int hb = blockIdx.x % hblocks;
    const int64_t nc = blockIdx.x / hblocks;            // n*C + c
    const int c = (int)(nc % C);
    const int64_t n = nc / C;
    const int h0 = hb * RB;
    const float *rplane = right + nc * (int64_t)H * W;
    // stage rows h0 .. h0+RB (clamped to H-1) of the right plane
    for (int i = threadIdx.x; i < (RB + 1) * W4; i += blockDim.x) {
        const int r = i / W4, q = i - r * W4;
        int gh = h0 + r;
        gh = gh < H ? gh : H - 1;
        reinterpret_cast<float4 *>(rrows)[i] = reinterpret_cast<const float4 *>(rplane + (int64_t)gh * W)[q];
    }
    const int row = threadIdx.x / W4, q = threadIdx.x - row * W4;
    const int h = h0 + row;
    const bool live = row < RB && h < H;
    float4 lv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (WITH_LEFT && live) lv = reinterpret_cast<const float4 *>(left + nc * (int64_t)H * W + (int64_t)h * W)[q];
    __syncthreads();
    int wm = live ? (q << 2) + 3 : 0;                   // largest column of this wave (wave-uniform: a scalar test per plane)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) wm = max(wm, __shfl_xor(wm, off));
    const float wave_w3 = (float)__builtin_amdgcn_readfirstlane(wm);
    if (!live) return;
    // image seen by sample_right: rows [row, row+1] of the LDS slab act as rows [h, min(h+1,H-1)]
    const float *slab = rrows + row * W;
    const int slab_h = (h >= H - 1) ? 1 : 2;            // y_hi == y_lo on the last image row
    const int w = q << 2;
    const int rowoff = (slab_h - 1) * W;
    const float wlast = (float)(W - 1);
    const int d0 = blockIdx.y * dchunk;
    const int d1 = d0 + dchunk < D ? d0 + dchunk : D;
    const int64_t plane = (int64_t)H * W;
    // WITH_LEFT: out is the full [N,2C,D,H,W] volume; otherwise only its right half [N,C,D,H,W]
    float *oL = out + ((n * (WITH_LEFT ? 2 : 1) * C + c) * (int64_t)D + d0) * plane + (int64_t)h * W + w;
    float *oR = WITH_LEFT ? oL + (int64_t)C * D * plane : oL;
#pragma unroll 2
    for (int d = d0; d < d1; ++d) {
        const float ns = -shift[n * D + d];
        const float x0 = (float)(w + 0) + ns, x1 = (float)(w + 1) + ns, x2 = (float)(w + 2) + ns, x3 = (float)(w + 3) + ns;
        float4 v;
        if (!(wave_w3 + ns >= wlast)) {     // no lane of the wave at or beyond the last column (x ascends with w: float add is monotonic)
            v.x = right_value_lds_inner(slab, rowoff, x0);
            v.y = right_value_lds_inner(slab, rowoff, x1);
            v.z = right_value_lds_inner(slab, rowoff, x2);
            v.w = right_value_lds_inner(slab, rowoff, x3);
        } else {
            v.x = right_value_lds(slab, slab_h, W, w + 0, ns);
            v.y = right_value_lds(slab, slab_h, W, w + 1, ns);
            v.z = right_value_lds(slab, slab_h, W, w + 2, ns);
            v.w = right_value_lds(slab, slab_h, W, w + 3, ns);
        }
        if (WITH_LEFT) *reinterpret_cast<float4 *>(oL) = lv;
        *reinterpret_cast<float4 *>(oR) = v;
        oL += plane;
        oR += plane;
    }
}

// Backward gather.  One thread per element (n, c, iy, ix) of grad_left / grad_right
// [N,C,H*ds,W*ds].  For grad_right the thread visits, for every d, the few output columns w
// whose sample position x = w*ds - shift has x_low == ix or x_high == ix, recomputing the
// reference's gate / clamp / weight arithmetic for each candidate (BuildCostVolume_cuda.cu:
// 101-150,178-202), and adds g*w1 / g*w2 in (d, w, tap) order.  Taps 3,4 carry weight
// ly*hx = 0 < 1e-10 and are never added (:199-202).
template <typename T>
__global__ void __launch_bounds__(256)
cost_volume_bwd_gather(const T *__restrict__ grad, const T *__restrict__ shift,
                       T *__restrict__ grad_left, T *__restrict__ grad_right, int C, int D, int H,
                       int W, int ds, int64_t total) {
    const int img_h = H * ds, img_w = W * ds;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ix = (int)(idx % img_w);
    const int iy = (int)((idx / img_w) % img_h);
    const int64_t nc = idx / ((int64_t)img_w * img_h);
    const int c = (int)(nc % C);
    const int64_t n = nc / C;
    T acc_l = (T)0, acc_r = (T)0;
    if (iy % ds == 0) {
        const int h = iy / ds;
        const int64_t hw = (int64_t)H * W;
        const T *gl = grad + ((n * 2 * C + c) * (int64_t)D) * hw + (int64_t)h * W;
        const T *gr = gl + (int64_t)C * D * hw;
        const bool on_lattice = (ix % ds == 0);
        const int wl = ix / ds;
        for (int d = 0; d < D; ++d) {
            const T s = shift[n * D + d];
            const T neg_shift = -s;
            if (on_lattice) acc_l = acc_l + gl[(int64_t)d * hw + wl];
            // candidate columns: w*ds - s within (ix - 1, ix + 1), widened by one column
            // on each side to absorb rounding of the float expression
            const T centre = ((T)ix + s) / (T)ds;
            int w0 = (int)floor((double)centre - 1.0 / ds) - 1;
            int w1 = (int)floor((double)centre + 1.0 / ds) + 1;
            if (w0 < 0) w0 = 0;
            if (w1 > W - 1) w1 = W - 1;
            for (int w = w0; w <= w1; ++w) {
                const int iw = w * ds;
                T x = (T)iw + neg_shift;
                if (!(x >= (T)0 && x <= (T)(img_w - 1))) continue;
                int x_lo = (int)x, x_hi;
                if (x_lo >= img_w - 1) { x_hi = x_lo = img_w - 1; x = (T)x_lo; } else { x_hi = x_lo + 1; }
                if (x_lo != ix && x_hi != ix) continue;
                const T lx = x - (T)x_lo;
                const T hy = (T)1, hx = (T)1 - lx;
                const T wt1 = hy * hx, wt2 = hy * lx;
                const T g = gr[(int64_t)d * hw + w];
                const T g1 = g * wt1, g2 = g * wt2;
                if (x_lo == ix && (double)wt1 >= 1e-10) acc_r = acc_r + g1;
                if (x_hi == ix && (double)wt2 >= 1e-10) acc_r = acc_r + g2;
            }
        }
    }
    grad_left[idx] = acc_l;
    grad_right[idx] = acc_r;
}


// fp32, downsample 1: the same sum in the same order (d ascending, column ascending, tap 1 before tap 2), arranged
// for memory-level parallelism.  With downsample 1 and s = shift[n,d] >= 0, pixel ix can only receive from the two
// columns w = ix + floor(s) and w + 1: the sample position x = w - s of any other column lies outside [ix - 1, ix + 1],
// and rounding of the float expression (float)w + (-s) is monotone around the representable integers ix - 1 and
// ix + 1, where the touching tap has weight exactly 0 (< 1e-10: never added, BuildCostVolume_cuda.cu:195-202).  The
// two candidates are adjacent floats of the gradient row (one 8-byte load); each goes through the reference's own
// gate / clamp / weight arithmetic.  Eight disparity planes are in flight per thread before the first add, where the
// generic kernel issues ~6 dependent scalar loads per plane behind double-precision index arithmetic: 0.82 -> 0.36 ms on cfg2 (1.48 GB read, 52 % of 8 TB/s).
// `wt >= 1e-10` (a double comparison in the reference) == `wt >= 0x1.b7cdfep-34f`: the float nearest to 1e-10 lies
// above it and its predecessor below.
struct __attribute__((packed, aligned(4))) Pair2 { float v[2]; };

template <bool WITH_LEFT>   // false: `grad` is the right (warped) half only, [N,C,D,H,W]; grad_left is not written
__global__ void __launch_bounds__(256)
cost_volume_bwd_rows_f32(const float *__restrict__ grad, const float *__restrict__ shift, float *__restrict__ grad_left,
                         float *__restrict__ grad_right, int C, int D, int H, int W, int64_t total) {
    constexpr float kMinWeight = 0x1.b7cdfep-34f;
    constexpr int PL = 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ix = (int)(idx % W);
    const int h = (int)((idx / W) % H);
    const int64_t nc = idx / ((int64_t)W * H);
    const int c = (int)(nc % C);
    const int64_t n = nc / C;
    const int64_t hw = (int64_t)H * W;
    const float *gl = grad + ((n * (WITH_LEFT ? 2 : 1) * C + c) * (int64_t)D) * hw + (int64_t)h * W;
    const float *gr = WITH_LEFT ? gl + (int64_t)C * D * hw : gl;
    const float *sh = shift + n * D;
    float acc_l = 0.0f, acc_r = 0.0f;
    auto one_plane = [&](float s, float glv, const Pair2 &q, int wq) {
        if (WITH_LEFT) acc_l = acc_l + glv;
        const float neg_shift = -s;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int w = wq + k;
            float x = (float)w + neg_shift;
            const bool gate = (unsigned)w < (unsigned)W && x >= 0.0f && x <= (float)(W - 1);
            int x_lo = gate ? (int)x : -2, x_hi;
            if (x_lo >= W - 1) { x_hi = x_lo = W - 1; x = (float)x_lo; } else { x_hi = x_lo + 1; }
            const float lx = x - (float)x_lo;
            const float hx = 1.0f - lx;
            const float wt1 = 1.0f * hx, wt2 = 1.0f * lx;
            const float g = q.v[k];
            const float g1 = g * wt1, g2 = g * wt2;
            if (gate && x_lo == ix && wt1 >= kMinWeight) acc_r = acc_r + g1;
            if (gate && x_hi == ix && wt2 >= kMinWeight) acc_r = acc_r + g2;
        }
    };
    // the pair (w, w + 1), w = ix + floor(s); the 8-byte load is kept inside the row (W >= 2): a shifted pair still
    // holds every in-row candidate, and a column that is not a candidate can never match
    auto load_pair = [&](int d, float s, Pair2 &q, int &wq) {
        // floor(s), made safe for any float (negative, huge, NaN: fmaxf / fminf return the non-NaN operand): for s < 0 the two
        // candidates are still ix + floor(s) and the column after it, and whatever falls outside the row cannot match
        const int fs = (int)fminf(fmaxf(floorf(s), -(float)W - 2.0f), (float)W + 2.0f);
        int st = ix + fs;
        st = st > W - 2 ? W - 2 : st;
        st = st < 0 ? 0 : st;
        wq = st;
        q = *reinterpret_cast<const Pair2 *>(gr + (int64_t)d * hw + st);
    };
    int d = 0;
    for (; d + PL <= D; d += PL) {
        float s[PL], glv[PL];
        Pair2 q[PL];
        int wq[PL];
#pragma unroll
        for (int j = 0; j < PL; ++j) s[j] = sh[d + j];
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            glv[j] = WITH_LEFT ? gl[(int64_t)(d + j) * hw + ix] : 0.0f;
            load_pair(d + j, s[j], q[j], wq[j]);
        }
#pragma unroll
        for (int j = 0; j < PL; ++j) one_plane(s[j], glv[j], q[j], wq[j]);
    }
    for (; d < D; ++d) {
        Pair2 q;
        int wq;
        const float s = sh[d];
        const float glv = WITH_LEFT ? gl[(int64_t)d * hw + ix] : 0.0f;
        load_pair(d, s, q, wq);
        one_plane(s, glv, q, wq);
    }
    if (WITH_LEFT) grad_left[idx] = acc_l;
    grad_right[idx] = acc_r;
}

// Adjoint of the depth-class planes of the factored first convolution (snvc_conv3d_forward_ex): plane 0 was added to
// output depth 0, plane 2 to the last depth, plane 1 to every depth in between, so
//   out[n,c,0] = g[n,c,0],  out[n,c,2] = g[n,c,D-1],  out[n,c,1] = sum_{d=1..D-2} g[n,c,d]   (d ascending: one fixed order)
__global__ void __launch_bounds__(256)
depth_class_sums_kernel(const float *__restrict__ g, float *__restrict__ out, int D, int64_t HW4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW4) return;
    const int64_t nc = blockIdx.y;
    const f4 *gp = reinterpret_cast<const f4 *>(g) + nc * D * HW4 + i;
    f4 *op = reinterpret_cast<f4 *>(out) + nc * 3 * HW4 + i;
    f4 mid = f4(0.0f);
    for (int d = 1; d + 1 < D; ++d) mid = mid + gp[(int64_t)d * HW4];
    op[0] = gp[0];
    op[HW4] = mid;
    op[2 * HW4] = gp[(int64_t)(D - 1) * HW4];
}

template <typename T>
int launch_forward(const void *left, const void *right, const void *shift, void *out, int64_t N,
                   int64_t C, int64_t Hi, int64_t Wi, int64_t D, int64_t ds, hipStream_t st) {
    const int64_t H = Hi / ds, W = Wi / ds;
    const int64_t planes = N * 2 * C * D;
    if (planes == 0 || H * W == 0) return SNVC_OK;  // BuildCostVolume_cuda.cu:235-238
    const unsigned gy = (unsigned)(planes < 65535 ? planes : 65535);
    const bool fast = sizeof(T) == 4 && ds == 1 && (W % 4) == 0 &&
                      ((reinterpret_cast<uintptr_t>(left) | reinterpret_cast<uintptr_t>(right) |
                        reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (!left && !(fast && W / 4 <= 512 && N * C * H < ((int64_t)1 << 30)))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_cost_volume_forward_right: needs fp32, downsample 1, W % 4 == 0, W <= 2048");
    const int64_t W4 = W / 4;
    if (fast && W4 <= 512 && N * C * H < ((int64_t)1 << 30)) {
        // rows per workgroup: prefer RB*W*4 bytes to be a multiple of 128 (every workgroup then writes
        // whole cache lines per plane), with RB*W/4 threads <= 512
        int RB = 0;
        for (int r = 1; r <= 8 && r * W4 <= 512 && r <= H; ++r)
            if ((r * W * 4) % 128 == 0) RB = r;
        if (RB == 0) {
            RB = (int)(256 / W4);
            if (RB > 8) RB = 8;
            if (RB < 1) RB = 1;
        }
        if (RB > H) RB = (int)H;
        const int threads = (int)ceil_div<int64_t>(RB * W4, 64) * 64;
        const int hblocks = (int)ceil_div<int64_t>(H, RB);
        // enough workgroups to fill the chip without shortening the per-thread plane walk too much
        int dsplit = 1;
        while (N * C * hblocks * dsplit < 2048 && D / (dsplit * 2) >= 8) dsplit *= 2;
        const int dchunk = (int)ceil_div<int64_t>(D, dsplit);
        dim3 grid((unsigned)(N * C * hblocks), (unsigned)ceil_div<int64_t>(D, dchunk));
        const size_t lds = (size_t)(RB + 1) * W * sizeof(float);
        if (left)
            cost_volume_fwd_rows<true><<<grid, threads, lds, st>>>((const float *)left, (const float *)right,
                                                                   (const float *)shift, (float *)out, (int)C, (int)D,
                                                                   (int)H, (int)W, RB, hblocks, dchunk);
        else
            cost_volume_fwd_rows<false><<<grid, threads, lds, st>>>(nullptr, (const float *)right, (const float *)shift,
                                                                    (float *)out, (int)C, (int)D, (int)H, (int)W, RB,
                                                                    hblocks, dchunk);
    } else if (fast) {
        dim3 grid((unsigned)ceil_div<int64_t>(H * W / 4, 256), gy);
        cost_volume_fwd_f32x4<<<grid, 256, 0, st>>>((const float *)left, (const float *)right,
                                                    (const float *)shift, (float *)out, (int)C,
                                                    (int)D, (int)H, (int)W, planes);
    } else if (sizeof(T) == 8 && ds == 1 && (W % 2) == 0 &&
               ((reinterpret_cast<uintptr_t>(left) | reinterpret_cast<uintptr_t>(right) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        dim3 grid((unsigned)ceil_div<int64_t>(H * W / 2, 256), gy);
        cost_volume_fwd_f64x2<<<grid, 256, 0, st>>>((const double *)left, (const double *)right, (const double *)shift, (double *)out,
                                                    (int)C, (int)D, (int)H, (int)W, planes);
    } else {
        dim3 grid((unsigned)ceil_div<int64_t>(H * W, 256), gy);
        cost_volume_fwd_generic<T><<<grid, 256, 0, st>>>((const T *)left, (const T *)right,
                                                         (const T *)shift, (T *)out, (int)C, (int)D,
                                                         (int)H, (int)W, (int)ds, planes);
    }
    return check_launch("snvc_cost_volume_forward");
}

template <typename T>
int launch_backward(const void *grad, const void *shift, void *gl, void *gr, int64_t N, int64_t C,
                    int64_t H, int64_t W, int64_t D, int64_t ds, hipStream_t st) {
    const int64_t total = N * C * H * ds * W * ds;
    if (total == 0) return SNVC_OK;
    const int64_t blocks = ceil_div<int64_t>(total, 256);
    if constexpr (sizeof(T) == 4) {
        if (ds == 1 && W >= 2 && blocks < ((int64_t)1 << 31)) {
            cost_volume_bwd_rows_f32<true><<<dim3((unsigned)blocks), 256, 0, st>>>((const float *)grad, (const float *)shift, (float *)gl,
                                                                          (float *)gr, (int)C, (int)D, (int)H, (int)W, total);
            return check_launch("snvc_cost_volume_backward");
        }
    }
    cost_volume_bwd_gather<T><<<dim3((unsigned)blocks), 256, 0, st>>>(
        (const T *)grad, (const T *)shift, (T *)gl, (T *)gr, (int)C, (int)D, (int)H, (int)W, (int)ds,
        total);
    return check_launch("snvc_cost_volume_backward");
}

}  // namespace
}  // namespace snvc

extern "C" {

int snvc_cost_volume_forward(const void *left, const void *right, const void *shift, void *out,
                             int64_t N, int64_t C, int64_t Hi, int64_t Wi, int64_t D,
                             int64_t downsample, int dtype, void *stream) {
    using namespace snvc;
    if (N < 0 || C < 0 || Hi < 0 || Wi < 0 || D < 0 || downsample < 1)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_cost_volume_forward: negative size or downsample < 1");
    if (Hi % downsample || Wi % downsample)
        return fail(SNVC_ERR_INVALID_ARGUMENT,
                    "snvc_cost_volume_forward: H and W must be multiples of downsample");
    if (Hi * Wi >= (int64_t)1 << 31)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_cost_volume_forward: feature plane exceeds 2^31 elements");
    if (N * 2 * C * D * (Hi / downsample) * (Wi / downsample) > 0 && (!left || !right || !shift || !out))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_cost_volume_forward: null pointer");
    if (dtype == SNVC_F32)
        return launch_forward<float>(left, right, shift, out, N, C, Hi, Wi, D, downsample, as_stream(stream));
    if (dtype == SNVC_F64)
        return launch_forward<double>(left, right, shift, out, N, C, Hi, Wi, D, downsample, as_stream(stream));
    return fail(SNVC_ERR_UNSUPPORTED, "snvc_cost_volume_forward: dtype must be f32 or f64 (AT_DISPATCH_FLOATING_TYPES)");
}

int snvc_cost_volume_forward_right(const float *right, const float *shift, float *out, int64_t N, int64_t C,
                                   int64_t Hi, int64_t Wi, int64_t D, int64_t downsample, void *stream) {
    using namespace snvc;
    if (N < 0 || C < 0 || Hi < 0 || Wi < 0 || D < 0 || downsample != 1)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_cost_volume_forward_right: downsample must be 1 and sizes non-negative");
    if (N * C * D * Hi * Wi == 0) return SNVC_OK;
    if (!right || !shift || !out) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_cost_volume_forward_right: null pointer");
    return launch_forward<float>(nullptr, right, shift, out, N, C, Hi, Wi, D, 1, as_stream(stream));
}

int snvc_cost_volume_backward_right(const float *grad_right_half, const float *shift, float *grad_right, int64_t N, int64_t C,
                                    int64_t H, int64_t W, int64_t D, void *stream) {
    using namespace snvc;
    if (N < 0 || C < 0 || H < 0 || W < 0 || D < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_cost_volume_backward_right: negative size");
    const int64_t total = N * C * H * W;
    if (total == 0) return SNVC_OK;
    if (W < 2 || H * W >= ((int64_t)1 << 31) || ceil_div<int64_t>(total, 256) >= ((int64_t)1 << 31))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_cost_volume_backward_right: needs 2 <= W and a plane below 2^31 elements");
    if (!grad_right || (D > 0 && (!grad_right_half || !shift)))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_cost_volume_backward_right: null pointer");
    cost_volume_bwd_rows_f32<false><<<dim3((unsigned)ceil_div<int64_t>(total, 256)), 256, 0, as_stream(stream)>>>(
        grad_right_half, shift, nullptr, grad_right, (int)C, (int)D, (int)H, (int)W, total);
    return check_launch("snvc_cost_volume_backward_right");
}

int snvc_depth_class_sums(const float *g, float *out, int64_t NC, int64_t D, int64_t HW, void *stream) {
    using namespace snvc;
    if (NC < 0 || D < 2 || HW < 0 || HW % 4 != 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_depth_class_sums: needs D >= 2 and H*W % 4 == 0");
    if (NC == 0 || HW == 0) return SNVC_OK;
    if (!g || !out || ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(out)) & 15))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_depth_class_sums: null or unaligned pointer");
    if (NC > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_depth_class_sums: N*C > 65535");
    depth_class_sums_kernel<<<dim3((unsigned)ceil_div<int64_t>(HW / 4, 256), (unsigned)NC), 256, 0, as_stream(stream)>>>(g, out, (int)D, HW / 4);
    return check_launch("snvc_depth_class_sums");
}

int snvc_cost_volume_backward(const void *grad, const void *shift, void *grad_left, void *grad_right,
                              int64_t N, int64_t C, int64_t H, int64_t W, int64_t D,
                              int64_t downsample, int dtype, void *stream) {
    using namespace snvc;
    if (N < 0 || C < 0 || H < 0 || W < 0 || D < 0 || downsample < 1)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_cost_volume_backward: negative size or downsample < 1");
    if (H * downsample * W * downsample >= (int64_t)1 << 31)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_cost_volume_backward: feature plane exceeds 2^31 elements");
    if (N * C * H * W > 0 && (!grad_left || !grad_right || (D > 0 && (!grad || !shift))))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_cost_volume_backward: null pointer");
    if (dtype == SNVC_F32)
        return launch_backward<float>(grad, shift, grad_left, grad_right, N, C, H, W, D, downsample, as_stream(stream));
    if (dtype == SNVC_F64)
        return launch_backward<double>(grad, shift, grad_left, grad_right, N, C, H, W, D, downsample, as_stream(stream));
    return fail(SNVC_ERR_UNSUPPORTED, "snvc_cost_volume_backward: dtype must be f32 or f64");
}

}  // extern "C"
