// First 3D convolution over the WARPED half of a plane-sweep volume whose disparity planes are uniformly spaced
// (gfx950).  Replaces, for that case, build_cost_volume's right half + nn.Conv3d(k3) over it
// (snvc/extension/build_cost_volume/src/BuildCostVolume_cuda.cu:63-98, snvc/models/submodule.py:32-50).
//
// With shift[n][d] = (m0 + d) / q pixels (q = 1 or 2: whole- or half-pixel disparity steps; BASELINE.json configs[1] is
// linspace(0, 95.5, 192): q = 2, m0 = 0) the warped half of the volume is a SHEAR of one 2D image:
//     V[c][d][h][w] = Rq[c][h][q*w - d - m0],      Rq[u] = the reference's own lerp of the right feature at x = u / q
//                                                   (u even / q = 1: R[u/q];  u odd: 0.5*R[j] + 0.5*R[j+1];  x < 0: 0,
//                                                   BuildCostVolume_cuda.cu:88 gates x to [0, W-1]),
// so a 3x3x3 convolution over it only depends on (h, u = q*w - d - m0):
//     conv(V)[co][d][h][w] = sum_{c,kd,kh,kw} Wt[co][c][kd][kh][kw] * Rq[c][h+kh][u + (q*kw - kd)] = G[co][h][u],
// a 2D convolution of Rq with the 3 x (2q+3)-tap kernel K[kh][t] = sum_{q*kw - kd = t} Wt[kd][kh][kw]: 318 GFLOP become
// 3.4 (cfg2), and the warped volume is never built.  Three borders are not sheared: the zero padding in d removes the
// kd = -1 taps at d = 0 and the kd = +1 taps at d = D-1, and the zero padding in w removes the kw = +1 taps at w = W-1
// (the kw = -1 taps at w = 0 read u < 0, which is zero in Rq as well).  The host (models/stereo_volume.py) therefore
//   1. builds Rq on a padded grid                                      (sheared_upsample_kernel, here)
//   2. runs the depth-1 3 x 7 convolution twice: G over all columns, and G' (the kernel without its kw = +1 taps, used
//      at w = W-1) over the window of D + 6 columns that column reads    (conv3d.hip, desc.ksize_d = 1, ksize_h = 3, ksize = 7)
//      -- each for the three depth classes (first plane: kernel without kd = -1; interior; last plane: without kd = +1),
//      stacked as 3*Cout output channels
//   3. expands v1[co][d][h][w] = act(scale * (G or G')[class(d)][co][h][q*w - d - m0] + planes[co][class(d)][h][w] + bias)
//      -- a pure 0.74 GB write stream                                   (sheared_expand_kernel, here)
// `planes` are the depth-class planes of the LEFT half of the concat volume (snvc_conv3d_forward_ex); the result equals
// the factored path's (tests/test_gpu_parity.py::test_sheared_first_conv_*), fp32 summation order aside.
#include "common.hpp"
#include "elementwise_internal.hpp"

namespace snvc {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#pragma clang fp contract(off)

// out[n][c][h][i], i = u + off:  Rq[u] for 0 <= u <= q*(W-1), else 0
__global__ void __launch_bounds__(256)
sheared_upsample_kernel(const float *__restrict__ r, float *__restrict__ out, int W, int q, int WU, int off, int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * WU) return;
    const int64_t row = idx / WU;
    const int i = (int)(idx - row * WU), u = i - off;
    float v = 0.0f;
    if (u >= 0 && u <= q * (W - 1)) {
        const float *rr = r + row * W;
        const int j = u / q;
        if (u - j * q == 0) v = rr[j];
        else v = 0.5f * rr[j] + 0.5f * rr[j + 1];      // w1*v1 + w2*v2 of the reference at lx = 0.5 (products exact)
    }
    out[idx] = v;
}

// sheared_upsample_kernel with the result written as a split C8 pair [N][2][C/8][H][WU][8] (r5: the 3x7 layer that reads it runs in
// split mode): value * *mul_dev = hi + lo.  A thread = one position of one 8-channel group.
typedef _Float16 h8u __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256)
sheared_upsample_split_kernel(const float *__restrict__ r, _Float16 *__restrict__ yh, _Float16 *__restrict__ yl, const float *__restrict__ mul_dev,
                              int C, int H, int W, int q, int WU, int off, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // ((n * G + g) * H + h) * WU + i
    if (idx >= total) return;
    const int i = (int)(idx % WU), u = i - off;
    const int64_t t = idx / WU;
    const int h = (int)(t % H);
    const int64_t ng = t / H;
    const int G = (C + 7) >> 3, g = (int)(ng % G);
    const int64_t n = ng / G;
    const float mul = mul_dev[0];
    h8u hi, lo;
    const bool in = u >= 0 && u <= q * (W - 1);
    const int j = in ? u / q : 0;
    const bool whole = u - j * q == 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float v = 0.0f;
        const int co = g * 8 + c;
        if (in && co < C) {
            const float *rr = r + ((n * C + co) * (int64_t)H + h) * W;
            v = whole ? rr[j] : 0.5f * rr[j] + 0.5f * rr[j + 1];
        }
        v *= mul;
        hi[c] = (_Float16)v;
        lo[c] = (_Float16)(v - (float)hi[c]);
    }
    // [N][2 (hi | lo)][G][H][WU] pieces: a sample's two planes sit next to each other
    const int64_t pos = ((n * 2 * G + g) * (int64_t)H + h) * WU + i;
    *reinterpret_cast<h8u *>(yh + pos * 8) = hi;
    *reinterpret_cast<h8u *>(yl + pos * 8) = lo;
}

// One workgroup = RB rows of one (n, co) plane; thread = (row, quad of 4 columns) and walks d = 1 .. D-2.
// Window: the thread's four values G[i0 + q*k], k = 0..3, with i0 = q*4*quad - d - m0 + off.  For q = 2 the even- and the
// odd-indexed elements of a G row are kept as two LDS arrays, so a window is 4 CONSECUTIVE elements of one of them and
// moving from plane d to d + q shifts it down by one: one 4-byte LDS read per thread and plane.
// MODE 0 writes y; MODE 1 (training: train-mode BatchNorm needs the statistics of the layer's raw result before it can write
// anything) walks the same values and leaves the workgroup's fp64 (sum, sum of squares) in stats[(n*C + co) * gridDim.x + blockIdx.x]
// -- the raw result is never stored: MODE 0 then writes act(scale * raw + shift) directly, and the backward pass
// (sheared_bwd_kernel) recomputes raw from G the same way.
template <int Q, int MODE>
__global__ void __launch_bounds__(512)
sheared_expand_kernel(const float *__restrict__ g, const float *__restrict__ gcol, const float *__restrict__ planes,
                      const float *__restrict__ scale, const float *__restrict__ bias, float *__restrict__ y,
                      double *__restrict__ stats, int C, int D, int H, int W, int m0, int WG, int off, int WG2, int off2, int RB,
                      int flags) {
    extern __shared__ float lds[];
    const int quads = W >> 2;
    const int tid = threadIdx.x;
    const int co = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int h0 = blockIdx.x * RB;
    const int rows = (H - h0) < RB ? (H - h0) : RB;
    const int LW = (WG + Q - 1) / Q + 4;                 // elements per phase array of a row (+ slack for the window start)
    float *const phase = lds;                            // [RB][Q][LW]
    float *const lastcol = lds + RB * Q * LW;            // [RB][D]: G' at the last column, per plane
    // g / gcol are [N][3][C][H][*]: depth class 0 = first plane (no kd = -1 taps), 1 = interior, 2 = last plane (no kd = +1)
    const float *gn = g + ((n * 3 + 1) * C + co) * (int64_t)H * WG, *gc = gcol + ((n * 3 + 1) * C + co) * (int64_t)H * WG2;
    for (int e = tid; e < rows * WG; e += blockDim.x) {
        const int r = e / WG, i = e - r * WG;
        phase[(r * Q + (i % Q)) * LW + i / Q] = gn[(int64_t)(h0 + r) * WG + i];
    }
    for (int e = tid; e < rows * D; e += blockDim.x) {
        const int r = e / D, d = e - r * D;
        const int i = Q * (W - 1) - d - m0 + off2;
        lastcol[r * D + d] = (i >= 0 && i < WG2) ? gc[(int64_t)(h0 + r) * WG2 + i] : 0.0f;
    }
    __syncthreads();
    double st0 = 0.0, st1 = 0.0;
    unsigned mxb = 0;            // MODE 2 (r6): bits of max|y| over what this thread stores, published into SNVC_AMAX_SLOTS words at `stats`
    auto amax4 = [&](const f32x4 &o) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned b = __float_as_uint(o[k]) & 0x7fffffffu;
            mxb = b > mxb ? b : mxb;
        }
    };
    if (tid < rows * quads) {
    const int r = tid / quads, qd = tid - r * quads, w0 = 4 * qd;
    const float sc = scale ? scale[co] : 1.0f, bi = scale ? bias[co] : 0.0f;
    f32x4 pl = {0.0f, 0.0f, 0.0f, 0.0f};
    if (planes)      // class 1 = interior planes of the left half
        pl = *reinterpret_cast<const f32x4 *>(planes + (((n * C + co) * 3 + 1) * (int64_t)H + h0 + r) * W + w0);
    const bool relu = (flags & SNVC_EPI_RELU) != 0;
    const bool last = qd == quads - 1;
    const int64_t plane_sz = (int64_t)H * W;
    float *yp = y + ((n * C + co) * (int64_t)D) * plane_sz + (int64_t)(h0 + r) * W + w0;

    // window of chain p (planes d with (d + m0) % Q == p): element k is phase[p'][j + k], index i = Q*(w0 + k) - d - m0 + off
    auto load = [&](int i) -> float {         // G[row r][i], zero outside [0, WG)
        return (i >= 0 && i < WG) ? phase[(r * Q + (i % Q)) * LW + i / Q] : 0.0f;
    };
    float win[Q][4];
#pragma unroll
    for (int p = 0; p < Q; ++p) {
        const int d = 1 + p;                  // first plane of the chain that starts at d = 1 + p
#pragma unroll
        for (int k = 0; k < 4; ++k) win[p][k] = load(Q * (w0 + k) - d - m0 + off);
    }
    // the two end planes: their own G / G' (depth classes 0 and 2) and planes of the left half, read straight from L2
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int d = e ? D - 1 : 0, cls = e ? 2 : 0;
        const float *ge = g + ((n * 3 + cls) * C + co) * (int64_t)H * WG + (int64_t)(h0 + r) * WG;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = Q * (w0 + k) - d - m0 + off;
            o[k] = (i >= 0 && i < WG) ? ge[i] : 0.0f;
        }
        if (last) {
            const int i = Q * (W - 1) - d - m0 + off2;
            o[3] = (i >= 0 && i < WG2) ? gcol[((n * 3 + cls) * C + co) * (int64_t)H * WG2 + (int64_t)(h0 + r) * WG2 + i] : 0.0f;
        }
        f32x4 pe = {0.0f, 0.0f, 0.0f, 0.0f};
        if (planes) pe = *reinterpret_cast<const f32x4 *>(planes + (((n * C + co) * 3 + cls) * (int64_t)H + h0 + r) * W + w0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = (o[k] + pe[k]) * sc + bi;
            o[k] = relu ? (v > 0.0f ? v : 0.0f) : v;
        }
        if (MODE == 0 || MODE == 2) {
            *reinterpret_cast<f32x4 *>(yp + (int64_t)d * plane_sz) = o;
            if (MODE == 2) amax4(o);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { st0 += (double)o[k]; st1 += (double)o[k] * (double)o[k]; }
        }
    }
    for (int d0 = 1; d0 < D - 1; d0 += Q) {
#pragma unroll
        for (int p = 0; p < Q; ++p) {
            const int d = d0 + p;
            if (d >= D - 1) break;
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = win[p][k];
            if (last) o[3] = lastcol[r * D + d];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = (o[k] + pl[k]) * sc + bi;
                o[k] = relu ? (v > 0.0f ? v : 0.0f) : v;
            }
            if (MODE == 0 || MODE == 2) {
                *reinterpret_cast<f32x4 *>(yp + (int64_t)d * plane_sz) = o;      // plain store: conv2 reads the tail from the caches
                if (MODE == 2) amax4(o);
            } else {
                st0 += ((double)o[0] + (double)o[1]) + ((double)o[2] + (double)o[3]);
                st1 += ((double)o[0] * (double)o[0] + (double)o[1] * (double)o[1]) + ((double)o[2] * (double)o[2] + (double)o[3] * (double)o[3]);
            }
            // next plane of this chain is d + Q: every index drops by Q, i.e. by ONE element of the chain's phase array
            win[p][3] = win[p][2]; win[p][2] = win[p][1]; win[p][1] = win[p][0];
            win[p][0] = load(Q * w0 - (d + Q) - m0 + off);
        }
    }
    }   // active threads
    if (MODE == 2) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned v = (unsigned)__shfl_xor((int)mxb, o);
            mxb = v > mxb ? v : mxb;
        }
        if ((tid & 63) == 0 && mxb)
            atomicMax(reinterpret_cast<unsigned *>(stats) + ((blockIdx.x + 7u * blockIdx.y + 13u * blockIdx.z) & (SNVC_AMAX_SLOTS - 1)), mxb);
    }
    if (MODE == 1) {
        for (int o = 32; o > 0; o >>= 1) { st0 += __shfl_down(st0, o, 64); st1 += __shfl_down(st1, o, 64); }
        __shared__ double red[16];
        const int wave = tid >> 6, nw = (int)(blockDim.x >> 6);
        if ((tid & 63) == 0) { red[2 * wave] = st0; red[2 * wave + 1] = st1; }
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < nw; ++k) { st0 += red[2 * k]; st1 += red[2 * k + 1]; }
            double *o = stats + (((int64_t)n * C + co) * gridDim.x + blockIdx.x) * 2;
            o[0] = st0; o[1] = st1;
        }
    }
}

// Structure of a shift array in one launch (the host reads 4 floats back: the one device -> host sync that replaces the
// `assert torch.all(shift >= 0)` of the reference's wrapper, snvc/extension/build_cost_volume/__init__.py:12):
//   out = { all shift >= 0,  every row == shift[0][0] + d,  every row == shift[0][0] + d/2,  shift[0][0] }
// compared exactly in fp32 (s0 + d and s0 + 0.5*d are formed as the host would form them: one rounding each).
__global__ void __launch_bounds__(256)
shift_structure_kernel(const float *__restrict__ shift, float *__restrict__ out, int64_t total, int D) {
    __shared__ int flags[3];
    if (threadIdx.x < 3) flags[threadIdx.x] = 1;
    __syncthreads();
    const float s0 = shift[0];
    bool nonneg = true, whole = true, half_ = true;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const float v = shift[i], d = (float)(int)(i % D);
        nonneg = nonneg && v >= 0.0f;                 // NaN fails, as torch.all(shift >= 0) does
        whole = whole && v == s0 + d;
        half_ = half_ && v == s0 + 0.5f * d;
    }
    if (!nonneg) flags[0] = 0;
    if (!whole) flags[1] = 0;
    if (!half_) flags[2] = 0;
    __syncthreads();
    if (threadIdx.x < 3) out[threadIdx.x] = (float)flags[threadIdx.x];
    if (threadIdx.x == 3) out[3] = s0;
}

// ------------------------------------------------------------------------------------ any shift array: warp AFTER the convolution
// For an ARBITRARY shift array the warped half is still a per-plane horizontal resampling of ONE image, V[c][d][h][w'] =
// lerp(R0[c][h][:], w' - s_d) (R0 = the right feature, zero outside the image; BuildCostVolume_cuda.cu:63-98), and a linear
// interpolation along w commutes with a convolution along (c, h, w).  So
//     conv3d(V)[co][d][h][w] = sum_{kd, 0 <= d+kd-1 < D}  lerp(P_kd[co][h][:], w - s_{d+kd-1}),     P_kd = conv2d(R0, Wt[:, :, kd])
// -- three 2D convolutions of the feature, computed ONCE, and three interpolations per output voxel instead of 27 x C
// multiply-adds -- up to the two places where the reference is NOT "interpolate the zero-extended signal":
//   (a) zero padding of the 3D convolution on the right: at w = W-1 the kw = +1 tap reads V[W] = 0, while P_kd there includes
//       lerp(R0, W - s) != 0.  Q_kd[co][h][j] = sum_{c,kh} Wt[co][c][kd][kh][kw=+1] R0[c][h+kh-1][j] (the same depth-1 launch,
//       kernels with only their centre column set) gives that term exactly: out[W-1] -= lerp(Q_kd, W - s).
//   (b) the left edge, two single-column terms around w = m = floor(s), both multiples of E[kd][kw][co][h] = the (c, kh)
//       contraction of the image's FIRST column with the kw taps (a depth-1 launch on a 4-column slab): the correlation at
//       column -1, P_kd[-1] = E[kd][+1], which zero-extended rows lack; and the gate x >= 0 (BuildCostVolume_cuda.cu:88): for a
//       fractional s the sample at w' = m has x = -frac(s) < 0 and is ZERO in the reference while the zero-extended
//       interpolation gives (1 - frac(s)) * R[0]; it feeds the outputs w = m + 1 - kw through one kw tap each.
// The interpolation's right edge needs nothing: for x in (W-2, W-1] the zero-extended lerp is the reference's clamped one.
// Same values as the 3D convolution over the built volume up to fp32 summation order (tests: random shift arrays against
// that path and against the oracle).  p / q: [N][3 kd][C][H][W]; e: [N][3 kd][3 kw][C][H][4] (column 0 used).
// Inner loop without a data-dependent branch: the border terms are folded into the staged rows.
//   rows F (the f-term reads them): P_kd zero-extended, with the one column the zero extension gets wrong set right: F[-1] = E2
//   rows G (the g-term reads them for a fractional shift): F with the gated sample's share removed, G[-1] = 0, G[0] = P[0] - E1,
//          G[1] = P[1] - E0  (for a whole-pixel shift the g-term reads F: nothing is gated)
//   ta[kd][row][dd] = the last column's (a) term f * Q0[W-m-1] + g * Qg[W-m] of plane dd (Qg[0] = 0 for a fractional shift),
//          tabulated once per workgroup: no global load inside the walk
__global__ void __launch_bounds__(512)
warped_expand_kernel(const float *__restrict__ p, const float *__restrict__ q, const float *__restrict__ e,
                     const float *__restrict__ planes, const float *__restrict__ shift, const float *__restrict__ scale,
                     const float *__restrict__ bias, float *__restrict__ y, int C, int D, int H, int W, int RB, int flags) {
    // rows [2 (F | G)][3 kd][RB][LW], each with W + 4 zeros in front and 4 behind: the interpolation reads columns
    // w - m - 1 .. w - m + 3 (0 <= m <= W) without a bounds test
    extern __shared__ float lds[];
    const int quads = W >> 2, LW = 2 * W + 8, Z = W + 4;
    const int tid = threadIdx.x;
    const int co = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int h0 = blockIdx.x * RB;
    const int rows = (H - h0) < RB ? (H - h0) : RB;
    const int64_t hw = (int64_t)H * W;
    float *const rowsG = lds + 3 * RB * LW;
    float *const ta = rowsG + 3 * RB * LW;              // [3][RB][D]
    const float *sh = shift + n * D;
    for (int i = tid; i < 3 * rows * (LW >> 2); i += blockDim.x) {
        const int row = i / (LW >> 2), pc = i - row * (LW >> 2);          // piece pc of LDS row (kd, r)
        const int kd = row / rows, r = row - kd * rows, col = 4 * pc - Z;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (col >= 0 && col < W) v = *reinterpret_cast<const f32x4 *>(p + (((n * 3 + kd) * C + co) * (int64_t)H + h0 + r) * W + col);
        f32x4 g4 = v;
        const float *er = e + (((n * 3 + kd) * 3) * C + co) * (int64_t)H * 4 + (int64_t)(h0 + r) * 4;     // E[kd][kw][co][h][0], kw stride C*H*4
        const int64_t kws = (int64_t)C * H * 4;
        if (col == -4) { v[3] = er[2 * kws]; g4[3] = 0.0f; }              // column -1
        if (col == 0) { g4[0] = v[0] - er[kws]; g4[1] = v[1] - er[0]; }   // columns 0, 1
        *reinterpret_cast<f32x4 *>(lds + (kd * RB + r) * LW + 4 * pc) = v;
        *reinterpret_cast<f32x4 *>(rowsG + (kd * RB + r) * LW + 4 * pc) = g4;
    }
    for (int i = tid; i < 3 * rows * D; i += blockDim.x) {
        const int kd = i / (rows * D), r = (i / D) % rows, dd = i % D;
        const float s = sh[dd];
        float t = 0.0f;
        if (s <= (float)W) {
            const float mf = __builtin_floorf(s), f = s - mf, g = 1.0f - f;
            const int m = (int)mf, i0 = W - m - 1;
            const float *qr = q + (((n * 3 + kd) * C + co) * (int64_t)H + h0 + r) * W;
            const float a0 = (unsigned)i0 < (unsigned)W ? qr[i0] : 0.0f;
            float a1 = (unsigned)(i0 + 1) < (unsigned)W ? qr[i0 + 1] : 0.0f;
            if (i0 + 1 == 0 && f > 0.0f) a1 = 0.0f;                       // Qg[0]: the gated sample's share
            t = f * a0 + g * a1;
        }
        ta[(kd * RB + r) * D + dd] = t;
    }
    __syncthreads();
    if (tid >= rows * quads) return;
    const int r = tid / quads, qd = tid - r * quads, w0 = 4 * qd, h = h0 + r;
    const bool last = qd == quads - 1;
    const float sc = scale ? scale[co] : 1.0f, bi = scale ? bias[co] : 0.0f;
    const bool relu = (flags & SNVC_EPI_RELU) != 0;
    f32x4 pl[3];
#pragma unroll
    for (int cls = 0; cls < 3; ++cls) {
        pl[cls] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (planes) pl[cls] = *reinterpret_cast<const f32x4 *>(planes + (((n * C + co) * 3 + cls) * (int64_t)H + h) * W + w0);
    }
    float *yp = y + ((n * C + co) * (int64_t)D) * hw + (int64_t)h * W + w0;
    const int lane_off = r * LW + Z + w0 - 4;           // + kd * RB * LW - m, rounded down to a piece: the quad's window
    const float *tar = ta + r * D;
    // the shifts of planes d-1, d, d+1 ride in scalar registers; the next one is requested an iteration ahead.  +inf marks
    // "no such plane" (zero padding in depth) and takes the same exit as a shift beyond the image
    const float none = __builtin_inff();
    float sk[3] = {none, sh[0], D > 1 ? sh[1] : none};
    for (int d = 0; d < D; ++d) {
        const float snext = d + 2 < D ? sh[d + 2] : none;
        f32x4 o = d == 0 ? pl[0] : (d == D - 1 ? pl[2] : pl[1]);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            // wave-uniform, and SAID so: m, f and the alignment pick below live in scalar registers, every branch on them is scalar
            const float s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sk[kd])));
            if (!(s <= (float)W)) continue;             // no plane, or every sample left of the image
            const float mf = __builtin_floorf(s), f = s - mf, g = 1.0f - f;
            const int m = (int)mf;
            // out[w] += f * F[w - m - 1] + g * Gsel[w - m]: the window F[w0 - m - 1 .. w0 - m + 3] starts (3 - m) & 3 floats past a
            // 16-byte boundary that is the same for every lane: three aligned 16-byte reads per row and a wave-uniform pick
            const int a = (3 - m) & 3, mb = m + a - 3;  // mb = 4 * (m / 4)
            const float *fr = lds + kd * RB * LW + lane_off - mb;
            const float *gr = (f > 0.0f ? rowsG : lds) + kd * RB * LW + lane_off - mb;
            const f32x4 FA = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(fr, 16));
            const f32x4 FB = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(fr + 4, 16));
            const f32x4 GA = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(gr, 16));
            const f32x4 GB = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(gr + 4, 16));
            // the eight floats read hold window element i (column w0 - m - 1 + i) at position a + i: F needs i = 0..3, G i = 1..4
            float fv[4], gv[4];
            if (a == 0) { fv[0] = FA[0]; fv[1] = FA[1]; fv[2] = FA[2]; fv[3] = FA[3]; gv[0] = GA[1]; gv[1] = GA[2]; gv[2] = GA[3]; gv[3] = GB[0]; }
            else if (a == 1) { fv[0] = FA[1]; fv[1] = FA[2]; fv[2] = FA[3]; fv[3] = FB[0]; gv[0] = GA[2]; gv[1] = GA[3]; gv[2] = GB[0]; gv[3] = GB[1]; }
            else if (a == 2) { fv[0] = FA[2]; fv[1] = FA[3]; fv[2] = FB[0]; fv[3] = FB[1]; gv[0] = GA[3]; gv[1] = GB[0]; gv[2] = GB[1]; gv[3] = GB[2]; }
            else { fv[0] = FA[3]; fv[1] = FB[0]; fv[2] = FB[1]; fv[3] = FB[2]; gv[0] = GB[0]; gv[1] = GB[1]; gv[2] = GB[2]; gv[3] = GB[3]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] += f * fv[k] + g * gv[k];
            const float tl = tar[kd * RB * D + d + kd - 1];
            o[3] -= last ? tl : 0.0f;                   // (a): the kw = +1 tap of the last column reads the zero padding
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = o[k] * sc + bi;
            o[k] = relu ? (t > 0.0f ? t : 0.0f) : t;
        }
        *reinterpret_cast<f32x4 *>(yp + (int64_t)d * hw) = o;
        sk[0] = sk[1]; sk[1] = sk[2]; sk[2] = snext;
    }
}

// r4 form of the same layer: the interpolation windows live in REGISTERS across the walk over d.
// Output plane d adds, for kd = 0..2, the interpolation of row P_kd at the shift of plane e = d + kd - 1; as d advances every kd
// sees the same sequence of shifts (one plane apart), so the five floats F_kd[w0 - m - 1 .. w0 - m + 3] a quad needs move only when
// m = floor(s_e) moves -- in a plane sweep by 0, +-1 or +-2 columns per plane.  The window (one for the F rows, one for the G rows,
// per kd) is then shifted in registers and refilled with |dm| 4-byte LDS reads; only a jump re-reads it (two aligned 16-byte
// pieces per row and the wave-uniform alignment pick of the r3 form).  r3 read four 16-byte pieces per kd and plane (192 B of
// LDS per 16 B stored: 0.288 ms at cfg2, LDS-bound); on cfg2's half-pixel steps this form reads 3 x 2 x 4 B every other plane.
// A whole-pixel shift s = m is evaluated as (m - 1, f = 1, g = 0): F[w - m] with nothing gated, which is what the reference
// computes there, so the choice "G rows or F rows for the g-term" disappears from the loop.
// Rows are staged with 12 zeros in front and >= 4 behind (LW = W + 16 instead of 2 W + 8: 26 KB per workgroup at cfg2, every
// workgroup of the launch resident at once); a window that starts left of them is all zeros and reads the zero pad (clamped address).
// tas[r][d] = the last column's zero-padding term of output plane d, summed over kd once per workgroup.
__global__ void __launch_bounds__(512)
warped_expand_win_kernel(const float *__restrict__ p, const float *__restrict__ q, const float *__restrict__ e,
                         const float *__restrict__ planes, const float *__restrict__ shift, const float *__restrict__ scale,
                         const float *__restrict__ bias, float *__restrict__ y, int C, int D, int H, int W, int RB, int flags) {
    extern __shared__ float lds[];
    constexpr int FP = 12;                               // zero columns in front of column 0 (column -1 is staged, not zero)
    constexpr int kInvalid = (int)0x80000001, kNone = (int)0x80000000;
    const int quads = W >> 2, LW = W + 16;
    const int tid = threadIdx.x;
    const int co = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int h0 = blockIdx.x * RB;
    const int rows = (H - h0) < RB ? (H - h0) : RB;
    const int64_t hw = (int64_t)H * W;
    float *const rowsG = lds + 3 * RB * LW;
    float *const tas = rowsG + 3 * RB * LW;              // [RB][D]
    f32x4 *const ptab = reinterpret_cast<f32x4 *>(tas + RB * D + ((4 - ((RB * D) & 3)) & 3));    // [D + 2]: {m, f, g, valid} of plane d
    const float *sh = shift + n * D;
    // plane parameters, once per workgroup (the walk reads them back as broadcast LDS reads: no scalar memory load inside the
    // loop, whose out-of-order return would force every LDS wait in it down to lgkmcnt(0))
    for (int i = tid; i < D + 2; i += blockDim.x) {
        f32x4 t = {__builtin_bit_cast(float, kInvalid), 0.0f, 0.0f, 0.0f};      // no such plane / every sample left of the image: adds 0
        if (i < D) {
            const float s = sh[i];
            if (s <= (float)W) {
                const float mf = __builtin_floorf(s);
                float ff = s - mf;
                int mm = (int)mf;
                if (ff == 0.0f) { mm -= 1; ff = 1.0f; }      // whole pixel: F[w - m], nothing gated
                t = f32x4{__builtin_bit_cast(float, mm), ff, 1.0f - ff, 0.0f};
            }
        }
        ptab[i] = t;
    }
    for (int i = tid; i < 3 * rows * (LW >> 2); i += blockDim.x) {
        const int row = i / (LW >> 2), pc = i - row * (LW >> 2);
        const int kd = row / rows, r = row - kd * rows, col = 4 * pc - FP;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (col >= 0 && col < W) v = *reinterpret_cast<const f32x4 *>(p + (((n * 3 + kd) * C + co) * (int64_t)H + h0 + r) * W + col);
        f32x4 g4 = v;
        const float *er = e + (((n * 3 + kd) * 3) * C + co) * (int64_t)H * 4 + (int64_t)(h0 + r) * 4;
        const int64_t kws = (int64_t)C * H * 4;
        if (col == -4) { v[3] = er[2 * kws]; g4[3] = 0.0f; }              // column -1: F = E2, G = 0
        if (col == 0) { g4[0] = v[0] - er[kws]; g4[1] = v[1] - er[0]; }   // columns 0, 1 of G: the gated sample's share removed
        *reinterpret_cast<f32x4 *>(lds + (kd * RB + r) * LW + 4 * pc) = v;
        *reinterpret_cast<f32x4 *>(rowsG + (kd * RB + r) * LW + 4 * pc) = g4;
    }
    __syncthreads();                                     // ptab is complete
    for (int i = tid; i < rows * D; i += blockDim.x) {
        const int r = i / D, d = i - r * D;
        // the three kd terms' plane parameters from ptab (no dependent global load), their six q values requested together
        float fk[3], gk[3], a0[3], a1[3];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const int ee = d + kd - 1;
            const f32x4 t = ptab[ee < 0 ? D : ee];       // entry D: "no such plane"
            const int mm = __builtin_bit_cast(int, t[0]);
            const bool valid = mm != kInvalid;
            // ptab holds a whole-pixel shift s = m as (m - 1, f = 1, g = 0): Q0[W - m - 1] = Q[W - (m - 1) - 2]: same element
            const int i0 = W - mm - 1;                   // f * Q[i0] + g * Q[i0 + 1]
            const float *qr = q + (((n * 3 + kd) * C + co) * (int64_t)H + h0 + r) * W;
            fk[kd] = t[1]; gk[kd] = t[2];
            a0[kd] = (valid && (unsigned)i0 < (unsigned)W) ? qr[i0] : 0.0f;
            a1[kd] = (valid && (unsigned)(i0 + 1) < (unsigned)W) ? qr[i0 + 1] : 0.0f;
            if (i0 + 1 == 0) a1[kd] = 0.0f;              // Qg[0]: the gated sample's share (only reached with g > 0, i.e. a fractional shift)
        }
        tas[r * D + d] = (fk[0] * a0[0] + gk[0] * a1[0]) + (fk[1] * a0[1] + gk[1] * a1[1]) + (fk[2] * a0[2] + gk[2] * a1[2]);
    }
    __syncthreads();
    if (tid >= rows * quads) return;
    const int r = tid / quads, qd = tid - r * quads, w0 = 4 * qd, h = h0 + r;
    const bool last = qd == quads - 1;
    const float sc = scale ? scale[co] : 1.0f, bi = scale ? bias[co] : 0.0f;
    const bool relu = (flags & SNVC_EPI_RELU) != 0;
    f32x4 pl[3];
#pragma unroll
    for (int cls = 0; cls < 3; ++cls) {
        pl[cls] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (planes) pl[cls] = *reinterpret_cast<const f32x4 *>(planes + (((n * C + co) * 3 + cls) * (int64_t)H + h) * W + w0);
    }
    float *yp = y + ((n * C + co) * (int64_t)D) * hw + (int64_t)h * W + w0;
    const float *tar = tas + r * D;
    const int lane_idx = r * LW + FP + w0 - 1;          // index of window element 0 in an LDS row set, before "- m" and "+ kd * RB * LW"
    const int row_lo = r * LW;                           // clamp: a window that starts left of the row's zero pad reads the pad
    auto uni_i = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni_f = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    auto at = [&](const float *rows_, int idx) { return rows_[idx < row_lo ? row_lo : idx]; };

    // The walk is bound by the VALU instructions a wave issues per plane (~120 at 65 % of HBM: a 1 KB store per wave and plane),
    // so everything wave-uniform lives in scalar registers and the per-plane vector work is 12 packed FMAs + the epilogue:
    //   * plane parameters {m, f, g} come from ptab (one broadcast LDS read a step ahead, three v_readfirstlane); a plane that
    //     does not exist or lies wholly left of the image has f = g = 0 and m = kInvalid: it adds 0 and moves no window
    //   * window kd = F[kd] at columns w0 - mw - 1 .. w0 - mw + 2 (the f-term's four) and G[kd] one column further (the g-term's
    //     four): two aligned register quads that the packed FMAs read in place
    int pm[3];
    float pf[3], pg[3];
    auto take = [&](const f32x4 &t, int &m, float &f, float &g) {
        m = uni_i(__builtin_bit_cast(int, t[0])); f = uni_f(t[1]); g = uni_f(t[2]);
    };
    pm[0] = kInvalid; pf[0] = pg[0] = 0.0f;
    take(ptab[0], pm[1], pf[1], pg[1]);
    take(ptab[1], pm[2], pf[2], pg[2]);                  // ptab has D + 2 entries, the last two invalid
    f32x4 nxt = ptab[2];
    f32x4 wf[3], wg[3];
    int mw[3] = {kNone, kNone, kNone};
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) wf[kd] = wg[kd] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // move window kd to column offset m (wave-uniform; only called when m != mw[kd])
    auto refill = [&](int kd, int m) {
        const float *fr = lds + kd * RB * LW, *gr = rowsG + kd * RB * LW;
        const int idx0 = lane_idx - m;
        const int dm = m - mw[kd];
        f32x4 &F = wf[kd], &G = wg[kd];
        if (mw[kd] != kNone && dm == 1) {
            F = f32x4{at(fr, idx0), F[0], F[1], F[2]};
            G = f32x4{at(gr, idx0 + 1), G[0], G[1], G[2]};
        } else if (mw[kd] != kNone && dm == 2) {
            F = f32x4{at(fr, idx0), at(fr, idx0 + 1), F[0], F[1]};
            G = f32x4{at(gr, idx0 + 1), at(gr, idx0 + 2), G[0], G[1]};
        } else if (dm == -1) {
            F = f32x4{F[1], F[2], F[3], at(fr, idx0 + 3)};
            G = f32x4{G[1], G[2], G[3], at(gr, idx0 + 4)};
        } else if (dm == -2) {
            F = f32x4{F[2], F[3], at(fr, idx0 + 2), at(fr, idx0 + 3)};
            G = f32x4{G[2], G[3], at(gr, idx0 + 3), at(gr, idx0 + 4)};
        } else {
            // a jump: two aligned 16-byte pieces per row; window column i sits at position a + i of the eight floats
            // (a = (3 - m) & 3 is the same for every lane: FP and w0 are multiples of 4)
            const int a = (3 - m) & 3;
            int base = idx0 - a;
            base = base < row_lo ? row_lo : base;        // left of the pad: eight zeros
            const f32x4 FA = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(fr + base, 16));
            const f32x4 FB = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(fr + base + 4, 16));
            const f32x4 GA = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(gr + base, 16));
            const f32x4 GB = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(gr + base + 4, 16));
            if (a == 0) { F = FA; G = f32x4{GA[1], GA[2], GA[3], GB[0]}; }
            else if (a == 1) { F = f32x4{FA[1], FA[2], FA[3], FB[0]}; G = f32x4{GA[2], GA[3], GB[0], GB[1]}; }
            else if (a == 2) { F = f32x4{FA[2], FA[3], FB[0], FB[1]}; G = f32x4{GA[3], GB[0], GB[1], GB[2]}; }
            else { F = f32x4{FA[3], FB[0], FB[1], FB[2]}; G = GB; }
        }
        mw[kd] = m;
    };

    for (int d = 0; d < D; ++d) {
        // (requesting the refills a step ahead, so that their LDS latency hides behind the arithmetic, was measured and is
        // SLOWER: the walk is bound by instruction issue, not by LDS latency; profiles/r4/kernel_experiments_r4.txt)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
            if (pm[kd] != mw[kd] && pm[kd] != kInvalid) refill(kd, pm[kd]);
        int nm;
        float nff, ngg;
        take(nxt, nm, nff, ngg);                         // plane d + 2: kd = 2 of the next step
        nxt = ptab[d + 3 < D + 2 ? d + 3 : D + 1];
        const float tl = tar[d];
        f32x4 o = d == 0 ? pl[0] : (d == D - 1 ? pl[2] : pl[1]);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const f32x4 f4 = {pf[kd], pf[kd], pf[kd], pf[kd]}, g4 = {pg[kd], pg[kd], pg[kd], pg[kd]};
            o = __builtin_elementwise_fma(g4, wg[kd], __builtin_elementwise_fma(f4, wf[kd], o));
        }
        o[3] -= last ? tl : 0.0f;                        // the kw = +1 taps of the last column read the zero padding
        const f32x4 sc4 = {sc, sc, sc, sc}, bi4 = {bi, bi, bi, bi};
        o = __builtin_elementwise_fma(o, sc4, bi4);
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_fmaxf(o[k], 0.0f);      // NaN -> 0, as the generic epilogues (v > 0 ? v : 0)
        }
        *reinterpret_cast<f32x4 *>(yp + (int64_t)d * hw) = o;
        pm[0] = pm[1]; pf[0] = pf[1]; pg[0] = pg[1];
        pm[1] = pm[2]; pf[1] = pf[2]; pg[1] = pg[2];
        pm[2] = nm; pf[2] = nff; pg[2] = ngg;
    }
}

// ------------------------------------------------------------------------------------ adjoint of the any-shift layer (r4)
// Backward of the first convolution over the warped half for an ARBITRARY shift array, without the warped volume, its
// gradient or a 3D kernel.  With V_e[w'] = f_e R[w' - m_e - 1] + g_e R[w' - m_e] for m_e + 1 <= w' <= W-1 and 0 elsewhere
// (BuildCostVolume_cuda.cu:63-98 for s_e = m_e + f_e >= 0: the gate x >= 0 is "w' > m_e", a whole-pixel shift is (m - 1, 1, 0))
//     raw[co][d][h][w] = sum_{c,kd,kh,kw} Wt[co][c][kd][kh][kw] * V_{d+kd-1}[c][h+kh-1][w+kw-1]      (zero padding in d and w)
// is linear in R through the nine shifted-and-warped-back gradients
//     A[kd][kw][co][h][j] = sum_e  f_e [j + m_e + 1 <= W-1] dy[co][e-kd+1][h][j + m_e + 2 - kw]
//                                + g_e [1 <= j, j + m_e <= W-1] dy[co][e-kd+1][h][j + m_e + 1 - kw]     (dy = 0 outside the tensor)
// from which  dWt[co][c][kd][kh][kw] = sum_{h,j} A[kd][kw][co][h][j] R[c][h+kh-1][j]  and
//             dR[c][y][j] = sum_{co,kd,kw,kh} Wt[co][c][kd][kh][kw] A[kd][kw][co][y-kh+1][j]
// are depth-1 work on 9 C planes (the host runs them on the depth-1 conv / wgrad kernels).  This kernel is the one pass over dy
// (0.74 GB at cfg4): a workgroup owns RB rows of one (n, co), thread = one column j; the rows of dy walk through an LDS ring of
// 16 planes (8 loaded per step, each thread 8 independent 4-byte loads a step ahead of their use), plane e adds into the 9
// register accumulators from planes d = e-1, e, e+1 of the ring: 4 conflict-free LDS reads and 6 FMAs per (e, kd).  Every sum runs
// in ascending e: deterministic, no atomics.  The loading thread also owns column j of every plane, so the depth-class sums of dy
// (the left half's gradient planes, snvc_depth_class_sums) come out of the same pass.
__global__ void __launch_bounds__(1024)
warped_expand_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ shift, float *__restrict__ a,
                         float *__restrict__ dplanes, int C, int D, int H, int W, int RB, int TW) {
    extern __shared__ float lds[];
    constexpr int NS = 16, CH = 8;                       // ring slots, planes per step
    constexpr int kInvalid = (int)0x80000001;
    const int LWB = W + 4;                               // a staged row: columns -2 .. W+1 (zeros outside 0 .. W-1)
    const int tid = threadIdx.x;
    const int r = tid / TW, j = tid - r * TW;
    const int co = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int h = blockIdx.x * RB + r;
    const bool live = h < H && j < W;
    const int64_t hw = (int64_t)H * W;
    float *const ring = lds + (size_t)r * NS * LWB;      // [NS][LWB]
    f32x4 *const ptab = reinterpret_cast<f32x4 *>(lds + (((size_t)RB * NS * LWB + 3) & ~(size_t)3));   // [D]: {m, f, g, -}
    const float *sh = shift + n * D;
    for (int i = tid; i < D; i += blockDim.x) {
        f32x4 t = {__builtin_bit_cast(float, kInvalid), 0.0f, 0.0f, 0.0f};
        const float s = sh[i];
        if (s >= 0.0f && s <= (float)W) {
            const float mf = __builtin_floorf(s);
            float ff = s - mf;
            int mm = (int)mf;
            if (ff == 0.0f) { mm -= 1; ff = 1.0f; }
            t = f32x4{__builtin_bit_cast(float, mm), ff, 1.0f - ff, 0.0f};
        }
        ptab[i] = t;
    }
    // the zero borders of every slot (columns -2, -1, W, W+1) are written once; the loads below only touch 0 .. W-1
    if (j < 4 && r < RB) {
        for (int sl = 0; sl < NS; ++sl) ring[sl * LWB + (j < 2 ? j : W + j)] = 0.0f;
    }
    const float *src = dy + ((n * C + co) * (int64_t)D) * hw + (int64_t)(live ? h : 0) * W + (live ? j : 0);
    float acc[3][3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kd][kw] = 0.0f;
    float c_first = 0.0f, c_mid = 0.0f, c_last = 0.0f;
    float nx[CH];
    // prologue: planes 0 .. CH
    {
        const float v0 = live ? src[0] : 0.0f;
        c_first = v0;
        if (D == 1) c_last = v0;
        if (live) ring[0 * LWB + j + 2] = v0;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int d = 1 + i;
            nx[i] = (live && d < D) ? src[(int64_t)d * hw] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int d = 1 + i;
            if (d < D) {
                if (live) ring[(d & (NS - 1)) * LWB + j + 2] = nx[i];
                if (d == D - 1) c_last = nx[i]; else c_mid += nx[i];
            }
        }
    }
    __syncthreads();
    for (int e0 = 0; e0 < D; e0 += CH) {
        // planes e0 + CH + 1 .. e0 + 2 CH: requested now, stored after this step's arithmetic
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int d = e0 + CH + 1 + i;
            nx[i] = (live && d < D) ? src[(int64_t)d * hw] : 0.0f;
        }
        if (live) {
            const int e1 = e0 + CH < D ? e0 + CH : D;
            for (int e = e0; e < e1; ++e) {
                const f32x4 t = ptab[e];
                const int m = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t[0]));
                if (m == kInvalid) continue;
                const int base = j + m;
                const bool ok = base <= W - 1;
                const float fe = (ok && base + 1 <= W - 1) ? t[1] : 0.0f;
                const float ge = (ok && j >= 1) ? t[2] : 0.0f;
                const int idx = (ok ? base : 0) + 2;
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const int d = e - kd + 1;
                    if (d < 0 || d >= D) continue;
                    const float *row = ring + (d & (NS - 1)) * LWB + idx;
                    const float vm = row[-1], v0 = row[0], v1 = row[1], v2 = row[2];
                    acc[kd][0] = __builtin_fmaf(ge, v1, __builtin_fmaf(fe, v2, acc[kd][0]));
                    acc[kd][1] = __builtin_fmaf(ge, v0, __builtin_fmaf(fe, v1, acc[kd][1]));
                    acc[kd][2] = __builtin_fmaf(ge, vm, __builtin_fmaf(fe, v0, acc[kd][2]));
                }
            }
        }
        __syncthreads();                                 // every wave is done with planes <= e0 + CH - 2 ... the slots written next
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int d = e0 + CH + 1 + i;
            if (d < D) {
                if (live) ring[(d & (NS - 1)) * LWB + j + 2] = nx[i];
                if (d == D - 1) c_last = nx[i]; else c_mid += nx[i];
            }
        }
        __syncthreads();
    }
    if (!live) return;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            a[((((n * 3 + kd) * 3 + kw) * C + co) * (int64_t)H + h) * W + j] = acc[kd][kw];
    if (dplanes) {
        float *dp = dplanes + (((n * C + co) * 3) * (int64_t)H + h) * W + j;
        dp[0] = c_first; dp[hw] = c_mid; dp[2 * hw] = c_last;
    }
}

// ------------------------------------------------------------------------------------ backward (training, cfg4)
// Adjoint of sheared_expand_kernel w.r.t. G and G' (scale = 1: the caller applies the norm's backward first):
//     dG[n][cls][co][h][i]  = sum over (d in class cls, w <= W-2) with Q*w - d - m0 + off  == i of dy[n][co][d][h][w]
//     dG'[n][cls][co][h][i] = sum over (d in class cls)           with Q*(W-1) - d - m0 + off2 == i of dy[n][co][d][h][W-1]
// One workgroup per (n, co, h) row.  Thread t owns the Q slots i = Q*t + p; at a fixed plane d exactly one phase p is
// hit, at column w = t + c(d): consecutive threads read consecutive columns (coalesced), every dy element is read once and
// every sum runs in ascending d: deterministic, no atomics.
template <int Q>
__global__ void __launch_bounds__(256)
sheared_reduce_kernel(const float *__restrict__ dy, float *__restrict__ dg, float *__restrict__ dgcol, int C, int D, int H, int W,
                      int m0, int WG, int off, int WG2, int off2) {
    const int h = blockIdx.x, co = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int64_t plane_sz = (int64_t)H * W;
    const float *row = dy + ((n * C + co) * (int64_t)D) * plane_sz + (int64_t)h * W;
    auto gout = [&](int cls) { return dg + (((n * 3 + cls) * C + co) * (int64_t)H + h) * WG; };
    auto cout_ = [&](int cls) { return dgcol + (((n * 3 + cls) * C + co) * (int64_t)H + h) * WG2; };
    const int nslots = (WG + Q - 1) / Q;
    for (int t = threadIdx.x; t < nslots; t += blockDim.x) {
        float acc[Q], first[Q], lastp[Q];
#pragma unroll
        for (int p = 0; p < Q; ++p) acc[p] = first[p] = lastp[p] = 0.0f;
        // plane d hits phase p(d) = (off - m0 - d) mod Q of this thread, at column w = t + (p + d + m0 - off) / Q
        for (int d = 0; d < D; ++d) {
            const int e = off - m0 - d;
            const int p = ((e % Q) + Q) % Q;
            const int w = t + (p - e) / Q;                      // (p - e) is a multiple of Q
            const float v = ((unsigned)w < (unsigned)(W - 1)) ? row[(int64_t)d * plane_sz + w] : 0.0f;
#pragma unroll
            for (int pp = 0; pp < Q; ++pp) {
                if (pp != p) continue;
                if (d == 0) first[pp] = v;
                else if (d == D - 1) lastp[pp] = v;
                else acc[pp] = acc[pp] + v;
            }
        }
#pragma unroll
        for (int p = 0; p < Q; ++p) {
            const int i = Q * t + p;
            if (i < WG) { gout(0)[i] = first[p]; gout(1)[i] = acc[p]; gout(2)[i] = lastp[p]; }
        }
    }
    // last column: slot i2 = Q*(W-1) - d - m0 + off2 receives dy[d][h][W-1]
    for (int i = threadIdx.x; i < WG2; i += blockDim.x) {
        const int d = Q * (W - 1) - m0 + off2 - i;
        const float v = ((unsigned)d < (unsigned)D) ? row[(int64_t)d * plane_sz + (W - 1)] : 0.0f;
        cout_(0)[i] = d == 0 ? v : 0.0f;
        cout_(1)[i] = (d > 0 && d < D - 1) ? v : 0.0f;
        cout_(2)[i] = (d == D - 1 && d != 0) ? v : 0.0f;
    }
}

// Backward of  y = relu(scale * raw + shift)  through the sheared layer in ONE pass over gy (train-mode BatchNorm, cfg4).
// raw is not stored: it is recomputed from G / G' / planes exactly as sheared_expand_kernel forms it.  The BatchNorm backward
//     draw = A * g + B * raw + Cc      (g = gy where scale*raw + shift > 0; A, B, Cc per channel from sum(g), sum(g*raw))
// is LINEAR in (g, raw, 1), and the layer's own backward only needs sums of draw -- along the shear lines (dG, dG') and over
// the depth classes (the left half's planes) -- so this kernel accumulates those sums of g and of raw separately, plus the
// per-channel fp64 (sum g, sum g*raw); the host combines them with the coefficients on 8 MB instead of 736.
//   line[q][n][cls][co][h][i]   q = 0: g, 1: raw;  sum over (d in cls, w <= W-2) with Q*w - d - m0 + off == i
//   colsum[q][n][co][cls][h][w] sum over d in cls
//   lastc[q][n][cls][co][h][i2] the value at (d, W-1) with Q*(W-1) - d - m0 + off2 == i2
// One wave per image row, lane = 8 consecutive columns, walking d.  A shear line's running sum moves one column to the right
// per plane of its parity chain: it lives in a window of 8 registers that shifts by one per step, the value leaving lane l
// enters lane l+1 (one __shfl_up per step), the one leaving the row's last lane is final.  Every sum has one fixed order.
template <int Q>
__global__ void __launch_bounds__(256)
sheared_bwd_kernel(const float *__restrict__ g, const float *__restrict__ gcol, const float *__restrict__ planes,
                   const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ gy,
                   float *__restrict__ line, float *__restrict__ colsum, float *__restrict__ lastc, double *__restrict__ partial,
                   int N, int C, int D, int H, int W, int m0, int WG, int off, int WG2, int off2) {
    constexpr int RB = 4;                                    // rows (waves) per workgroup
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int h0 = blockIdx.x * RB, h = h0 + wave;
    const int rows = (H - h0) < RB ? (H - h0) : RB;
    const int LW = (WG + Q - 1) / Q + 4;
    float *const phase = lds;                                // [RB][Q][LW]   G (interior class) as phase arrays
    float *const lastg = phase + RB * Q * LW;                // [RB][D]       G' at the last column, per plane
    float *const lineb = lastg + RB * D;                     // [RB][2][WG]   line sums being built
    float *const lcb = lineb + RB * 2 * WG;                  // [RB][2][WG2]  last-column values by slot
    const int64_t CH = (int64_t)C * H;
    const float *gn = g + ((n * 3 + 1) * C + co) * (int64_t)H * WG, *gc = gcol + ((n * 3 + 1) * C + co) * (int64_t)H * WG2;
    for (int e = tid; e < rows * WG; e += 256) {
        const int r = e / WG, i = e - r * WG;
        phase[(r * Q + (i % Q)) * LW + i / Q] = gn[(int64_t)(h0 + r) * WG + i];
    }
    for (int e = tid; e < rows * D; e += 256) {
        const int r = e / D, d = e - r * D;
        const int i = Q * (W - 1) - d - m0 + off2;
        lastg[r * D + d] = (i >= 0 && i < WG2) ? gc[(int64_t)(h0 + r) * WG2 + i] : 0.0f;
    }
    for (int e = tid; e < RB * 2 * WG; e += 256) lineb[e] = 0.0f;
    for (int e = tid; e < RB * 2 * WG2; e += 256) lcb[e] = 0.0f;
    __syncthreads();
    const int nT = W >> 3;                                   // active lanes per row
    const bool act = lane < nT && h < H;
    const bool lastlane = lane == nT - 1;
    const int w0 = 8 * lane;
    const float sc = scale[co], sh = shift[co];
    const int64_t plane_sz = (int64_t)H * W;
    const float *gyp = gy + ((n * C + co) * (int64_t)D) * plane_sz + (int64_t)(act ? h : 0) * W + (act ? w0 : 0);
    const int64_t lrow = (int64_t)H * WG, crow = (int64_t)H * WG2;
    const int64_t line_q = (int64_t)N * 3 * CH * WG, lc_q = (int64_t)N * 3 * CH * WG2, col_q = (int64_t)N * C * 3 * plane_sz;
    double s0 = 0.0, s1 = 0.0;
    auto gload = [&](int i) -> float { return (i >= 0 && i < WG) ? phase[(wave * Q + (i % Q)) * LW + i / Q] : 0.0f; };
    // g and raw of plane d at this lane's 8 columns; rawv in: G part, out: raw
    auto grad8 = [&](const f32x4 a, const f32x4 b, float (&rawv)[8], const float (&pl)[8], float (&gg)[8]) {
        float p0 = 0.0f, p1 = 0.0f;                          // this plane's 8 terms in fp32, then fp64 across planes and lanes
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float x = rawv[k] + pl[k];
            const float v = x * sc + sh;                     // the same two roundings as the forward (fp contract is off)
            const float gyk = k < 4 ? a[k] : b[k - 4];
            gg[k] = v > 0.0f ? gyk : 0.0f;
            rawv[k] = x;
            p0 = p0 + gg[k];
            p1 = p1 + gg[k] * x;
        }
        s0 += (double)p0;
        s1 += (double)p1;
    };
    // ---- the two end planes (depth classes 0 and 2): single terms everywhere
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const int d = e ? D - 1 : 0, cls = e ? 2 : 0;
        if (act) {
            const float *ge = g + ((n * 3 + cls) * C + co) * lrow + (int64_t)h * WG;
            float rawv[8], pl[8], gg[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = Q * (w0 + k) - d - m0 + off;
                rawv[k] = (i >= 0 && i < WG) ? ge[i] : 0.0f;
            }
            if (lastlane) {
                const int i = Q * (W - 1) - d - m0 + off2;
                rawv[7] = (i >= 0 && i < WG2) ? gcol[((n * 3 + cls) * C + co) * crow + (int64_t)h * WG2 + i] : 0.0f;
            }
            const float *pp = planes + (((n * C + co) * 3 + cls) * (int64_t)H + h) * W + w0;
#pragma unroll
            for (int k = 0; k < 8; ++k) pl[k] = pp[k];
            grad8(*reinterpret_cast<const f32x4 *>(gyp + (int64_t)d * plane_sz), *reinterpret_cast<const f32x4 *>(gyp + (int64_t)d * plane_sz + 4),
                  rawv, pl, gg);
            float *cs = colsum + (((n * C + co) * 3 + cls) * (int64_t)H + h) * W + w0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                cs[k] = gg[k];
                cs[col_q + k] = rawv[k];
                const int i = Q * (w0 + k) - d - m0 + off;
                if (lastlane && k == 7) {
                    const int i2 = Q * (W - 1) - d - m0 + off2;
                    if (i2 >= 0 && i2 < WG2) { lcb[(wave * 2 + 0) * WG2 + i2] = gg[k]; lcb[(wave * 2 + 1) * WG2 + i2] = rawv[k]; }
                } else if (i >= 0 && i < WG) {
                    lineb[(wave * 2 + 0) * WG + i] = gg[k];
                    lineb[(wave * 2 + 1) * WG + i] = rawv[k];
                }
            }
        }
        __syncthreads();
        for (int e2 = tid; e2 < rows * 2 * WG; e2 += 256) {       // coalesced rows out, zeros included; then clear
            const int r = e2 / (2 * WG), rem = e2 - r * 2 * WG, qn = rem / WG, i = rem - qn * WG;
            line[qn * line_q + ((n * 3 + cls) * C + co) * lrow + (int64_t)(h0 + r) * WG + i] = lineb[e2];
            lineb[e2] = 0.0f;
        }
        for (int e2 = tid; e2 < rows * 2 * WG2; e2 += 256) {
            const int r = e2 / (2 * WG2), rem = e2 - r * 2 * WG2, qn = rem / WG2, i = rem - qn * WG2;
            lastc[qn * lc_q + ((n * 3 + cls) * C + co) * crow + (int64_t)(h0 + r) * WG2 + i] = lcb[e2];
            lcb[e2] = 0.0f;
        }
        __syncthreads();
    }
    // ---- interior planes d = 1 .. D-2 (depth class 1)
    float pl[8], cg[8], cr[8], wing[Q][8], lg[Q][8], lr[Q][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { pl[k] = 0.0f; cg[k] = 0.0f; cr[k] = 0.0f; }
    if (act) {
        const float *pp = planes + (((n * C + co) * 3 + 1) * (int64_t)H + h) * W + w0;
#pragma unroll
        for (int k = 0; k < 8; ++k) pl[k] = pp[k];
    }
#pragma unroll
    for (int p = 0; p < Q; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            wing[p][k] = gload(Q * (w0 + k) - (1 + p) - m0 + off);
            lg[p][k] = 0.0f;
            lr[p][k] = 0.0f;
        }
    // gy of the NEXT block of 4 planes is requested before this block is processed (~60 KB in flight per CU)
    constexpr int PB = 4;
    static_assert(PB % Q == 0, "a block is whole parity chains");
    f32x4 cur[PB][2], nxt[PB][2];
    auto fetch = [&](int d, f32x4 (&dst)[2]) {
        if (act && d < D - 1) {
            dst[0] = *reinterpret_cast<const f32x4 *>(gyp + (int64_t)d * plane_sz);
            dst[1] = *reinterpret_cast<const f32x4 *>(gyp + (int64_t)d * plane_sz + 4);
        } else {
            dst[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            dst[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
#pragma unroll
    for (int j = 0; j < PB; ++j) fetch(1 + j, cur[j]);
    for (int d0 = 1; d0 < D - 1; d0 += PB) {
#pragma unroll
        for (int j = 0; j < PB; ++j) fetch(d0 + PB + j, nxt[j]);
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int p = j % Q;                              // d0 - 1 is a multiple of PB: the chain of plane d0 + j
            const int d = d0 + j;
            if (d >= D - 1) break;
            float rawv[8], gg[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) rawv[k] = wing[p][k];
            if (lastlane) rawv[7] = lastg[wave * D + d];
            if (act) {
                grad8(cur[j][0], cur[j][1], rawv, pl, gg);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) { gg[k] = 0.0f; rawv[k] = 0.0f; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                cg[k] = cg[k] + gg[k];
                cr[k] = cr[k] + rawv[k];
            }
            if (lastlane) {                                   // column W-1 belongs to G', not to a shear line of G
                const int i2 = Q * (W - 1) - d - m0 + off2;
                if (i2 >= 0 && i2 < WG2) { lcb[(wave * 2 + 0) * WG2 + i2] = gg[7]; lcb[(wave * 2 + 1) * WG2 + i2] = rawv[7]; }
                gg[7] = 0.0f; rawv[7] = 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                lg[p][k] = lg[p][k] + gg[k];
                lr[p][k] = lr[p][k] + rawv[k];
            }
            // the line at k = 7 moves on to the next lane (or is complete); everything shifts by one element
            const float og = lg[p][7], orr = lr[p][7];
            if (lastlane) {
                const int i = Q * (w0 + 7) - d - m0 + off;
                if (i >= 0 && i < WG) { lineb[(wave * 2 + 0) * WG + i] = og; lineb[(wave * 2 + 1) * WG + i] = orr; }
            }
            const float ig = __shfl_up(og, 1, 64), ir = __shfl_up(orr, 1, 64);
#pragma unroll
            for (int k = 7; k > 0; --k) { lg[p][k] = lg[p][k - 1]; lr[p][k] = lr[p][k - 1]; wing[p][k] = wing[p][k - 1]; }
            lg[p][0] = lane == 0 ? 0.0f : ig;
            lr[p][0] = lane == 0 ? 0.0f : ir;
            wing[p][0] = gload(Q * w0 - (d + Q) - m0 + off);
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) { cur[j][0] = nxt[j][0]; cur[j][1] = nxt[j][1]; }
    }
    if (act) {
        // lines still inside the windows: chain p stands at its next plane d = dn(p), element k is slot Q*(w0+k) - dn - m0 + off
#pragma unroll
        for (int p = 0; p < Q; ++p) {
            int dn = 1 + p;
            while (dn < D - 1) dn += Q;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = Q * (w0 + k) - dn - m0 + off;
                if (i >= 0 && i < WG) {
                    lineb[(wave * 2 + 0) * WG + i] = lg[p][k];
                    lineb[(wave * 2 + 1) * WG + i] = lr[p][k];
                }
            }
        }
        float *cs = colsum + (((n * C + co) * 3 + 1) * (int64_t)H + h) * W + w0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { cs[k] = cg[k]; cs[col_q + k] = cr[k]; }
    }
    __syncthreads();
    for (int e2 = tid; e2 < rows * 2 * WG; e2 += 256) {
        const int r = e2 / (2 * WG), rem = e2 - r * 2 * WG, qn = rem / WG, i = rem - qn * WG;
        line[qn * line_q + ((n * 3 + 1) * C + co) * lrow + (int64_t)(h0 + r) * WG + i] = lineb[e2];
    }
    for (int e2 = tid; e2 < rows * 2 * WG2; e2 += 256) {
        const int r = e2 / (2 * WG2), rem = e2 - r * 2 * WG2, qn = rem / WG2, i = rem - qn * WG2;
        lastc[qn * lc_q + ((n * 3 + 1) * C + co) * crow + (int64_t)(h0 + r) * WG2 + i] = lcb[e2];
    }
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o, 64); s1 += __shfl_down(s1, o, 64); }
    __shared__ double red[8];
    if (lane == 0) { red[2 * wave] = s0; red[2 * wave + 1] = s1; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 4; ++k) { s0 += red[2 * k]; s1 += red[2 * k + 1]; }
        double *o = partial + (((int64_t)n * C + co) * gridDim.x + blockIdx.x) * 2;
        o[0] = s0; o[1] = s1;
    }
}

// sums[n*C + c][2] = the workgroups' partials in a fixed order
__global__ void sheared_fold_kernel(const double *__restrict__ partial, double *__restrict__ sums, int64_t rows, int splits) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < splits; ++k) { a += partial[(row * splits + k) * 2]; b += partial[(row * splits + k) * 2 + 1]; }
    sums[row * 2] = a;
    sums[row * 2 + 1] = b;
}

// Weight gradient of the depth-1 3 x 7 convolution  y[co][h][i] = sum_{c,kh,t} K[co][c][kh][t] * x[c][h+kh-1][i+t-3]:
//     dK[co][c][kh][t] = sum_{n,h,i} dy[n][co][h][i] * x[n][c][h+kh-1][i+t-3]
// on the matrix pipe: per tap a 32 x 32 (co x c) tile contracted over the positions (h, i), two positions per
// v_mfma_f32_32x32x2_f32.  A workgroup owns one row h and a range of columns, its four waves split the 21 taps (6,5,5,5
// accumulators), tiles of 64 columns go through LDS (dy [32][65], x [32][3][71]: odd strides, conflict-free column reads).
// Partial sums per workgroup go to a workspace; sheared_wgrad_reduce_kernel adds them in a fixed order (deterministic).
constexpr int SW_IC = 64, SW_XS = SW_IC + 7, SW_YS = SW_IC + 1;
__global__ void __launch_bounds__(256)
sheared_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ ws, int C, int CO, int H, int WU,
                     int col_chunks, int chunk_cols) {
    __shared__ float ylds[32 * SW_YS];
    __shared__ float xlds[32 * 3 * SW_XS];
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x / col_chunks, cc = blockIdx.x - h * col_chunks;
    const int g = blockIdx.y;                      // 32-channel group of dy
    const int64_t n = blockIdx.z;
    const int ntap = wave == 0 ? 6 : 5;            // taps wave, wave + 4, ... of the 21
    f32x16 acc[6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    const float *xn = x + n * (int64_t)C * H * WU;
    const float *yn = dy + (n * CO + g * 32) * (int64_t)H * WU + (int64_t)h * WU;
    const int i_begin = cc * chunk_cols, i_end = (i_begin + chunk_cols) < WU ? (i_begin + chunk_cols) : WU;
    for (int i0 = i_begin; i0 < i_end; i0 += SW_IC) {
        __syncthreads();
        for (int e = tid; e < 32 * SW_IC; e += 256) {
            const int co = e / SW_IC, ii = e - co * SW_IC, i = i0 + ii;
            ylds[co * SW_YS + ii] = (i < i_end && co < CO - g * 32) ? yn[(int64_t)co * H * WU + i] : 0.0f;
        }
        for (int e = tid; e < 32 * 3 * (SW_IC + 6); e += 256) {
            const int c = e / (3 * (SW_IC + 6)), r = e - c * 3 * (SW_IC + 6), kh = r / (SW_IC + 6), ii = r - kh * (SW_IC + 6);
            const int hh = h + kh - 1, i = i0 + ii - 3;
            const bool ok = c < C && (unsigned)hh < (unsigned)H && (unsigned)i < (unsigned)WU;
            xlds[(c * 3 + kh) * SW_XS + ii] = ok ? xn[((int64_t)c * H + hh) * WU + i] : 0.0f;
        }
        __syncthreads();
        const int col = lane & 31, k = lane >> 5;
#pragma unroll 4
        for (int ii = 0; ii < SW_IC; ii += 2) {
            const float af = ylds[col * SW_YS + ii + k];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                if (a < ntap) {
                    const int tap = wave + 4 * a, kh = tap / 7, t = tap - kh * 7;
                    const float bf = xlds[(col * 3 + kh) * SW_XS + ii + k + t];
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[a], 0, 0, 0);
                }
            }
        }
    }
    // partials: ws[((n * groups + g) * nwg + wg)][tap][co_local][c]
    const int nwg = gridDim.x;
    float *wp = ws + ((((n * gridDim.y + g) * nwg + blockIdx.x) * 21) * (int64_t)1024);
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        if (a < ntap) {
            const int tap = wave + 4 * a;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                wp[(tap * 32 + co) * 32 + (lane & 31)] = acc[a][r];
            }
        }
    }
}

// dK[g*32 + co][c][kh][t] = sum over (n, workgroup) of the partials, in that order
__global__ void __launch_bounds__(256)
sheared_wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dk, int C, int CO, int N, int groups, int nwg) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // (g, tap, co, c)
    if (idx >= groups * 21 * 1024) return;
    const int c = idx & 31, co = (idx >> 5) & 31, tap = (idx >> 10) % 21, g = idx / (21 * 1024);
    float s = 0.0f;
    for (int n = 0; n < N; ++n)
        for (int w = 0; w < nwg; ++w) s = s + ws[((((int64_t)n * groups + g) * nwg + w) * 21 + tap) * 1024 + co * 32 + c];
    if (g * 32 + co < CO && c < C) dk[((int64_t)(g * 32 + co) * C + c) * 21 + tap] = s;
}

// Adjoint of sheared_upsample_kernel: dR[j] = sum_u dRq[u] * dRq[u]/dR[j]   (u = Q*j exactly; u = 2j +- 1 with weight 1/2)
__global__ void __launch_bounds__(256)
sheared_upsample_bwd_kernel(const float *__restrict__ drq, float *__restrict__ dr, int W, int q, int WU, int off, int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * W) return;
    const int64_t row = idx / W;
    const int j = (int)(idx - row * W);
    const float *g = drq + row * WU;
    auto at = [&](int u) { const int i = u + off; return (u >= 0 && u <= q * (W - 1) && i >= 0 && i < WU) ? g[i] : 0.0f; };
    float v = at(q * j);
    if (q == 2) v = v + (0.5f * at(2 * j - 1) + 0.5f * at(2 * j + 1));
    dr[idx] = v;
}

// ------------------------------------------------------------------------------------ split-mode output (r4, "f16x3")
// sheared_expand_kernel with the result written as a split C8 pair (conv3d_f16.hip, F16Cfg::PL: v = hi + lo, two half planes
// [C/8][D][H][W][8]) for a consumer on the split-mode kernels: the SAME bytes as the fp32 tensor.  A C8 piece holds the 8 channels
// of one voxel, so here a THREAD is one voxel column of one image row and owns its 8 channels: per plane it reads the 8 values
// G[c][h][q*w - d - m0] from LDS (the index drops by one per plane), forms the 8 results and stores ONE 16-byte piece to each
// plane of the pair -- consecutive lanes = consecutive voxels, 1 KB contiguous per store instruction.  (First form: a thread = 4
// columns x 8 channels with register windows, four pieces per store at a 64-byte lane stride: 0.345 ms; this form: see DESIGN.)
// A workgroup = one row x one channel group x a CHUNK of the depth range (blockIdx.y = group * DCH + chunk), so that the launch
// has >= 1024 workgroups.  scale / bias carry the tensor's exponent (v * 2^e, folded by the host: exact).  Values beyond half's
// range are clamped and flagged (`overflow`).
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

template <int Q>
__global__ void __launch_bounds__(512)
sheared_expand_split_kernel(const float *__restrict__ g, const float *__restrict__ gcol, const float *__restrict__ planes,
                            const float *__restrict__ scale, const float *__restrict__ bias, _Float16 *__restrict__ yh,
                            _Float16 *__restrict__ yl, int *__restrict__ overflow, int C, int D, int H, int W, int m0, int WG, int off,
                            int WG2, int off2, int DCH, int DC, int64_t y_bs, int flags) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int cg = blockIdx.y / DCH, dch = blockIdx.y - cg * DCH;
    const int64_t n = blockIdx.z;
    const int h = blockIdx.x;
    const int d_lo = dch * DC, d_hi = (d_lo + DC) < D ? (d_lo + DC) : D;
    const int LW = (WG + Q - 1) / Q + 4;
    float *const phase = lds;                            // [8][Q][LW]: G (interior class) of this row, even / odd elements apart
    float *const lastcol = lds + 8 * Q * LW;             // [8][DC]: G' at the last column, per plane of this chunk
    for (int e = tid; e < 8 * WG; e += blockDim.x) {
        const int c = e / WG, i = e - c * WG;
        const int co = cg * 8 + c;
        phase[(c * Q + (i % Q)) * LW + i / Q] = co < C ? g[(((n * 3 + 1) * C + co) * (int64_t)H + h) * WG + i] : 0.0f;
    }
    for (int e = tid; e < 8 * DC; e += blockDim.x) {
        const int c = e / DC, dd = e - c * DC;
        const int co = cg * 8 + c, d = d_lo + dd;
        const int i = Q * (W - 1) - d - m0 + off2;
        lastcol[c * DC + dd] = (co < C && d < D && i >= 0 && i < WG2) ? gcol[(((n * 3 + 1) * C + co) * (int64_t)H + h) * WG2 + i] : 0.0f;
    }
    __syncthreads();
    const int w = tid;
    if (w >= W) return;
    const bool relu = (flags & SNVC_EPI_RELU) != 0, last = w == W - 1, nt = (flags & SNVC_EPI_STREAM_OUT) != 0;
    const int64_t plane_sz = (int64_t)H * W;
    _Float16 *yhp = yh + n * y_bs + (((int64_t)cg * D) * plane_sz + (int64_t)h * W + w) * 8;
    _Float16 *ylp = yl + n * y_bs + (((int64_t)cg * D) * plane_sz + (int64_t)h * W + w) * 8;
    float sc[8], bi[8], pl[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int co = cg * 8 + c;
        sc[c] = co < C ? (scale ? scale[co] : 1.0f) : 0.0f;
        bi[c] = (scale && co < C) ? bias[co] : 0.0f;
        pl[c] = (planes && co < C) ? planes[(((n * C + co) * 3 + 1) * (int64_t)H + h) * W + w] : 0.0f;      // interior class
    }
    // per element: one add, one FMA, ONE v_med3 (ReLU and the clamp to half's range together), one running max for the overflow
    // flag (compared once at the end), three conversions and a subtraction -- r4's form (max, min + max, a compare chain per
    // element) spent ~13 VALU instructions where this spends 8; the pass is co-limited by them (54 % of HBM in the step)
    float vmax = 0.0f;
    const float lo_bound = relu ? 0.0f : -65504.0f;
    auto emit = [&](int d, const float (&raw)[8], const float (&pe)[8]) {
        h8v hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float t = __builtin_fmaf(raw[c] + pe[c], sc[c], bi[c]);
            const float tc = __builtin_amdgcn_fmed3f(t, lo_bound, 65504.0f);
            vmax = __builtin_fmaxf(vmax, __builtin_fabsf(tc));      // what is stored (|x| is a source modifier); a NaN never raises the flag
            hi[c] = (_Float16)tc;
            lo[c] = (_Float16)(tc - (float)hi[c]);
        }
        if (nt) {       // block-uniform: a streamed-out tensor that is read once, much later than the L2 keeps it
            __builtin_nontemporal_store(hi, reinterpret_cast<h8v *>(yhp + (int64_t)d * plane_sz * 8));
            __builtin_nontemporal_store(lo, reinterpret_cast<h8v *>(ylp + (int64_t)d * plane_sz * 8));
        } else {
            *reinterpret_cast<h8v *>(yhp + (int64_t)d * plane_sz * 8) = hi;
            *reinterpret_cast<h8v *>(ylp + (int64_t)d * plane_sz * 8) = lo;
        }
    };
    // the two end planes (depth classes 0 and 2): their own G / G' and planes, read straight from L2 by the chunk that holds them
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int d = e ? D - 1 : 0, cls = e ? 2 : 0;
        if (d < d_lo || d >= d_hi) continue;
        float raw[8], pe[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int co = cg * 8 + c;
            const int i = Q * w - d - m0 + off, i2 = Q * (W - 1) - d - m0 + off2;
            raw[c] = 0.0f;
            pe[c] = 0.0f;
            if (co < C) {
                if (last) raw[c] = (i2 >= 0 && i2 < WG2) ? gcol[(((n * 3 + cls) * C + co) * (int64_t)H + h) * WG2 + i2] : 0.0f;
                else raw[c] = (i >= 0 && i < WG) ? g[(((n * 3 + cls) * C + co) * (int64_t)H + h) * WG + i] : 0.0f;
                if (planes) pe[c] = planes[(((n * C + co) * 3 + cls) * (int64_t)H + h) * W + w];
            }
        }
        emit(d, raw, pe);
    }
    const int dstart = d_lo < 1 ? 1 : d_lo, dend = d_hi < D - 1 ? d_hi : D - 1;
    for (int d = dstart; d < dend; ++d) {
        const int i = Q * w - d - m0 + off;                  // drops by one per plane: alternates between the Q phase arrays
        const bool in = i >= 0 && i < WG;
        const int ph = in ? (i % Q) * LW + i / Q : 0;
        float raw[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = phase[c * Q * LW + ph];
            raw[c] = last ? lastcol[c * DC + (d - d_lo)] : (in ? v : 0.0f);
        }
        emit(d, raw, pl);
    }
    if (vmax >= 65504.0f && overflow) atomicOr(overflow, 1);
}

// warped_expand_win_kernel with the result written as a split C8 pair (see sheared_expand_split_kernel): a thread = one voxel
// column of one image row, 8 channels; a workgroup = one row x one channel group x a chunk of the depth range.  Per plane and kd
// the thread needs F_kd[c][w - m - 1] and G_kd[c][w - m] of its 8 channels: 16 LDS reads, kept in registers while m stands (every
// other plane on half-pixel steps).  Same plane parameters (ptab), same whole-pixel remap, same last-column table as the fp32 form.
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(5)))      // <= 96 VGPRs: four 5-wave workgroups per CU
warped_expand_split_kernel(const float *__restrict__ p, const float *__restrict__ q, const float *__restrict__ e,
                           const float *__restrict__ planes, const float *__restrict__ shift, const float *__restrict__ scale,
                           const float *__restrict__ bias, _Float16 *__restrict__ yh, _Float16 *__restrict__ yl, int *__restrict__ overflow,
                           int C, int D, int H, int W, int DCH, int DC, int64_t y_bs, int flags) {
    extern __shared__ float lds[];
    constexpr int FP = 12;
    constexpr int kInvalid = (int)0x80000001, kNone = (int)0x80000000;
    const int LW = W + 16;
    const int tid = threadIdx.x;
    const int cg = blockIdx.y / DCH, dch = blockIdx.y - cg * DCH;
    const int64_t n = blockIdx.z;
    const int h = blockIdx.x;
    const int d_lo = dch * DC, d_hi = (d_lo + DC) < D ? (d_lo + DC) : D;
    const int64_t hw = (int64_t)H * W;
    // Only the F rows are staged (31.5 KB instead of 63: four workgroups per CU): the G rows differ from them at columns -1, 0, 1
    // alone (G = F - {E2, E1, E0}: the gated sample's share), which the lanes that sit there subtract when their window moves.
    float *const rowsF = lds;                            // [3 kd][8][LW]
    float *const corr = lds + 24 * LW;                   // [3 kd][8][4]: {E2, E1, E0, 0}
    float *const tas = corr + 24 * 4;                    // [8][DC]
    f32x4 *const ptab = reinterpret_cast<f32x4 *>(tas + 8 * DC + ((4 - ((8 * DC) & 3)) & 3));       // [D + 2]
    const float *sh = shift + n * D;
    for (int i = tid; i < D + 2; i += blockDim.x) {
        f32x4 t = {__builtin_bit_cast(float, kInvalid), 0.0f, 0.0f, 0.0f};
        if (i < D) {
            const float s = sh[i];
            if (s <= (float)W) {
                const float mf = __builtin_floorf(s);
                float ff = s - mf;
                int mm = (int)mf;
                if (ff == 0.0f) { mm -= 1; ff = 1.0f; }      // whole pixel: F[w - m], nothing gated
                t = f32x4{__builtin_bit_cast(float, mm), ff, 1.0f - ff, 0.0f};
            }
        }
        ptab[i] = t;
    }
    for (int i = tid; i < 24 * (LW >> 2); i += blockDim.x) {
        const int row = i / (LW >> 2), pc = i - row * (LW >> 2);      // row = kd * 8 + c
        const int kd = row >> 3, c = row & 7, co = cg * 8 + c, col = 4 * pc - FP;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (co < C) {
            if (col >= 0 && col < W) v = *reinterpret_cast<const f32x4 *>(p + (((n * 3 + kd) * C + co) * (int64_t)H + h) * W + col);
            const float *er = e + (((n * 3 + kd) * 3) * C + co) * (int64_t)H * 4 + (int64_t)h * 4;
            const int64_t kws = (int64_t)C * H * 4;
            if (col == -4) {                                                    // column -1: F = E2
                v[3] = er[2 * kws];
                *reinterpret_cast<f32x4 *>(corr + row * 4) = f32x4{er[2 * kws], er[kws], er[0], 0.0f};
            }
        } else if (col == -4) {
            *reinterpret_cast<f32x4 *>(corr + row * 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        *reinterpret_cast<f32x4 *>(rowsF + row * LW + 4 * pc) = v;
    }
    __syncthreads();
    for (int i = tid; i < 8 * DC; i += blockDim.x) {
        const int c = i / DC, dd = i - c * DC, d = d_lo + dd, co = cg * 8 + c;
        float t = 0.0f;
        if (co < C && d < D) {
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const int ee = d + kd - 1;
                const f32x4 pt = ptab[ee < 0 ? D : ee];
                const int mm = __builtin_bit_cast(int, pt[0]);
                if (mm == kInvalid) continue;
                const int i0 = W - mm - 1;
                const float *qr = q + (((n * 3 + kd) * C + co) * (int64_t)H + h) * W;
                const float a0 = (unsigned)i0 < (unsigned)W ? qr[i0] : 0.0f;
                float a1 = (unsigned)(i0 + 1) < (unsigned)W ? qr[i0 + 1] : 0.0f;
                if (i0 + 1 == 0) a1 = 0.0f;
                t += pt[1] * a0 + pt[2] * a1;
            }
        }
        tas[c * DC + dd] = t;
    }
    __syncthreads();
    const int w = tid;
    if (w >= W) return;
    const bool last = w == W - 1, relu = (flags & SNVC_EPI_RELU) != 0, nt = (flags & SNVC_EPI_STREAM_OUT) != 0;
    float sc[8], bi[8], pl[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int co = cg * 8 + c;
        sc[c] = co < C ? (scale ? scale[co] : 1.0f) : 0.0f;
        bi[c] = (scale && co < C) ? bias[co] : 0.0f;
        pl[c] = (planes && co < C) ? planes[(((n * C + co) * 3 + 1) * (int64_t)H + h) * W + w] : 0.0f;      // interior class; the two end
    }                                                                                                       // planes fetch their own
    _Float16 *yhp = yh + n * y_bs + (((int64_t)cg * D) * hw + (int64_t)h * W + w) * 8;
    _Float16 *ylp = yl + n * y_bs + (((int64_t)cg * D) * hw + (int64_t)h * W + w) * 8;
    auto uni_i = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni_f = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    int pm[3];
    float pf[3], pg[3];
    auto take = [&](const f32x4 &t, int &m, float &f, float &g) {
        m = uni_i(__builtin_bit_cast(int, t[0])); f = uni_f(t[1]); g = uni_f(t[2]);
    };
    take(ptab[d_lo - 1 < 0 ? D : d_lo - 1], pm[0], pf[0], pg[0]);
    take(ptab[d_lo], pm[1], pf[1], pg[1]);
    take(ptab[d_lo + 1], pm[2], pf[2], pg[2]);            // ptab has D + 2 entries, the last two invalid
    f32x4 nxt = ptab[d_lo + 2 < D + 2 ? d_lo + 2 : D + 1];
    // window kd of channel c = the PAIR {F_kd[c][w - m - 1], G_kd[c][w - m]}: two adjacent floats of the staged row (one ds_read2_b32,
    // landing in the register pair the packed FMA reads), G's correction subtracted from the second by the <= 3 lanes it concerns.
    // Per plane and channel: three v_pk_fma_f32 {f, g} * {F, G} into a pair of partial sums (the f-terms | the g-terms) and one add
    // (r4's form: six v_fmac and the window in two separate registers, 16 ds_read_b32 per move)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 win[3][8];
    int mw[3] = {kNone, kNone, kNone};
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int c = 0; c < 8; ++c) win[kd][c] = f32x2{0.0f, 0.0f};
    float vmax = 0.0f;
    const float lo_bound = relu ? 0.0f : -65504.0f;
    for (int d = d_lo; d < d_hi; ++d) {
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            if (pm[kd] != mw[kd] && pm[kd] != kInvalid) {      // wave-uniform: the window of this kd moves
                const int idx = FP + w - pm[kd] - 1;             // column w - m - 1 of the staged rows; left of the pad: zeros
                const int fi = idx < 0 ? 0 : idx;                // (columns 0 .. FP - 2 of a staged row are zeros: fi = 0, fi + 1 = 1 read two of them)
                const int gcol = w - pm[kd];                     // the G-term's column: -1, 0, 1 carry the gate's correction
                const float *fr = rowsF + kd * 8 * LW + fi;
#pragma unroll
                for (int c = 0; c < 8; ++c) win[kd][c] = f32x2{fr[c * LW], fr[c * LW + 1]};
                if ((unsigned)(gcol + 1) <= 2u) {                // at most three lanes of the row
#pragma unroll
                    for (int c = 0; c < 8; ++c) win[kd][c][1] -= corr[(kd * 8 + c) * 4 + gcol + 1];
                }
                mw[kd] = pm[kd];
            }
        }
        int nm;
        float nff, ngg;
        take(nxt, nm, nff, ngg);
        nxt = ptab[d + 3 < D + 2 ? d + 3 : D + 1];
        const int cls = d == 0 ? 0 : (d == D - 1 ? 2 : 1);
        const f32x2 fg0 = {pf[0], pg[0]}, fg1 = {pf[1], pg[1]}, fg2 = {pf[2], pg[2]};
        float pe[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) pe[c] = pl[c];
        if (cls != 1) {                      // block-uniform, two planes of the whole walk
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int co = cg * 8 + c;
                pe[c] = (planes && co < C) ? planes[(((n * C + co) * 3 + cls) * (int64_t)H + h) * W + w] : 0.0f;
            }
        }
        h8v hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float base = pe[c];
            f32x2 a2 = __builtin_elementwise_fma(fg0, win[0][c], f32x2{base, 0.0f});
            a2 = __builtin_elementwise_fma(fg1, win[1][c], a2);
            a2 = __builtin_elementwise_fma(fg2, win[2][c], a2);
            float o = a2[0] + a2[1];
            if (last) o -= tas[c * DC + (d - d_lo)];
            float t = __builtin_fmaf(o, sc[c], bi[c]);
            vmax = __builtin_fmaxf(vmax, __builtin_fabsf(t));
            t = __builtin_amdgcn_fmed3f(t, lo_bound, 65504.0f);      // ReLU and the clamp to half's range in one
            hi[c] = (_Float16)t;
            lo[c] = (_Float16)(t - (float)hi[c]);
        }
        if (nt) {
            __builtin_nontemporal_store(hi, reinterpret_cast<h8v *>(yhp + (int64_t)d * hw * 8));
            __builtin_nontemporal_store(lo, reinterpret_cast<h8v *>(ylp + (int64_t)d * hw * 8));
        } else {
            *reinterpret_cast<h8v *>(yhp + (int64_t)d * hw * 8) = hi;
            *reinterpret_cast<h8v *>(ylp + (int64_t)d * hw * 8) = lo;
        }
        pm[0] = pm[1]; pf[0] = pf[1]; pg[0] = pg[1];
        pm[1] = pm[2]; pf[1] = pf[2]; pg[1] = pg[2];
        pm[2] = nm; pf[2] = nff; pg[2] = ngg;
    }
    if (vmax >= 65504.0f && overflow) atomicOr(overflow, 1);
}

}  // namespace
}  // namespace snvc

extern "C" {

int snvc_sheared_upsample_split(const float *right, void *y_hi, void *y_lo, const float *mul_dev, int64_t N, int64_t C, int64_t H, int64_t W,
                                int q, int64_t WU, int off, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || (q != 1 && q != 2) || WU <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_upsample_split: bad sizes (q in {1,2})");
    if (N == 0) return SNVC_OK;
    if (!right || !y_hi || !y_lo || !mul_dev) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_upsample_split: null pointer");
    if ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_upsample_split: y must be 16-byte aligned");
    const int64_t total = N * ((C + 7) / 8) * H * WU;
    if (ceil_div<int64_t>(total, 256) >= ((int64_t)1 << 31)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_upsample_split: too large");
    sheared_upsample_split_kernel<<<(unsigned)ceil_div<int64_t>(total, 256), 256, 0, as_stream(stream)>>>(
        right, reinterpret_cast<_Float16 *>(y_hi), reinterpret_cast<_Float16 *>(y_lo), mul_dev, (int)C, (int)H, (int)W, q, (int)WU, off, total);
    return check_launch("snvc_sheared_upsample_split");
}

int snvc_sheared_upsample(const float *right, float *out, int64_t N, int64_t C, int64_t H, int64_t W, int q, int64_t WU,
                          int off, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || (q != 1 && q != 2) || WU <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_upsample: bad sizes (q in {1,2})");
    if (N == 0) return SNVC_OK;
    if (!right || !out) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_upsample: null pointer");
    const int64_t rows = N * C * H, total = rows * WU;
    if (ceil_div<int64_t>(total, 256) >= ((int64_t)1 << 31)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_upsample: too large");
    sheared_upsample_kernel<<<(unsigned)ceil_div<int64_t>(total, 256), 256, 0, as_stream(stream)>>>(right, out, (int)W, q, (int)WU,
                                                                                                     off, rows);
    return check_launch("snvc_sheared_upsample");
}

// Launch geometry of sheared_expand_kernel, shared by the forward statistics pass
static int sheared_expand_rows(int64_t N, int64_t C, int64_t H, int quads, size_t lds_floats_per_row) {
    int RB = 512 / quads;                       // rows per workgroup: as many as 512 threads cover ...
    if (RB > 8) RB = 8;
    while (RB > 1 && snvc::ceil_div<int64_t>(H, RB) * C * N < 4 * 256) RB = (RB + 1) / 2;      // ... while the chip stays covered
    while (RB > 1 && sizeof(float) * RB * lds_floats_per_row > 150 * 1024) --RB;              // ... and the rows fit the LDS
    return RB;
}

int snvc_sheared_expand(const float *g, const float *gcol, const float *planes, const float *scale, const float *bias, float *y,
                        int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int q, int m0, int64_t WG, int off, int64_t WG2,
                        int off2, int flags, void *stream) {
    return snvc_sheared_expand_amax(g, gcol, planes, scale, bias, y, N, C, D, H, W, q, m0, WG, off, WG2, off2, flags, nullptr, stream);
}

int snvc_sheared_expand_amax(const float *g, const float *gcol, const float *planes, const float *scale, const float *bias, float *y,
                             int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int q, int m0, int64_t WG, int off, int64_t WG2,
                             int off2, int flags, uint32_t *amax, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || D < 2 || H <= 0 || W <= 0 || W % 4 != 0 || (q != 1 && q != 2 && q != 4) || m0 < 0 || WG <= 0 || WG2 <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand: bad sizes (W % 4 == 0, q in {1,2,4}, D >= 2)");
    if ((scale == nullptr) != (bias == nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand: scale and bias must both be given or both be NULL");
    if (flags & ~SNVC_EPI_RELU) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand: only SNVC_EPI_RELU");
    if (N == 0) return SNVC_OK;
    if (!g || !gcol || !y) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand: null pointer");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(planes)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand: y and planes must be 16-byte aligned");
    const int quads = (int)(W / 4);
    if (quads > 512 || C > 65535 || N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand: row too wide or too many channels");
    const int LW = (int)((WG + q - 1) / q) + 4;
    const int RB = sheared_expand_rows(N, C, H, quads, (size_t)q * LW + (size_t)D);
    const int threads = ceil_div(RB * quads, 64) * 64;
    const size_t lds = sizeof(float) * ((size_t)RB * q * LW + (size_t)RB * D);
    if (lds > 150 * 1024) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand: rows do not fit the LDS");
    const dim3 grid((unsigned)ceil_div<int64_t>(H, RB), (unsigned)C, (unsigned)N);
    static std::atomic<unsigned> attr1{0}, attr2{0}, attr4{0};
    if (q == 4) {      // r6: four phases (downsample 2 with half-pixel planes, downsample 4 with whole-pixel planes: index 4 w - m0 - d)
        static std::atomic<unsigned> attr_m2_4{0};
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_kernel<4, 0>), (int)lds, attr4) ||
            (amax && !allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_kernel<4, 2>), (int)lds, attr_m2_4)))
            return check_launch("snvc_sheared_expand");
        if (amax)
            sheared_expand_kernel<4, 2><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, y, reinterpret_cast<double *>(amax),
                                                                                 (int)C, (int)D, (int)H, (int)W, m0, (int)WG, off, (int)WG2, off2, RB, flags);
        else
            sheared_expand_kernel<4, 0><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, y, nullptr, (int)C, (int)D, (int)H,
                                                                                 (int)W, m0, (int)WG, off, (int)WG2, off2, RB, flags);
    } else if (q == 1) {
        static std::atomic<unsigned> attr_m2_1{0};
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_kernel<1, 0>), (int)lds, attr1) ||
            (amax && !allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_kernel<1, 2>), (int)lds, attr_m2_1)))
            return check_launch("snvc_sheared_expand");
        if (amax)
            sheared_expand_kernel<1, 2><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, y, reinterpret_cast<double *>(amax),
                                                                                 (int)C, (int)D, (int)H, (int)W, m0, (int)WG, off, (int)WG2, off2, RB, flags);
        else
            sheared_expand_kernel<1, 0><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, y, nullptr, (int)C, (int)D, (int)H,
                                                                                 (int)W, m0, (int)WG, off, (int)WG2, off2, RB, flags);
    } else {
        static std::atomic<unsigned> attr_m2_2{0};
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_kernel<2, 0>), (int)lds, attr2) ||
            (amax && !allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_kernel<2, 2>), (int)lds, attr_m2_2)))
            return check_launch("snvc_sheared_expand");
        if (amax)
            sheared_expand_kernel<2, 2><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, y, reinterpret_cast<double *>(amax),
                                                                                 (int)C, (int)D, (int)H, (int)W, m0, (int)WG, off, (int)WG2, off2, RB, flags);
        else
            sheared_expand_kernel<2, 0><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, y, nullptr, (int)C, (int)D, (int)H,
                                                                                 (int)W, m0, (int)WG, off, (int)WG2, off2, RB, flags);
    }
    return check_launch("snvc_sheared_expand");
}

int snvc_sheared_expand_split(const float *g, const float *gcol, const float *planes, const float *scale, const float *bias, void *y_hi,
                              void *y_lo, int *overflow, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int q, int m0, int64_t WG,
                              int off, int64_t WG2, int off2, int64_t y_batch_stride, int flags, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || D < 2 || H <= 0 || W <= 0 || (q != 1 && q != 2 && q != 4) || m0 < 0 || WG <= 0 || WG2 <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand_split: bad sizes (q in {1,2,4}, D >= 2)");
    if ((scale == nullptr) != (bias == nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand_split: scale and bias must both be given or both be NULL");
    if (flags & ~(SNVC_EPI_RELU | SNVC_EPI_STREAM_OUT)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand_split: only SNVC_EPI_RELU | SNVC_EPI_STREAM_OUT");
    if (N == 0) return SNVC_OK;
    if (!g || !gcol || !y_hi || !y_lo) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand_split: null pointer");
    if ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand_split: y must be 16-byte aligned");
    const int64_t G = ceil_div<int64_t>(C, 8);
    if (W > 512 || G > 4095 || N > 65535 || H >= ((int64_t)1 << 31))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand_split: row too wide (W <= 512) or too many channels");
    const int LW = (int)((WG + q - 1) / q) + 4;
    // depth chunks: at least two workgroups per CU, at least 8 planes each -- and no more: all of the launch's workgroups resident at
    // once (five fit a CU) and long walks measure best (r5, cfg2's 96 rows x 4 groups: 1 chunk 166 us, 2: 156, 4: 169, 8: 171, 16: 180)
    int DCH = 1;
    while (DCH < 16 && H * G * N * DCH < 2 * 256 && D / (2 * DCH) >= 8) DCH *= 2;
    const int DC = (int)ceil_div<int64_t>(D, DCH);
    const size_t lds = sizeof(float) * (8 * (size_t)q * LW + 8 * (size_t)DC);
    if (lds > 150 * 1024) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand_split: the row does not fit the LDS");
    const int threads = ceil_div((int)W, 64) * 64;
    const dim3 grid((unsigned)H, (unsigned)(G * DCH), (unsigned)N);
    const int64_t y_bs = y_batch_stride ? y_batch_stride : 2 * G * 8 * D * H * W;
    static std::atomic<unsigned> attr1{0}, attr2{0}, attr4{0};
    _Float16 *yh = reinterpret_cast<_Float16 *>(y_hi), *yl = reinterpret_cast<_Float16 *>(y_lo);
    if (q == 4) {
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_split_kernel<4>), (int)lds, attr4)) return check_launch("snvc_sheared_expand_split");
        sheared_expand_split_kernel<4><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, yh, yl, overflow, (int)C, (int)D, (int)H,
                                                                              (int)W, m0, (int)WG, off, (int)WG2, off2, DCH, DC, y_bs, flags);
    } else if (q == 1) {
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_split_kernel<1>), (int)lds, attr1)) return check_launch("snvc_sheared_expand_split");
        sheared_expand_split_kernel<1><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, yh, yl, overflow, (int)C, (int)D, (int)H,
                                                                              (int)W, m0, (int)WG, off, (int)WG2, off2, DCH, DC, y_bs, flags);
    } else {
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_split_kernel<2>), (int)lds, attr2)) return check_launch("snvc_sheared_expand_split");
        sheared_expand_split_kernel<2><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, yh, yl, overflow, (int)C, (int)D, (int)H,
                                                                              (int)W, m0, (int)WG, off, (int)WG2, off2, DCH, DC, y_bs, flags);
    }
    return check_launch("snvc_sheared_expand_split");
}

int64_t snvc_sheared_stats_workspace_bytes(int64_t N, int64_t C, int64_t H) {
    if (N < 0 || C <= 0 || H <= 0) return -1;
    return (N * C * H * 2 + 2 * N * C) * (int64_t)sizeof(double) + 16;     // one partial pair per (n, c, row block <= H)
}

int snvc_sheared_expand_stats(const float *g, const float *gcol, const float *planes, const float *gamma, const float *beta,
                              float *scale, float *shift, float *mean, float *var, void *workspace, int64_t N, int64_t C, int64_t D,
                              int64_t H, int64_t W, int q, int m0, int64_t WG, int off, int64_t WG2, int off2, float eps,
                              void *stream) {
    using namespace snvc;
    if (N <= 0 || C <= 0 || D < 2 || H <= 0 || W <= 0 || W % 4 != 0 || (q != 1 && q != 2) || m0 < 0 || WG <= 0 || WG2 <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand_stats: bad sizes (W % 4 == 0, q in {1,2}, D >= 2)");
    if (!g || !gcol || !scale || !shift || !workspace) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand_stats: null pointer");
    if (reinterpret_cast<uintptr_t>(planes) & 15) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand_stats: planes must be 16-byte aligned");
    const int quads = (int)(W / 4);
    if (quads > 512 || C > 65535 || N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand_stats: row too wide or too many channels");
    const int LW = (int)((WG + q - 1) / q) + 4;
    const int RB = sheared_expand_rows(N, C, H, quads, (size_t)q * LW + (size_t)D);
    const int threads = ceil_div(RB * quads, 64) * 64;
    const size_t lds = sizeof(float) * ((size_t)RB * q * LW + (size_t)RB * D);
    if (lds > 150 * 1024) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand_stats: rows do not fit the LDS");
    const dim3 grid((unsigned)ceil_div<int64_t>(H, RB), (unsigned)C, (unsigned)N);
    double *partial = static_cast<double *>(workspace);
    static std::atomic<unsigned> attr1{0}, attr2{0};
    if (q == 1) {
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_kernel<1, 1>), (int)lds, attr1)) return check_launch("snvc_sheared_expand_stats");
        sheared_expand_kernel<1, 1><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, nullptr, nullptr, nullptr, partial, (int)C, (int)D,
                                                                             (int)H, (int)W, m0, (int)WG, off, (int)WG2, off2, RB, 0);
    } else {
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_kernel<2, 1>), (int)lds, attr2)) return check_launch("snvc_sheared_expand_stats");
        sheared_expand_kernel<2, 1><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, nullptr, nullptr, nullptr, partial, (int)C, (int)D,
                                                                             (int)H, (int)W, m0, (int)WG, off, (int)WG2, off2, RB, 0);
    }
    int rc = check_launch("snvc_sheared_expand_stats(partial)");
    if (rc) return rc;
    launch_norm_finalize(partial, gamma, beta, scale, shift, mean, var, N, C, D * H * W, (int)grid.x, eps, as_stream(stream));
    return check_launch("snvc_sheared_expand_stats(finalize)");
}

int64_t snvc_sheared_backward_workspace_bytes(int64_t N, int64_t C, int64_t H) {
    if (N < 0 || C <= 0 || H <= 0) return -1;
    return N * C * ((H + 3) / 4) * 2 * (int64_t)sizeof(double) + 16;
}

int snvc_sheared_backward_reduce(const float *g, const float *gcol, const float *planes, const float *scale, const float *shift,
                                 const float *gy, float *line, float *colsum, float *lastc, double *sums, void *workspace, int64_t N,
                                 int64_t C, int64_t D, int64_t H, int64_t W, int q, int m0, int64_t WG, int off, int64_t WG2, int off2,
                                 void *stream) {
    using namespace snvc;
    if (N <= 0 || C <= 0 || D < 2 || H <= 0 || W <= 0 || W % 8 != 0 || W > 512 || (q != 1 && q != 2) || m0 < 0 || WG <= 0 || WG2 <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_backward_reduce: bad sizes (W % 8 == 0, W <= 512, q in {1,2}, D >= 2)");
    if (!g || !gcol || !planes || !scale || !shift || !gy || !line || !colsum || !lastc || !sums || !workspace)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_backward_reduce: null pointer");
    if (reinterpret_cast<uintptr_t>(gy) & 15) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_backward_reduce: gy must be 16-byte aligned");
    if (C > 65535 || N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_backward_reduce: too many channels or samples");
    const int LW = (int)((WG + q - 1) / q) + 4;
    const size_t lds = sizeof(float) * (4 * ((size_t)q * LW + D + 2 * WG + 2 * WG2));
    if (lds > 150 * 1024) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_backward_reduce: rows do not fit the LDS");
    const dim3 grid((unsigned)ceil_div<int64_t>(H, 4), (unsigned)C, (unsigned)N);
    double *partial = static_cast<double *>(workspace);
    static std::atomic<unsigned> attr1{0}, attr2{0};
    if (q == 1) {
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_bwd_kernel<1>), (int)lds, attr1)) return check_launch("snvc_sheared_backward_reduce");
        sheared_bwd_kernel<1><<<grid, 256, lds, as_stream(stream)>>>(g, gcol, planes, scale, shift, gy, line, colsum, lastc, partial, (int)N, (int)C,
                                                                   (int)D, (int)H, (int)W, m0, (int)WG, off, (int)WG2, off2);
    } else {
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_bwd_kernel<2>), (int)lds, attr2)) return check_launch("snvc_sheared_backward_reduce");
        sheared_bwd_kernel<2><<<grid, 256, lds, as_stream(stream)>>>(g, gcol, planes, scale, shift, gy, line, colsum, lastc, partial, (int)N, (int)C,
                                                                   (int)D, (int)H, (int)W, m0, (int)WG, off, (int)WG2, off2);
    }
    int rc = check_launch("snvc_sheared_backward_reduce");
    if (rc) return rc;
    sheared_fold_kernel<<<(unsigned)ceil_div<int64_t>(N * C, 128), 128, 0, as_stream(stream)>>>(partial, sums, N * C, (int)grid.x);
    return check_launch("snvc_sheared_backward_reduce(fold)");
}

int snvc_warped_expand(const float *p, const float *q, const float *e, const float *planes, const float *shift, const float *scale,
                       const float *bias, float *y, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int flags, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || D < 1 || H <= 0 || W <= 0 || W % 4 != 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_warped_expand: bad sizes (W % 4 == 0)");
    if ((scale == nullptr) != (bias == nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_warped_expand: scale and bias must both be given or both be NULL");
    if (flags & ~(SNVC_EPI_RELU | SNVC_WARPED_EXPAND_R3)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_warped_expand: only SNVC_EPI_RELU");
    if (N == 0) return SNVC_OK;
    if (!p || !q || !e || !shift || !y) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_warped_expand: null pointer");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(p)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_warped_expand: p, y and planes must be 16-byte aligned");
    const int quads = (int)(W / 4);
    if (quads > 512 || C > 65535 || N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_warped_expand: row too wide or too many channels");
    int RB = 512 / quads;
    if (RB > 8) RB = 8;
    while (RB > 1 && ceil_div<int64_t>(H, RB) * C * N < 4 * 256) RB = (RB + 1) / 2;
    if (flags & SNVC_WARPED_EXPAND_R3) {        // the r3 form (four 16-byte LDS reads per kd and plane): kept for A/B tests
        while (RB > 1 && sizeof(float) * (2 * 3 * (size_t)RB * (2 * W + 8) + 3 * (size_t)RB * D) > 150 * 1024) --RB;
        const int threads = ceil_div(RB * quads, 64) * 64;
        const size_t lds = sizeof(float) * (2 * 3 * (size_t)RB * (2 * W + 8) + 3 * (size_t)RB * D);
        if (lds > 150 * 1024) return fail(SNVC_ERR_UNSUPPORTED, "snvc_warped_expand: rows do not fit the LDS");
        static std::atomic<unsigned> attr{0};
        if (!allow_large_lds(reinterpret_cast<const void *>(&warped_expand_kernel), (int)lds, attr)) return check_launch("snvc_warped_expand");
        warped_expand_kernel<<<dim3((unsigned)ceil_div<int64_t>(H, RB), (unsigned)C, (unsigned)N), threads, lds, as_stream(stream)>>>(
            p, q, e, planes, shift, scale, bias, y, (int)C, (int)D, (int)H, (int)W, RB, flags & SNVC_EPI_RELU);
        return check_launch("snvc_warped_expand");
    }
    auto lds_of = [&](int rb) { return sizeof(float) * (2 * 3 * (size_t)rb * (W + 16) + (size_t)rb * D + 4 + 4 * (size_t)(D + 2)); };
    while (RB > 1 && lds_of(RB) > 150 * 1024) --RB;
    const int threads = ceil_div(RB * quads, 64) * 64;
    const size_t lds = lds_of(RB);
    if (lds > 150 * 1024) return fail(SNVC_ERR_UNSUPPORTED, "snvc_warped_expand: rows do not fit the LDS");
    static std::atomic<unsigned> attr_w{0};
    if (!allow_large_lds(reinterpret_cast<const void *>(&warped_expand_win_kernel), (int)lds, attr_w)) return check_launch("snvc_warped_expand");
    warped_expand_win_kernel<<<dim3((unsigned)ceil_div<int64_t>(H, RB), (unsigned)C, (unsigned)N), threads, lds, as_stream(stream)>>>(
        p, q, e, planes, shift, scale, bias, y, (int)C, (int)D, (int)H, (int)W, RB, flags);
    return check_launch("snvc_warped_expand");
}

int snvc_warped_expand_backward(const float *dy, const float *shift, float *a, float *dplanes, int64_t N, int64_t C, int64_t D,
                                int64_t H, int64_t W, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || D < 2 || H <= 0 || W <= 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_warped_expand_backward: bad sizes (D >= 2)");
    if (N == 0) return SNVC_OK;
    if (!dy || !shift || !a) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_warped_expand_backward: null pointer");
    if (W > 1024 || C > 65535 || N > 65535 || H * W >= ((int64_t)1 << 31))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_warped_expand_backward: W <= 1024, C and N <= 65535");
    const int TW = ceil_div((int)W, 64) * 64;
    int RB = 512 / TW;
    if (RB < 1) RB = 1;
    if (RB > H) RB = (int)H;
    auto lds_of = [&](int rb) { return sizeof(float) * ((((size_t)rb * 16 * (W + 4) + 3) & ~(size_t)3) + 4 * (size_t)D); };
    while (RB > 1 && lds_of(RB) > 150 * 1024) --RB;
    const size_t lds = lds_of(RB);
    if (lds > 150 * 1024) return fail(SNVC_ERR_UNSUPPORTED, "snvc_warped_expand_backward: rows do not fit the LDS");
    static std::atomic<unsigned> attr{0};
    if (!allow_large_lds(reinterpret_cast<const void *>(&warped_expand_bwd_kernel), (int)lds, attr))
        return check_launch("snvc_warped_expand_backward");
    warped_expand_bwd_kernel<<<dim3((unsigned)ceil_div<int64_t>(H, RB), (unsigned)C, (unsigned)N), RB * TW, lds, as_stream(stream)>>>(
        dy, shift, a, dplanes, (int)C, (int)D, (int)H, (int)W, RB, TW);
    return check_launch("snvc_warped_expand_backward");
}

int snvc_warped_expand_split(const float *p, const float *q, const float *e, const float *planes, const float *shift, const float *scale,
                             const float *bias, void *y_hi, void *y_lo, int *overflow, int64_t N, int64_t C, int64_t D, int64_t H,
                             int64_t W, int64_t y_batch_stride, int flags, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || D < 1 || H <= 0 || W <= 0 || W % 4 != 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_warped_expand_split: bad sizes (W % 4 == 0)");
    if ((scale == nullptr) != (bias == nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_warped_expand_split: scale and bias must both be given or both be NULL");
    if (flags & ~(SNVC_EPI_RELU | SNVC_EPI_STREAM_OUT)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_warped_expand_split: only SNVC_EPI_RELU | SNVC_EPI_STREAM_OUT");
    if (N == 0) return SNVC_OK;
    if (!p || !q || !e || !shift || !y_hi || !y_lo) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_warped_expand_split: null pointer");
    if ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo) | reinterpret_cast<uintptr_t>(p)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_warped_expand_split: p and y must be 16-byte aligned");
    const int64_t G = ceil_div<int64_t>(C, 8);
    if (W > 512 || G > 4095 || N > 65535 || H >= ((int64_t)1 << 31))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_warped_expand_split: row too wide (W <= 512) or too many channels");
    int DCH = 1;         // depth chunks: about three workgroups per CU (four fit; every workgroup re-stages its 24 rows).  r5, cfg2: 1 chunk
                         // 251 us, 2 chunks 185, 4 chunks 225
    while (DCH < 16 && H * G * N * DCH < 3 * 256 && D / (2 * DCH) >= 8) DCH *= 2;
    const int DC = (int)ceil_div<int64_t>(D, DCH);
    const size_t lds = sizeof(float) * (24 * (size_t)(W + 16) + 96 + 8 * (size_t)DC + 4 + 4 * (size_t)(D + 2));
    if (lds > 150 * 1024) return fail(SNVC_ERR_UNSUPPORTED, "snvc_warped_expand_split: the row does not fit the LDS");
    const int threads = ceil_div((int)W, 64) * 64;
    const dim3 grid((unsigned)H, (unsigned)(G * DCH), (unsigned)N);
    const int64_t y_bs = y_batch_stride ? y_batch_stride : 2 * G * 8 * D * H * W;
    static std::atomic<unsigned> attr{0};
    if (!allow_large_lds(reinterpret_cast<const void *>(&warped_expand_split_kernel), (int)lds, attr)) return check_launch("snvc_warped_expand_split");
    warped_expand_split_kernel<<<grid, threads, lds, as_stream(stream)>>>(p, q, e, planes, shift, scale, bias, reinterpret_cast<_Float16 *>(y_hi),
                                                                          reinterpret_cast<_Float16 *>(y_lo), overflow, (int)C, (int)D, (int)H,
                                                                          (int)W, DCH, DC, y_bs, flags);
    return check_launch("snvc_warped_expand_split");
}

int snvc_shift_structure(const float *shift, float *out4, int64_t N, int64_t D, void *stream) {
    using namespace snvc;
    if (N <= 0 || D <= 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_shift_structure: sizes must be positive");
    if (!shift || !out4) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_shift_structure: null pointer");
    if (D >= (1 << 24)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_shift_structure: D must stay exactly representable in fp32");
    shift_structure_kernel<<<1, 256, 0, as_stream(stream)>>>(shift, out4, N * D, (int)D);
    return check_launch("snvc_shift_structure");
}

int snvc_sheared_reduce(const float *dy, float *dg, float *dgcol, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int q,
                        int m0, int64_t WG, int off, int64_t WG2, int off2, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || D < 2 || H <= 0 || W < 2 || (q != 1 && q != 2) || m0 < 0 || WG <= 0 || WG2 <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_reduce: bad sizes (q in {1,2}, D >= 2, W >= 2)");
    if (N == 0) return SNVC_OK;
    if (!dy || !dg || !dgcol) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_reduce: null pointer");
    if (C > 65535 || N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_reduce: too many channels or samples");
    const dim3 grid((unsigned)H, (unsigned)C, (unsigned)N);
    if (q == 1)
        sheared_reduce_kernel<1><<<grid, 256, 0, as_stream(stream)>>>(dy, dg, dgcol, (int)C, (int)D, (int)H, (int)W, m0, (int)WG, off,
                                                                     (int)WG2, off2);
    else
        sheared_reduce_kernel<2><<<grid, 256, 0, as_stream(stream)>>>(dy, dg, dgcol, (int)C, (int)D, (int)H, (int)W, m0, (int)WG, off,
                                                                     (int)WG2, off2);
    return check_launch("snvc_sheared_reduce");
}

static void sheared_wgrad_geometry(int64_t WU, int &col_chunks, int &chunk_cols) {
    using snvc::ceil_div;
    col_chunks = WU > 320 ? 2 : 1;                                     // two workgroups per row on the wide grid
    chunk_cols = (int)((ceil_div<int64_t>(WU, col_chunks) + snvc::SW_IC - 1) / snvc::SW_IC * snvc::SW_IC);
}

int64_t snvc_sheared_wgrad_workspace_bytes(int64_t N, int64_t CO, int64_t H, int64_t WU) {
    if (N < 0 || CO <= 0 || H <= 0 || WU <= 0) return -1;
    int cc, cols;
    sheared_wgrad_geometry(WU, cc, cols);
    return N * ((CO + 31) / 32) * H * cc * 21 * 1024 * (int64_t)sizeof(float);
}

int snvc_sheared_wgrad(const float *x, const float *dy, float *dk, void *workspace, int64_t N, int64_t C, int64_t CO, int64_t H,
                       int64_t WU, void *stream) {
    using namespace snvc;
    if (N <= 0 || C <= 0 || C > 32 || CO <= 0 || H <= 0 || WU <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_wgrad: bad sizes (1 <= C <= 32)");
    if (!x || !dy || !dk || !workspace) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_wgrad: null pointer");
    int cc, cols;
    sheared_wgrad_geometry(WU, cc, cols);
    const int groups = (int)((CO + 31) / 32), nwg = (int)(H * cc);
    if (N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_wgrad: too many samples");
    sheared_wgrad_kernel<<<dim3((unsigned)nwg, (unsigned)groups, (unsigned)N), 256, 0, as_stream(stream)>>>(
        x, dy, static_cast<float *>(workspace), (int)C, (int)CO, (int)H, (int)WU, cc, cols);
    sheared_wgrad_reduce_kernel<<<(unsigned)ceil_div(groups * 21 * 1024, 256), 256, 0, as_stream(stream)>>>(
        static_cast<const float *>(workspace), dk, (int)C, (int)CO, (int)N, groups, nwg);
    return check_launch("snvc_sheared_wgrad");
}

int snvc_sheared_upsample_backward(const float *drq, float *dright, int64_t N, int64_t C, int64_t H, int64_t W, int q, int64_t WU,
                                   int off, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || (q != 1 && q != 2) || WU <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_upsample_backward: bad sizes (q in {1,2})");
    if (N == 0) return SNVC_OK;
    if (!drq || !dright) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_upsample_backward: null pointer");
    const int64_t rows = N * C * H, total = rows * W;
    sheared_upsample_bwd_kernel<<<(unsigned)ceil_div<int64_t>(total, 256), 256, 0, as_stream(stream)>>>(drq, dright, (int)W, q, (int)WU,
                                                                                                         off, rows);
    return check_launch("snvc_sheared_upsample_backward");
}

}  // extern "C"
