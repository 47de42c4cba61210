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

// One workgroup = RB rows of one (n, co) plane; thread = (row, quad of 4 columns) and walks d = 1 .. D-2.
// Window: the thread's four values G[i0 + q*k], k = 0..3, with i0 = q*4*quad - d - m0 + off.  For q = 2 the even- and the
// odd-indexed elements of a G row are kept as two LDS arrays, so a window is 4 CONSECUTIVE elements of one of them and
// moving from plane d to d + q shifts it down by one: one 4-byte LDS read per thread and plane.
template <int Q>
__global__ void __launch_bounds__(512)
sheared_expand_kernel(const float *__restrict__ g, const float *__restrict__ gcol, const float *__restrict__ planes,
                      const float *__restrict__ scale, const float *__restrict__ bias, float *__restrict__ y, int C, int D, int H,
                      int W, int m0, int WG, int off, int WG2, int off2, int RB, int flags) {
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
    if (tid >= rows * quads) return;
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
        *reinterpret_cast<f32x4 *>(yp + (int64_t)d * plane_sz) = o;
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
            *reinterpret_cast<f32x4 *>(yp + (int64_t)d * plane_sz) = o;      // plain store: conv2 reads the tail from the caches
            // next plane of this chain is d + Q: every index drops by Q, i.e. by ONE element of the chain's phase array
            win[p][3] = win[p][2]; win[p][2] = win[p][1]; win[p][1] = win[p][0];
            win[p][0] = load(Q * w0 - (d + Q) - m0 + off);
        }
    }
}

}  // namespace
}  // namespace snvc

extern "C" {

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

int snvc_sheared_expand(const float *g, const float *gcol, const float *planes, const float *scale, const float *bias, float *y,
                        int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int q, int m0, int64_t WG, int off, int64_t WG2,
                        int off2, int flags, void *stream) {
    using namespace snvc;
    if (N < 0 || C <= 0 || D < 2 || H <= 0 || W <= 0 || W % 4 != 0 || (q != 1 && q != 2) || m0 < 0 || WG <= 0 || WG2 <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand: bad sizes (W % 4 == 0, q in {1,2}, D >= 2)");
    if ((scale == nullptr) != (bias == nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand: scale and bias must both be given or both be NULL");
    if (flags & ~SNVC_EPI_RELU) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand: only SNVC_EPI_RELU");
    if (N == 0) return SNVC_OK;
    if (!g || !gcol || !y) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand: null pointer");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(planes)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_expand: y and planes must be 16-byte aligned");
    const int quads = (int)(W / 4);
    if (quads > 512 || C > 65535 || N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand: row too wide or too many channels");
    int RB = 512 / quads;                       // rows per workgroup: as many as 512 threads cover ...
    if (RB > 8) RB = 8;
    while (RB > 1 && ceil_div<int64_t>(H, RB) * C * N < 4 * 256) RB = (RB + 1) / 2;      // ... while the chip stays covered
    const int threads = ceil_div(RB * quads, 64) * 64;
    const int LW = (int)((WG + q - 1) / q) + 4;
    const size_t lds = sizeof(float) * ((size_t)RB * q * LW + (size_t)RB * D);
    if (lds > 150 * 1024) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_expand: rows do not fit the LDS");
    const dim3 grid((unsigned)ceil_div<int64_t>(H, RB), (unsigned)C, (unsigned)N);
    static std::atomic<unsigned> attr1{0}, attr2{0};
    if (q == 1) {
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_kernel<1>), (int)lds, attr1)) return check_launch("snvc_sheared_expand");
        sheared_expand_kernel<1><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, y, (int)C, (int)D, (int)H, (int)W,
                                                                          m0, (int)WG, off, (int)WG2, off2, RB, flags);
    } else {
        if (!allow_large_lds(reinterpret_cast<const void *>(&sheared_expand_kernel<2>), (int)lds, attr2)) return check_launch("snvc_sheared_expand");
        sheared_expand_kernel<2><<<grid, threads, lds, as_stream(stream)>>>(g, gcol, planes, scale, bias, y, (int)C, (int)D, (int)H, (int)W,
                                                                          m0, (int)WG, off, (int)WG2, off2, RB, flags);
    }
    return check_launch("snvc_sheared_expand");
}

}  // extern "C"
