// Small-layer kernels of the 3D stack that do not belong on the matrix pipe (gfx950).
//
// deconv3d_cout1_kernel: ConvTranspose3d(Cin, 1, k3, s2, p1, op1) + fused epilogue.
//   Where it comes from: the global model ends in `hourglass(v)[0] + v -> classifier Conv3d(C,1,1)` (the
//   composition of snvc/models/vernier.py:366-371; hourglass: snvc/models/submodule.py:127-146,166).  The
//   hourglass's last layer is ConvTranspose3d(2C, C) + BatchNorm with NO activation behind it
//   (submodule.py:166), so classifier(bn(deconv(post)) + v) is linear in `post`:
//       = deconv'(post) + b' + classifier(v),   W'[ci][tap] = sum_co h[co]*bn_scale[co]*W[ci][co][tap],
//   a transposed convolution to ONE channel (1/32 of the multiply-adds; models/submodule.py folds the weights in
//   fp64).  One output channel is 1/32 of an MFMA tile, and the layer reads its input once: a VALU kernel.
//   A thread owns 4 consecutive input columns of one input row (16-byte loads, lanes consecutive along W and on
//   into the next rows: the tensor is walked linearly) and produces the 2 x 2 x 8 outputs above them; the 27
//   weights of a channel are wave-uniform (scalar loads).  Per dimension, output parity 0 takes tap 1 of input
//   i, parity 1 takes tap 2 of input i and tap 0 of input i+1 (o = 2i - 1 + k).
#include "conv3d_internal.hpp"

namespace snvc {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float epi_f(float v, float res, int flags) {
    if (flags & SNVC_EPI_ADD_PRE) v += res;
    if (flags & SNVC_EPI_RELU) v = v > 0.0f ? v : 0.0f;
    if (flags & SNVC_EPI_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    if (flags & SNVC_EPI_ADD_POST) v += res;
    return v;
}

__global__ void __launch_bounds__(256)
deconv3d_cout1_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ scale,
                      const float *__restrict__ bias, const float *__restrict__ res, float *__restrict__ y, int Cin,
                      int Din, int Hin, int Win, int64_t x_bs, int64_t y_bs, int64_t r_bs, int flags) {
    const int nq = Win >> 2;
    const int total = Din * Hin * nq;
    // workgroups in XCD-local order: workgroup b runs on XCD b % 8, and a thread reads the rows of depth id AND id + 1 -- dealt out
    // round-robin, the 7 workgroups of a depth plane and those of the next plane sit on different XCDs and every input element is
    // fetched into two L2s; with a contiguous range of planes per XCD only the range borders are (cfg2: 84 -> 82 us, not the
    // kernel's limit)
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, k8 = blockIdx.x >> 3;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k8;
    const int i = wg * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = i % nq, t = i / nq, ih = t % Hin, id = t / Hin;
    const int64_t n = blockIdx.y;
    const int in_hw = Hin * Win;
    const int64_t in_dhw = (int64_t)in_hw * Din;
    const bool h1 = ih + 1 < Hin, d1 = id + 1 < Din, w1 = 4 * q + 4 < Win;
    // neighbours beyond the tensor are read from a valid address and zeroed by a select
    const int oh = h1 ? Win : 0, od = d1 ? in_hw : 0, on = w1 ? 4 : 3;
    const float *p0 = x + n * x_bs + (int64_t)id * in_hw + ih * Win + 4 * q;

    float acc[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[a][b][j] = 0.0f;

    // Software pipeline: the 8 loads of channel c+1 are issued before channel c's 108 FMAs (the scheduler, left alone,
    // sinks every load to its first use and waits for each one: 8 exposed round trips per channel, 122 us on cfg2).
    struct Raw {
        f32x4 v[2][2];
        float nb[2][2];
    };
    auto issue = [&](int c, Raw &r) {
        const float *pc = p0 + c * in_dhw;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const float *pr = pc + (dd ? od : 0) + (hh ? oh : 0);
                r.v[dd][hh] = *reinterpret_cast<const f32x4 *>(pr);
                r.nb[dd][hh] = pr[on];
            }
    };
    Raw cur;
    issue(0, cur);
    for (int c = 0; c < Cin; ++c) {
        Raw nxt;
        issue(c + 1 < Cin ? c + 1 : c, nxt);     // the last iteration re-reads its own channel (no branch, discarded)
        const float *wc = w + c * 27;            // wave-uniform: scalar loads
        float wk[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) wk[k] = wc[k];
        __builtin_amdgcn_sched_barrier(0);
        float X[2][2][5];
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const bool ok = (dd == 0 || d1) && (hh == 0 || h1);
#pragma unroll
                for (int j = 0; j < 4; ++j) X[dd][hh][j] = ok ? cur.v[dd][hh][j] : 0.0f;
                X[dd][hh][4] = (ok && w1) ? cur.nb[dd][hh] : 0.0f;
            }
#pragma unroll
        for (int pd = 0; pd < 2; ++pd)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int jd = 0; jd <= pd; ++jd)
#pragma unroll
                    for (int jh = 0; jh <= ph; ++jh) {
                        // parity 0: (tap 1, offset 0); parity 1: j = 0 -> (tap 2, offset 0), j = 1 -> (tap 0, offset 1)
                        const int kd = pd ? (jd ? 0 : 2) : 1, kh = ph ? (jh ? 0 : 2) : 1;
                        const float w0 = wk[(kd * 3 + kh) * 3 + 0], w1k = wk[(kd * 3 + kh) * 3 + 1],
                                    w2 = wk[(kd * 3 + kh) * 3 + 2];
                        const float(&r)[5] = X[jd][jh];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[pd][ph][2 * j] = __builtin_fmaf(r[j], w1k, acc[pd][ph][2 * j]);
                            acc[pd][ph][2 * j + 1] =
                                __builtin_fmaf(r[j + 1], w0, __builtin_fmaf(r[j], w2, acc[pd][ph][2 * j + 1]));
                        }
                    }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }

    const float sc = scale ? scale[0] : 1.0f, bi = scale ? bias[0] : 0.0f;
    const int Wout = 2 * Win;
    const int64_t out_hw = (int64_t)4 * in_hw;
    float *yn = y + n * y_bs;
    const float *rn = res ? res + n * r_bs : nullptr;
    f32x4 rv[2][2][2];
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int64_t sp = (int64_t)(2 * id + pd) * out_hw + (int64_t)(2 * ih + ph) * Wout + 8 * q;
#pragma unroll
            for (int k = 0; k < 2; ++k)
                rv[pd][ph][k] = rn ? *reinterpret_cast<const f32x4 *>(rn + sp + 4 * k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int64_t sp = (int64_t)(2 * id + pd) * out_hw + (int64_t)(2 * ih + ph) * Wout + 8 * q;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = epi_f(acc[pd][ph][4 * k + j] * sc + bi, rv[pd][ph][k][j], flags);
                *reinterpret_cast<f32x4 *>(yn + sp + 4 * k) = o;
            }
        }
}

}  // namespace

bool deconv3d_cout1_qualifies(const snvc_conv3d_desc &d, const float *x, const float *y, const float *res, int64_t x_bs,
                              int64_t y_bs, int64_t r_bs) {
    if (!d.transposed || d.Cout != 1 || d.ksize != 3 || d.stride != 2 || d.pad != 1 || d.dilation != 1 || d.ksize_d == 1)
        return false;
    if (d.Win % 4 != 0 || x_bs % 4 != 0 || y_bs % 4 != 0 || r_bs % 4 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) & 15) return false;
    const int64_t threads = (int64_t)d.Din * d.Hin * (d.Win / 4);
    return threads < ((int64_t)1 << 31) - 256 && d.N <= 65535;
}

void deconv3d_cout1_launch(const snvc_conv3d_desc &d, const float *x, const float *w, const float *scale,
                           const float *bias, const float *res, float *y, int64_t x_bs, int64_t y_bs, int64_t r_bs,
                           hipStream_t st) {
    const int64_t threads = (int64_t)d.Din * d.Hin * (d.Win / 4);
    const dim3 grid((unsigned)ceil_div<int64_t>(threads, 256), (unsigned)d.N);
    deconv3d_cout1_kernel<<<grid, 256, 0, st>>>(x, w, scale, bias, res, y, d.Cin, d.Din, d.Hin, d.Win, x_bs, y_bs, r_bs,
                                                d.flags);
}

}  // namespace snvc
