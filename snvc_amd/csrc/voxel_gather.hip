// Feature -> voxel resampling for gfx950.
//
// Replaces VernierScale._sample_2d_feat (snvc/models/vernier.py:323-349): the reference
// normalises the projected coordinates (two element-wise passes per side, :335-338), runs
// F.grid_sample twice (:339-340) and then copies both results again for torch.cat (:346).
// Here one kernel does all of it: each thread owns one voxel, derives the 4 bilinear taps
// and weights of both cameras once, then walks the 2F channels writing coalesced rows of
// the concatenated [N,2F,V] output.  The feature maps (<= 1 MB per side) stay in L2; the
// kernel is bound by the output write.
//
// Arithmetic mirrors ATen's CPU grid_sampler_2d for (bilinear, zeros, align_corners=False)
// -- aten/src/ATen/native/cpu/GridSamplerKernel.cpp: unnormalise = (g + 1) * (size / 2) - 0.5,
// w = x - floor(x), e = 1 - w, n = y - floor(y), s = 1 - n, weights nw = s*e, ne = s*w,
// sw = n*e, se = n*w, out-of-range taps contribute 0 -- with every operation rounded on its
// own (fp contract off), so it is bit-identical to oracle/numpy_ref.py:sample_2d_feat.
#include "common.hpp"

#include <cmath>

namespace snvc {
namespace {

#pragma clang fp contract(off)

struct Taps {
    int off[4];    // element offsets inside one feature plane, -1 = out of range
    float wt[4];   // nw, ne, sw, se
};

// POW2: res_x and res_y are powers of two and (res_x, res_y) hold their RECIPROCALS: p / 2^k and p * 2^-k are the same
// real number rounded once, so the product is bit-identical to the division (subnormal results included) and costs
// one instruction instead of the ~12 of a correctly rounded fp32 division (the path's crops are 256 x 256).
template <bool POW2 = false>
__device__ __forceinline__ Taps make_taps(float px, float py, float res_x, float res_y, int Hf,
                                          int Wf) {
    // vernier.py:335-338: p / res * 2 - 1 (three separately rounded fp32 ops)
    const float gx = (POW2 ? px * res_x : px / res_x) * 2.0f - 1.0f;
    const float gy = (POW2 ? py * res_y : py / res_y) * 2.0f - 1.0f;
    const float x = (gx + 1.0f) * ((float)Wf / 2.0f) - 0.5f;
    const float y = (gy + 1.0f) * ((float)Hf / 2.0f) - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float w = x - xf, e = 1.0f - w, n = y - yf, s = 1.0f - n;
    Taps t;
    t.wt[0] = s * e; t.wt[1] = s * w; t.wt[2] = n * e; t.wt[3] = n * w;
    // float -> int conversion only after a range test (NaN / huge coordinates give no taps)
    const bool finite_range = (xf >= -2.0f && xf <= (float)Wf + 1.0f && yf >= -2.0f && yf <= (float)Hf + 1.0f);
    const int x0 = finite_range ? (int)xf : -2, y0 = finite_range ? (int)yf : -2;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < Wf, vx1 = x1 >= 0 && x1 < Wf;
    const bool vy0 = y0 >= 0 && y0 < Hf, vy1 = y1 >= 0 && y1 < Hf;
    t.off[0] = (vx0 && vy0) ? y0 * Wf + x0 : -1;
    t.off[1] = (vx1 && vy0) ? y0 * Wf + x1 : -1;
    t.off[2] = (vx0 && vy1) ? y1 * Wf + x0 : -1;
    t.off[3] = (vx1 && vy1) ? y1 * Wf + x1 : -1;
    return t;
}

// Loads are unconditional (an out-of-range tap reads element 0 of the plane) and the padding zero is
// a select on the loaded value: no branches, so the four taps of a channel -- and, after unrolling,
// of several channels -- are in flight together.
__device__ __forceinline__ float apply_taps(const float *__restrict__ plane, const Taps &t) {
    const float a0 = plane[t.off[0] < 0 ? 0 : t.off[0]];
    const float b0 = plane[t.off[1] < 0 ? 0 : t.off[1]];
    const float c0 = plane[t.off[2] < 0 ? 0 : t.off[2]];
    const float d0 = plane[t.off[3] < 0 ? 0 : t.off[3]];
    const float a = t.off[0] >= 0 ? a0 : 0.0f;
    const float b = t.off[1] >= 0 ? b0 : 0.0f;
    const float c = t.off[2] >= 0 ? c0 : 0.0f;
    const float d = t.off[3] >= 0 ? d0 : 0.0f;
    return a * t.wt[0] + b * t.wt[1] + c * t.wt[2] + d * t.wt[3];
}

__global__ void __launch_bounds__(256)
voxel_gather_fwd(const float *__restrict__ left, const float *__restrict__ right,
                 const float *__restrict__ l_pts, const float *__restrict__ r_pts,
                 float *__restrict__ out, int F, int Hf, int Wf, int64_t V, float res_x,
                 float res_y) {
    const int64_t n = blockIdx.y;
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float *lp = l_pts + n * 2 * V, *rp = r_pts + n * 2 * V;
    const Taps tl = make_taps(lp[v], lp[V + v], res_x, res_y, Hf, Wf);
    const Taps tr = make_taps(rp[v], rp[V + v], res_x, res_y, Hf, Wf);
    const int plane = Hf * Wf;
    const float *lf = left + n * F * plane, *rf = right + n * F * plane;
    float *o = out + n * 2 * F * V + v;
#pragma unroll 8
    for (int c = 0; c < F; ++c) o[(int64_t)c * V] = apply_taps(lf + (int64_t)c * plane, tl);
    o += (int64_t)F * V;
#pragma unroll 8
    for (int c = 0; c < F; ++c) o[(int64_t)c * V] = apply_taps(rf + (int64_t)c * plane, tr);
}


// ---- channels-last forward path ----
// The plain kernel issues 4 gather loads per channel per camera (256 per voxel at F = 32) and is bound by
// load-instruction issue, not by HBM (L1 hit rate 99 %).  With the feature maps re-laid [pixel][F] (a 2 x
// N x F x Hf x Wf float workspace, <= 4 MB on the path) one 16-byte load fetches 4 channels of a tap:
// 64 loads per voxel instead of 256, same taps, same weights, same separately rounded
// a*nw + b*ne + c*sw + d*se per channel, so the result stays bit-identical.
__global__ void __launch_bounds__(256)
features_to_channels_last(const float *__restrict__ left, const float *__restrict__ right, float *__restrict__ ws,
                          int F, int plane) {
    // grid: (pixel blocks, N, 2 sides); LDS tile [64 pixels][F+1]
    extern __shared__ float tile[];
    const int side = blockIdx.z;
    const int64_t n = blockIdx.y;
    const float *src = (side == 0 ? left : right) + n * F * (int64_t)plane;
    float *dst = ws + ((int64_t)side * gridDim.y + n) * F * (int64_t)plane;
    const int p0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * F; i += blockDim.x) {
        const int c = i / 64, p = i % 64;                       // coalesced along pixels
        tile[p * (F + 1) + c] = p0 + p < plane ? src[(int64_t)c * plane + p0 + p] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * F; i += blockDim.x) {
        const int p = i / F, c = i % F;                         // coalesced along channels
        if (p0 + p < plane) dst[(int64_t)(p0 + p) * F + c] = tile[p * (F + 1) + c];
    }
}

__device__ __forceinline__ void gather_side_cl(const float *__restrict__ feat_cl, const Taps &t, int F, int64_t V,
                                               float *__restrict__ o) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 *pa = reinterpret_cast<const f4 *>(feat_cl + (int64_t)(t.off[0] < 0 ? 0 : t.off[0]) * F);
    const f4 *pb = reinterpret_cast<const f4 *>(feat_cl + (int64_t)(t.off[1] < 0 ? 0 : t.off[1]) * F);
    const f4 *pc = reinterpret_cast<const f4 *>(feat_cl + (int64_t)(t.off[2] < 0 ? 0 : t.off[2]) * F);
    const f4 *pd = reinterpret_cast<const f4 *>(feat_cl + (int64_t)(t.off[3] < 0 ? 0 : t.off[3]) * F);
    const bool va = t.off[0] >= 0, vb = t.off[1] >= 0, vc = t.off[2] >= 0, vd = t.off[3] >= 0;
#pragma unroll 4
    for (int c4 = 0; c4 < F / 4; ++c4) {
        const f4 a4 = pa[c4], b4 = pb[c4], c4v = pc[c4], d4 = pd[c4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = va ? a4[j] : 0.0f, b = vb ? b4[j] : 0.0f, c = vc ? c4v[j] : 0.0f, d = vd ? d4[j] : 0.0f;
            o[(int64_t)(4 * c4 + j) * V] = a * t.wt[0] + b * t.wt[1] + c * t.wt[2] + d * t.wt[3];
        }
    }
}

__global__ void __launch_bounds__(256)
voxel_gather_fwd_cl(const float *__restrict__ ws, const float *__restrict__ l_pts, const float *__restrict__ r_pts,
                    float *__restrict__ out, int F, int Hf, int Wf, int64_t V, float res_x, float res_y) {
    const int64_t n = blockIdx.y;
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float *lp = l_pts + n * 2 * V, *rp = r_pts + n * 2 * V;
    const Taps tl = make_taps(lp[v], lp[V + v], res_x, res_y, Hf, Wf);
    const Taps tr = make_taps(rp[v], rp[V + v], res_x, res_y, Hf, Wf);
    const int64_t fp = (int64_t)F * Hf * Wf;
    const float *lf = ws + n * fp, *rf = ws + ((int64_t)gridDim.y + n) * fp;
    float *o = out + n * 2 * F * V + v;
    gather_side_cl(lf, tl, F, V, o);
    gather_side_cl(rf, tr, F, V, o + (int64_t)F * V);
}

// Four consecutive voxels per thread on the channels-last maps: a wave writes 1 KB contiguous per channel
// row (8 whole cache lines) instead of 256 B, which is what the HBM write stream wants.
__global__ void __launch_bounds__(256)
voxel_gather_fwd_cl_x4(const float *__restrict__ ws, const float *__restrict__ l_pts, const float *__restrict__ r_pts,
                       float *__restrict__ out, int F, int Hf, int Wf, int64_t V, float res_x, float res_y) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int64_t n = blockIdx.y;
    const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (v >= V) return;
    const int64_t fp = (int64_t)F * Hf * Wf;
    float *o = out + n * 2 * F * V + v;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const float *pts = (side == 0 ? l_pts : r_pts) + n * 2 * V;
        const float *feat = ws + ((int64_t)side * gridDim.y + n) * fp;
        const f4 px = *reinterpret_cast<const f4 *>(pts + v);
        const f4 py = *reinterpret_cast<const f4 *>(pts + V + v);
        Taps t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = make_taps(px[k], py[k], res_x, res_y, Hf, Wf);
#pragma unroll 2
        for (int c4 = 0; c4 < F / 4; ++c4) {
            f4 res[4];   // res[j] = channel 4*c4+j of the 4 voxels
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f4 a4 = reinterpret_cast<const f4 *>(feat + (int64_t)(t[k].off[0] < 0 ? 0 : t[k].off[0]) * F)[c4];
                const f4 b4 = reinterpret_cast<const f4 *>(feat + (int64_t)(t[k].off[1] < 0 ? 0 : t[k].off[1]) * F)[c4];
                const f4 c4v = reinterpret_cast<const f4 *>(feat + (int64_t)(t[k].off[2] < 0 ? 0 : t[k].off[2]) * F)[c4];
                const f4 d4 = reinterpret_cast<const f4 *>(feat + (int64_t)(t[k].off[3] < 0 ? 0 : t[k].off[3]) * F)[c4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = t[k].off[0] >= 0 ? a4[j] : 0.0f, b = t[k].off[1] >= 0 ? b4[j] : 0.0f;
                    const float c = t[k].off[2] >= 0 ? c4v[j] : 0.0f, d = t[k].off[3] >= 0 ? d4[j] : 0.0f;
                    res[j][k] = a * t[k].wt[0] + b * t[k].wt[1] + c * t[k].wt[2] + d * t[k].wt[3];
                }
            }
#pragma unroll
            // non-temporal: the voxel tensor (0.2 GB per instance) is streamed out once and read by the next
            // kernel from HBM anyway; measured 3.9 -> 5.0 TB/s
            for (int j = 0; j < 4; ++j)
                __builtin_nontemporal_store(res[j], reinterpret_cast<f4 *>(o + (int64_t)(4 * c4 + j) * V));
        }
        o += (int64_t)F * V;
    }
}


// ---- LDS-staged forward path (the default when the feature plane fits) ----
// The channels-last kernels above read every bilinear tap through the vector L1: 1 KB of L1 traffic per voxel for
// 256 output bytes at F = 32.  On coherent coordinates (neighbouring lanes hit the same pixel) that is invisible; on
// real projections it is the bound (2.4-2.8 TB/s measured with GridProjector coordinates, 1.4 TB/s on uniform random
// ones, against 5.1 TB/s on a degenerate line).  Here a workgroup owns a run of voxels and a slice of CS channels:
// per camera it copies that channel slice of the WHOLE feature map into LDS ([pixel][CS] floats, 64 KB at 64x64x4;
// LDS-DMA from the channels-last workspace), then resolves its taps with ds_read_b128 -- LDS serves 256 B/clk/CU
// where the L1 serves 64, and its rate does not depend on how coherent the coordinates are.  Same taps, same
// separately rounded a*nw + b*ne + c*sw + d*se: bit-identical to the other kernels.  Two workgroups fit a CU, so one
// streams its outputs while the other refills.
// Taps of one voxel for the LDS kernels, written so that the x and y halves of the arithmetic pair up into packed fp32
// instructions (v_pk_mul_f32 / v_pk_add_f32: two floats per lane per issue slot -- the kernel is VALU-issue bound:
// profiles/r3/pmc_gather.txt, 88 % of the SIMD issue cycles): the same separately rounded operations as make_taps on
// the same values.  The range test + conversion of make_taps becomes a clamp of floor(x) to [-2, size + 1] before the
// conversion (fmax / fmin return the non-NaN operand: a NaN or far-away coordinate lands on -2 or size + 1, where both
// of its columns / rows fail the bounds test exactly as they fail make_taps's finite_range test).  `a[k]` are FLOAT
// indices into the LDS slice ([pixel][CS]); a tap outside the map points at the zero slot `zidx`.
typedef float f2 __attribute__((ext_vector_type(2)));
struct TapsLds {
    int a[4];
    float wt[4];   // nw, ne, sw, se
};

template <bool POW2, int CS>
__device__ __forceinline__ TapsLds make_taps_lds(float px, float py, f2 res, f2 half_size, int Hf, int Wf, int zidx) {
    const f2 p = {px, py};
    const f2 g = (POW2 ? p * res : p / res) * 2.0f - 1.0f;          // vernier.py:335-338, three separately rounded ops
    const f2 xy = (g + 1.0f) * half_size - 0.5f;
    const f2 fl = {floorf(xy[0]), floorf(xy[1])};
    const f2 wn = xy - fl, es = 1.0f - wn;                             // (w, n), (e, s)
    const f2 ew = {es[0], wn[0]};
    const f2 top = f2{es[1], es[1]} * ew, bot = f2{wn[1], wn[1]} * ew; // (s*e, s*w), (n*e, n*w)
    TapsLds t;
    t.wt[0] = top[0]; t.wt[1] = top[1]; t.wt[2] = bot[0]; t.wt[3] = bot[1];
    const int x0 = (int)fminf(fmaxf(fl[0], -2.0f), (float)Wf + 1.0f);
    const int y0 = (int)fminf(fmaxf(fl[1], -2.0f), (float)Hf + 1.0f);
    const bool vx0 = (unsigned)x0 < (unsigned)Wf, vx1 = (unsigned)(x0 + 1) < (unsigned)Wf;
    const bool vy0 = (unsigned)y0 < (unsigned)Hf, vy1 = (unsigned)(y0 + 1) < (unsigned)Hf;
    const int base = (y0 * Wf + x0) * CS;
    t.a[0] = (vx0 && vy0) ? base : zidx;
    t.a[1] = (vx1 && vy0) ? base + CS : zidx;
    t.a[2] = (vx0 && vy1) ? base + Wf * CS : zidx;
    t.a[3] = (vx1 && vy1) ? base + Wf * CS + CS : zidx;
    return t;
}

// r3: out-of-range taps point at a 16-byte ZERO slot behind the image instead of being zeroed by a select per loaded
// value (+0.0 * w is what the select form computes too: same bits, 16 selects per voxel fewer); the coordinate
// normalisation multiplies by the reciprocal when the crop resolution is a power of two (make_taps<POW2>); and the
// channel slices of one voxel run are dispatched back to back on ONE XCD (blocks b and b + 8 share an XCD), so that the
// run's coordinates are read from HBM once and from that XCD's L2 by the other slices.
// OUT: 0 = fp32 NCDHW; 1 = C8 half; 2 = a split C8 pair (r4, conv3d_f16.hip F16Cfg::PL: value * *mul_dev = hi + lo, the lo plane
// `lo_off` halves behind the hi plane) -- the SAME separately rounded fp32 bilinear sum in all three.
template <int CS, int OUT, int BS, bool POW2>
__global__ void __launch_bounds__(BS)
voxel_gather_fwd_lds(const float *__restrict__ ws, const float *__restrict__ l_pts, const float *__restrict__ r_pts,
                     void *__restrict__ out_, int F, int Hf, int Wf, int64_t V, int64_t vrun, int nruns, float res_x,
                     float res_y, const float *__restrict__ mul_dev, int64_t lo_off, int64_t out_bs) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr bool HALF = OUT != 0;
    static_assert(HALF ? CS == 8 : CS == 4, "fp32 output: 4-channel slices; C8 half output: one 8-channel group");
    const float mul = OUT == 2 ? *mul_dev : 1.0f;
    constexpr int U = HALF ? 1 : 4;          // voxels per thread and step
    extern __shared__ __attribute__((aligned(16))) float img[];
    const int tid = threadIdx.x;
    // block -> (voxel run, channel slice): XCD x = b % 8 owns the runs {x, x + 8, ...}; its blocks walk (run, slice)
    // with the slice fastest.  Blocks beyond the last run of their XCD have nothing to do (grid = 8 * ceil(nruns/8) * slices).
    const int nsl = F / CS;
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int cs = kk % nsl, run = (kk / nsl) * 8 + xcd;
    if (run >= nruns) return;
    const int64_t n = blockIdx.z;
    const int plane = Hf * Wf;
    const int64_t v0 = (int64_t)run * vrun;
    const int64_t v1 = v0 + vrun < V ? v0 + vrun : V;
    const int pieces = plane * (CS / 4);                      // 16-byte pieces of the slice
    const int zidx = plane * CS;                              // float index of the zero slot
    if (tid < CS / 4) reinterpret_cast<f4 *>(img)[pieces + tid] = f4(0.0f);
    const f2 res2 = {res_x, res_y};
    const f2 half_size = {(float)Wf / 2.0f, (float)Hf / 2.0f};
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
        const float *feat = ws + ((int64_t)side * gridDim.z + n) * plane * F + cs * CS;
        for (int i0 = 0; i0 < pieces; i0 += BS) {
            const int i = i0 + tid;
            const int px = i / (CS / 4), hq = i - px * (CS / 4);
            if (i < pieces)
                __builtin_amdgcn_global_load_lds(feat + (int64_t)px * F + 4 * hq, img + 4 * (i0 + (tid & ~63)), 16, 0, 0);
        }
        const float *pts = (side == 0 ? l_pts : r_pts) + n * 2 * V;
        // coordinates of the first step are fetched while the slice lands; every step fetches the next one's first
        int64_t v = v0 + (int64_t)tid * U;
        f4 cx = f4(0.0f), cy = f4(0.0f);
        auto fetch = [&](int64_t vv) {
            if (vv < v1) {
                if constexpr (U == 4) { cx = *reinterpret_cast<const f4 *>(pts + vv); cy = *reinterpret_cast<const f4 *>(pts + V + vv); }
                else { cx[0] = pts[vv]; cy[0] = pts[V + vv]; }
            }
        };
        fetch(v);
        __syncthreads();       // drains the DMA: the slice is in LDS
        for (; v < v1; v += BS * U) {
            const f4 px = cx, py = cy;
            fetch(v + BS * U);
            if constexpr (!HALF) {
                f4 res[4];       // res[j] = channel cs*4 + j of the 4 voxels
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const TapsLds t = make_taps_lds<POW2, CS>(px[k], py[k], res2, half_size, Hf, Wf, zidx);
                    const f4 a4 = *reinterpret_cast<const f4 *>(img + t.a[0]);
                    const f4 b4 = *reinterpret_cast<const f4 *>(img + t.a[1]);
                    const f4 c4 = *reinterpret_cast<const f4 *>(img + t.a[2]);
                    const f4 d4 = *reinterpret_cast<const f4 *>(img + t.a[3]);
                    const f4 r = a4 * t.wt[0] + b4 * t.wt[1] + c4 * t.wt[2] + d4 * t.wt[3];     // 4 channels of voxel k
#pragma unroll
                    for (int j = 0; j < 4; ++j) res[j][k] = r[j];
                }
                float *o = reinterpret_cast<float *>(out_) + (n * 2 * F + (int64_t)side * F + cs * 4) * V + v;
#pragma unroll
                for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(res[j], reinterpret_cast<f4 *>(o + (int64_t)j * V));
            } else {
                const TapsLds t = make_taps_lds<POW2, CS>(px[0], py[0], res2, half_size, Hf, Wf, zidx);
                h8 res, res_lo;
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const f4 a4 = *reinterpret_cast<const f4 *>(img + t.a[0] + 4 * hq);
                    const f4 b4 = *reinterpret_cast<const f4 *>(img + t.a[1] + 4 * hq);
                    const f4 c4 = *reinterpret_cast<const f4 *>(img + t.a[2] + 4 * hq);
                    const f4 d4 = *reinterpret_cast<const f4 *>(img + t.a[3] + 4 * hq);
                    const f4 r = a4 * t.wt[0] + b4 * t.wt[1] + c4 * t.wt[2] + d4 * t.wt[3];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (OUT == 2) {      // as snvc_f16x3_from_ncdhw: hi = rn(v * mul), lo = rn(v * mul - hi); a bilinear sample is
                            const float vv = r[j] * mul;      // a convex combination: |v| <= max|feature|, which is what mul was chosen for
                            const _Float16 hi = (_Float16)vv;
                            res[4 * hq + j] = hi;
                            res_lo[4 * hq + j] = (_Float16)(vv - (float)hi);
                        } else {
                            res[4 * hq + j] = (_Float16)r[j];
                        }
                    }
                }
                _Float16 *o = reinterpret_cast<_Float16 *>(out_) + n * out_bs + (((int64_t)side * (F / 8) + cs) * V + v) * 8;
                __builtin_nontemporal_store(res, reinterpret_cast<h8 *>(o));
                if constexpr (OUT == 2) __builtin_nontemporal_store(res_lo, reinterpret_cast<h8 *>(o + lo_off));
            }
        }
        __syncthreads();       // every tap of this camera is resolved before the slice is overwritten
    }
}

// True when x is a power of two whose reciprocal is a normal float (make_taps<POW2>).
inline bool pow2_res(float x, float &inv) {
    int e = 0;
    if (!(x > 0.0f) || std::frexp(x, &e) != 0.5f || e < -100 || e > 100) return false;
    inv = std::ldexp(1.0f, 1 - e);      // x = 0.5 * 2^e  ->  1 / x = 2^(1 - e)
    return true;
}

template <int CS, int OUT, int BS>
int launch_gather_lds(const float *ws, const float *l_pts, const float *r_pts, void *out, int64_t N, int64_t F, int64_t Hf,
                      int64_t Wf, int64_t V, int64_t run, float res_x, float res_y, hipStream_t st, const float *mul_dev = nullptr,
                      int64_t lo_off = 0, int64_t out_bs = -1) {
    if (out_bs < 0) out_bs = 2 * F * V;        // dense C8: [N][2F/8][V][8] halves (unused by the fp32 form)
    const int plane = (int)(Hf * Wf);
    const size_t lds = (size_t)plane * CS * 4 + (CS / 4) * 16;     // the slice + the zero slot
    const int nruns = (int)ceil_div<int64_t>(V, run);
    const dim3 grid((unsigned)(8 * ceil_div(nruns, 8) * (int)(F / CS)), 1, (unsigned)N);
    float ix = 0.0f, iy = 0.0f;
    if (pow2_res(res_x, ix) && pow2_res(res_y, iy)) {
        static std::atomic<unsigned> attr_done{0};
        if (allow_large_lds(reinterpret_cast<const void *>(&voxel_gather_fwd_lds<CS, OUT, BS, true>), (int)lds, attr_done))
            voxel_gather_fwd_lds<CS, OUT, BS, true><<<grid, BS, lds, st>>>(ws, l_pts, r_pts, out, (int)F, (int)Hf, (int)Wf, V, run,
                                                                         nruns, ix, iy, mul_dev, lo_off, out_bs);
    } else {
        static std::atomic<unsigned> attr_done{0};
        if (allow_large_lds(reinterpret_cast<const void *>(&voxel_gather_fwd_lds<CS, OUT, BS, false>), (int)lds, attr_done))
            voxel_gather_fwd_lds<CS, OUT, BS, false><<<grid, BS, lds, st>>>(ws, l_pts, r_pts, out, (int)F, (int)Hf, (int)Wf, V, run,
                                                                          nruns, res_x, res_y, mul_dev, lo_off, out_bs);
    }
    return 0;
}

// voxels per workgroup: enough workgroups to fill the chip twice over, runs as long as that allows (the refill of the
// slice is amortised over the run), whole 4096-voxel steps
inline int64_t gather_run_length(int64_t V, int64_t slices, int64_t N) {
    const int64_t want = ceil_div<int64_t>(4 * (int64_t)device_cu_count(), slices * N);     // workgroups along the voxel axis
    int64_t run = ceil_div<int64_t>(ceil_div<int64_t>(V, want), 4096) * 4096;
    return run < 4096 ? 4096 : run;
}

// aggregate="concat-atten" (snvc/models/vernier.py:341-344): the concatenated voxel features are multiplied by
// clamp(cosine_similarity(left half, right half, dim=channels), 0).  One thread per voxel: pass 1 accumulates the two
// squared norms, pass 2 the dot product of the normalised halves (torch >= 2.0: x / max(||x||, eps) . y / max(||y||, eps),
// eps = 1e-8), pass 3 scales the 2F channels in place.  Rows of one channel are coalesced across the wave.
__global__ void __launch_bounds__(256)
voxel_atten_scale_kernel(float *__restrict__ vox, int F, int64_t V) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float *p = vox + (int64_t)blockIdx.y * 2 * F * V + v;
    float nl = 0.0f, nr = 0.0f;
    for (int c = 0; c < F; ++c) {
        const float a = p[(int64_t)c * V], b = p[(int64_t)(F + c) * V];
        nl += a * a; nr += b * b;
    }
    const float dl = fmaxf(sqrtf(nl), 1e-8f), dr = fmaxf(sqrtf(nr), 1e-8f);
    float dot = 0.0f;
    for (int c = 0; c < F; ++c) dot += (p[(int64_t)c * V] / dl) * (p[(int64_t)(F + c) * V] / dr);
    const float w = dot > 0.0f ? dot : 0.0f;
    for (int c = 0; c < 2 * F; ++c) p[(int64_t)c * V] *= w;
}

// fp16-storage output (conv3d_f16.hip's C8 layout [N][2F/8][V][8]; BASELINE.json configs[4]): same taps, same
// separately rounded fp32 a*nw + b*ne + c*sw + d*se per channel as the kernels above, rounded to half once on
// the way out.  One voxel per thread: a wave stores 1 KB contiguous per channel group (64 voxels x 16 bytes), and
// the output stream is 4F + 16 bytes per voxel instead of 8F + 16.
__global__ void __launch_bounds__(256)
voxel_gather_fwd_cl_c8(const float *__restrict__ ws, const float *__restrict__ l_pts, const float *__restrict__ r_pts,
                       _Float16 *__restrict__ out, int F, int Hf, int Wf, int64_t V, float res_x, float res_y) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int64_t n = blockIdx.y;
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int64_t fp = (int64_t)F * Hf * Wf;
    const int G = F / 8;
    _Float16 *o = out + n * 2 * F * V + v * 8;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const float *pts = (side == 0 ? l_pts : r_pts) + n * 2 * V;
        const float *feat = ws + ((int64_t)side * gridDim.y + n) * fp;
        const Taps t = make_taps(pts[v], pts[V + v], res_x, res_y, Hf, Wf);
        const f4 *pa = reinterpret_cast<const f4 *>(feat + (int64_t)(t.off[0] < 0 ? 0 : t.off[0]) * F);
        const f4 *pb = reinterpret_cast<const f4 *>(feat + (int64_t)(t.off[1] < 0 ? 0 : t.off[1]) * F);
        const f4 *pc = reinterpret_cast<const f4 *>(feat + (int64_t)(t.off[2] < 0 ? 0 : t.off[2]) * F);
        const f4 *pd = reinterpret_cast<const f4 *>(feat + (int64_t)(t.off[3] < 0 ? 0 : t.off[3]) * F);
        const bool va = t.off[0] >= 0, vb = t.off[1] >= 0, vc = t.off[2] >= 0, vd = t.off[3] >= 0;
#pragma unroll 2
        for (int g = 0; g < G; ++g) {
            h8 res;
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const f4 a4 = pa[2 * g + hq], b4 = pb[2 * g + hq], c4 = pc[2 * g + hq], d4 = pd[2 * g + hq];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = va ? a4[j] : 0.0f, b = vb ? b4[j] : 0.0f, c = vc ? c4[j] : 0.0f, d = vd ? d4[j] : 0.0f;
                    res[4 * hq + j] = (_Float16)(a * t.wt[0] + b * t.wt[1] + c * t.wt[2] + d * t.wt[3]);
                }
            }
            __builtin_nontemporal_store(res, reinterpret_cast<h8 *>(o + ((int64_t)(side * G + g) * V) * 8));
        }
    }
}

__global__ void __launch_bounds__(256)
voxel_gather_bwd(const float *__restrict__ grad_out, const float *__restrict__ l_pts,
                 const float *__restrict__ r_pts, float *__restrict__ grad_left,
                 float *__restrict__ grad_right, int F, int Hf, int Wf, int64_t V, float res_x,
                 float res_y) {
    const int64_t n = blockIdx.y;
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float *lp = l_pts + n * 2 * V, *rp = r_pts + n * 2 * V;
    const Taps tl = make_taps(lp[v], lp[V + v], res_x, res_y, Hf, Wf);
    const Taps tr = make_taps(rp[v], rp[V + v], res_x, res_y, Hf, Wf);
    const int plane = Hf * Wf;
    const float *g = grad_out + n * 2 * F * V + v;
    float *gl = grad_left + n * F * plane, *gr = grad_right + n * F * plane;
    for (int c = 0; c < F; ++c) {
        const float gv = g[(int64_t)c * V];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tl.off[k] >= 0) atomicAdd(gl + (int64_t)c * plane + tl.off[k], gv * tl.wt[k]);
    }
    g += (int64_t)F * V;
    for (int c = 0; c < F; ++c) {
        const float gv = g[(int64_t)c * V];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tr.off[k] >= 0) atomicAdd(gr + (int64_t)c * plane + tr.off[k], gv * tr.wt[k]);
    }
}


// Backward, privatised: nearby voxels project onto the same few feature pixels, so plain global
// float atomics collide heavily (64 lanes -> 1 address on a coherent projection).  A workgroup
// therefore owns CPB channel planes of one (sample, camera) and a large run of voxels, accumulates
// into an LDS copy of those planes with LDS atomics, and flushes only the touched pixels to global
// memory with one atomic each.  Taps are recomputed per channel group (cheap ALU) instead of being
// stored.  Summation order is not deterministic (like grid_sampler_2d_backward on any GPU).
template <int CPB>
__global__ void __launch_bounds__(256)
voxel_gather_bwd_lds(const float *__restrict__ grad_out, const float *__restrict__ l_pts,
                     const float *__restrict__ r_pts, float *__restrict__ grad_left,
                     float *__restrict__ grad_right, int F, int Hf, int Wf, int64_t V, int64_t vchunk,
                     float res_x, float res_y) {
    extern __shared__ float acc[];   // [CPB][Hf*Wf]
    const int plane = Hf * Wf;
    const int64_t n = blockIdx.z >> 1;
    const int side = blockIdx.z & 1;
    const int c0 = blockIdx.y * CPB;
    for (int i = threadIdx.x; i < CPB * plane; i += blockDim.x) acc[i] = 0.0f;
    __syncthreads();
    const float *pts = (side == 0 ? l_pts : r_pts) + n * 2 * V;
    const float *g = grad_out + (n * 2 * F + (int64_t)side * F + c0) * V;
    const int64_t v0 = (int64_t)blockIdx.x * vchunk;
    const int64_t v1 = v0 + vchunk < V ? v0 + vchunk : V;
    for (int64_t v = v0 + threadIdx.x; v < v1; v += blockDim.x) {
        const Taps t = make_taps(pts[v], pts[V + v], res_x, res_y, Hf, Wf);
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            if (c0 + c >= F) break;
            const float gv = g[(int64_t)c * V + v];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t.off[k] >= 0) atomicAdd(&acc[c * plane + t.off[k]], gv * t.wt[k]);
        }
    }
    __syncthreads();
    float *dst = (side == 0 ? grad_left : grad_right) + (n * F + c0) * (int64_t)plane;
    for (int i = threadIdx.x; i < CPB * plane; i += blockDim.x) {
        const float a = acc[i];
        if (a != 0.0f && c0 + i / plane < F) atomicAdd(dst + i, a);
    }
}

}  // namespace
}  // namespace snvc

extern "C" {

int snvc_voxel_gather_forward(const float *left, const float *right, const float *l_pts,
                              const float *r_pts, float *out, int64_t N, int64_t F, int64_t Hf,
                              int64_t Wf, int64_t V, float res_x, float res_y, void *stream) {
    using namespace snvc;
    if (N < 0 || F < 0 || Hf < 0 || Wf < 0 || V < 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_forward: negative size");
    if (N == 0 || F == 0 || V == 0) return SNVC_OK;
    if (Hf * Wf >= (int64_t)1 << 30 || N > 65535)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_voxel_gather_forward: feature plane or batch too large");
    if (!left || !right || !l_pts || !r_pts || !out)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_forward: null pointer");
    dim3 grid((unsigned)ceil_div<int64_t>(V, 256), (unsigned)N);
    voxel_gather_fwd<<<grid, 256, 0, as_stream(stream)>>>(left, right, l_pts, r_pts, out, (int)F, (int)Hf, (int)Wf, V,
                                                          res_x, res_y);
    return check_launch("snvc_voxel_gather_forward");
}

int64_t snvc_voxel_gather_workspace_floats(int64_t N, int64_t F, int64_t Hf, int64_t Wf) {
    if (N < 0 || F < 0 || Hf < 0 || Wf < 0) return -1;
    return 2 * N * F * Hf * Wf;
}

int snvc_voxel_gather_forward_ws(const float *left, const float *right, const float *l_pts, const float *r_pts,
                                 float *out, float *workspace, int64_t N, int64_t F, int64_t Hf, int64_t Wf,
                                 int64_t V, float res_x, float res_y, void *stream) {
    using namespace snvc;
    const bool cl_ok = workspace && F > 0 && F % 4 == 0 && F <= 256 && N > 0 && V > 0 && N <= 65535 &&
                       Hf * Wf < ((int64_t)1 << 24) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
    if (!cl_ok) return snvc_voxel_gather_forward(left, right, l_pts, r_pts, out, N, F, Hf, Wf, V, res_x, res_y, stream);
    if (!left || !right || !l_pts || !r_pts || !out)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_forward_ws: null pointer");
    const int plane = (int)(Hf * Wf);
    dim3 tg((unsigned)ceil_div(plane, 64), (unsigned)N, 2);
    features_to_channels_last<<<tg, 256, (size_t)64 * (F + 1) * sizeof(float), as_stream(stream)>>>(left, right, workspace,
                                                                                                    (int)F, plane);
    const bool x4 = V % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(l_pts) | reinterpret_cast<uintptr_t>(r_pts) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (x4 && (int64_t)plane * 16 <= 72 * 1024 && V >= 4096) {     // LDS-staged 4-channel slices (two workgroups per CU)
        const int64_t run = gather_run_length(V, F / 4, N);
        launch_gather_lds<4, false, 512>(workspace, l_pts, r_pts, out, N, F, Hf, Wf, V, run, res_x, res_y, as_stream(stream));
    } else if (x4) {
        dim3 grid((unsigned)ceil_div<int64_t>(V / 4, 256), (unsigned)N);
        voxel_gather_fwd_cl_x4<<<grid, 256, 0, as_stream(stream)>>>(workspace, l_pts, r_pts, out, (int)F, (int)Hf, (int)Wf,
                                                                    V, res_x, res_y);
    } else {
        dim3 grid((unsigned)ceil_div<int64_t>(V, 256), (unsigned)N);
        voxel_gather_fwd_cl<<<grid, 256, 0, as_stream(stream)>>>(workspace, l_pts, r_pts, out, (int)F, (int)Hf, (int)Wf, V,
                                                                 res_x, res_y);
    }
    return check_launch("snvc_voxel_gather_forward_ws");
}

int snvc_voxel_gather_forward_f16(const float *left, const float *right, const float *l_pts, const float *r_pts,
                                  void *out, float *workspace, int64_t N, int64_t F, int64_t Hf, int64_t Wf,
                                  int64_t V, float res_x, float res_y, void *stream) {
    using namespace snvc;
    if (N < 0 || F <= 0 || F % 8 != 0 || F > 256 || Hf <= 0 || Wf <= 0 || V < 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_forward_f16: bad sizes (F % 8 == 0, F <= 256)");
    if (N == 0 || V == 0) return SNVC_OK;
    if (N > 65535 || Hf * Wf >= ((int64_t)1 << 24))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_voxel_gather_forward_f16: feature plane or batch too large");
    if (!left || !right || !l_pts || !r_pts || !out || !workspace)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_forward_f16: null pointer");
    if ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(out)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_forward_f16: workspace and output must be 16-byte aligned");
    const int plane = (int)(Hf * Wf);
    dim3 tg((unsigned)ceil_div(plane, 64), (unsigned)N, 2);
    features_to_channels_last<<<tg, 256, (size_t)64 * (F + 1) * sizeof(float), as_stream(stream)>>>(left, right, workspace,
                                                                                                    (int)F, plane);
    if ((int64_t)plane * 32 <= 144 * 1024 && V >= 4096) {      // LDS-staged 8-channel groups (one workgroup per CU)
        const int64_t run = gather_run_length(V, F / 8, N);
        launch_gather_lds<8, 1, 1024>(workspace, l_pts, r_pts, out, N, F, Hf, Wf, V, run, res_x, res_y, as_stream(stream));
    } else {
        dim3 grid((unsigned)ceil_div<int64_t>(V, 256), (unsigned)N);
        voxel_gather_fwd_cl_c8<<<grid, 256, 0, as_stream(stream)>>>(workspace, l_pts, r_pts, reinterpret_cast<_Float16 *>(out),
                                                                    (int)F, (int)Hf, (int)Wf, V, res_x, res_y);
    }
    return check_launch("snvc_voxel_gather_forward_f16");
}

int snvc_voxel_gather_forward_split(const float *left, const float *right, const float *l_pts, const float *r_pts, void *out_hi,
                                    void *out_lo, const float *mul_dev, float *workspace, int64_t N, int64_t F, int64_t Hf, int64_t Wf,
                                    int64_t V, int64_t out_batch_stride, float res_x, float res_y, void *stream) {
    using namespace snvc;
    if (N < 0 || F <= 0 || F % 8 != 0 || F > 256 || Hf <= 0 || Wf <= 0 || V < 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_forward_split: bad sizes (F % 8 == 0, F <= 256)");
    if (N == 0 || V == 0) return SNVC_OK;
    if (N > 65535 || Hf * Wf >= ((int64_t)1 << 24))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_voxel_gather_forward_split: feature plane or batch too large");
    if (!left || !right || !l_pts || !r_pts || !out_hi || !out_lo || !mul_dev || !workspace)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_forward_split: null pointer");
    if ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(out_hi) | reinterpret_cast<uintptr_t>(out_lo)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_forward_split: workspace and outputs must be 16-byte aligned");
    const int plane = (int)(Hf * Wf);
    if (!((int64_t)plane * 32 <= 144 * 1024 && V >= 4096))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_voxel_gather_forward_split: needs the LDS-staged form (Hf * Wf <= 4608, V >= 4096)");
    dim3 tg((unsigned)ceil_div(plane, 64), (unsigned)N, 2);
    features_to_channels_last<<<tg, 256, (size_t)64 * (F + 1) * sizeof(float), as_stream(stream)>>>(left, right, workspace,
                                                                                                    (int)F, plane);
    const int64_t run = gather_run_length(V, F / 8, N);
    const int64_t lo_off = static_cast<const _Float16 *>(out_lo) - static_cast<const _Float16 *>(out_hi);
    launch_gather_lds<8, 2, 1024>(workspace, l_pts, r_pts, out_hi, N, F, Hf, Wf, V, run, res_x, res_y, as_stream(stream), mul_dev, lo_off,
                                  out_batch_stride ? out_batch_stride : 4 * F * V);
    return check_launch("snvc_voxel_gather_forward_split");
}

int snvc_voxel_atten_scale(float *vox, int64_t N, int64_t F, int64_t V, void *stream) {
    using namespace snvc;
    if (N < 0 || F < 0 || V < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_atten_scale: negative size");
    if (N == 0 || F == 0 || V == 0) return SNVC_OK;
    if (!vox) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_atten_scale: null pointer");
    if (N > 65535 || ceil_div<int64_t>(V, 256) >= ((int64_t)1 << 31)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_voxel_atten_scale: too large");
    voxel_atten_scale_kernel<<<dim3((unsigned)ceil_div<int64_t>(V, 256), (unsigned)N), 256, 0, as_stream(stream)>>>(vox, (int)F, V);
    return check_launch("snvc_voxel_atten_scale");
}

int snvc_voxel_gather_backward(const float *grad_out, const float *l_pts, const float *r_pts,
                               float *grad_left, float *grad_right, int64_t N, int64_t F, int64_t Hf,
                               int64_t Wf, int64_t V, float res_x, float res_y, void *stream) {
    using namespace snvc;
    if (N < 0 || F < 0 || Hf < 0 || Wf < 0 || V < 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_backward: negative size");
    if (Hf * Wf >= (int64_t)1 << 30 || N > 65535)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_voxel_gather_backward: feature plane or batch too large");
    const size_t bytes = sizeof(float) * (size_t)(N * F * Hf * Wf);
    if (bytes) {
        if (!grad_left || !grad_right)
            return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_backward: null pointer");
        if (hipMemsetAsync(grad_left, 0, bytes, as_stream(stream)) != hipSuccess ||
            hipMemsetAsync(grad_right, 0, bytes, as_stream(stream)) != hipSuccess)
            return fail(SNVC_ERR_HIP, "snvc_voxel_gather_backward: hipMemsetAsync failed");
    }
    if (N == 0 || F == 0 || V == 0) return SNVC_OK;
    constexpr int CPB = 2;
    if ((int64_t)CPB * Hf * Wf * 4 <= 64 * 1024 && 2 * N <= 65535) {
        // ~64k voxels per workgroup: the flush (<= CPB*Hf*Wf atomics) is amortised over >= 8x as many LDS adds
        int64_t vchunk = 65536;
        if (vchunk > V) vchunk = V;
        dim3 grid((unsigned)ceil_div<int64_t>(V, vchunk), (unsigned)ceil_div<int64_t>(F, CPB), (unsigned)(2 * N));
        voxel_gather_bwd_lds<CPB><<<grid, 256, (size_t)CPB * Hf * Wf * 4, as_stream(stream)>>>(
            grad_out, l_pts, r_pts, grad_left, grad_right, (int)F, (int)Hf, (int)Wf, V, vchunk, res_x, res_y);
    } else {
        dim3 grid((unsigned)ceil_div<int64_t>(V, 256), (unsigned)N);
        voxel_gather_bwd<<<grid, 256, 0, as_stream(stream)>>>(grad_out, l_pts, r_pts, grad_left, grad_right,
                                                              (int)F, (int)Hf, (int)Wf, V, res_x, res_y);
    }
    return check_launch("snvc_voxel_gather_backward");
}

}  // extern "C"
