// Weight gradient of the 3D convolutions for gfx950 (training step, BASELINE.json configs[3]).
//
// Reference: torch autograd through nn.Conv3d / nn.ConvTranspose3d (cuDNN wgrad) as composed by
// convbn_3d / hourglass (snvc/models/submodule.py:32-50,85-168).
//
//   dW[cg][cx][t] = sum_n sum_o  G[n,cg,o] * X[n,cx, o*stride - pad + t*dil]
//
// G lives on the SMALL grid (the conv's output gradient), X on the BIG grid (the conv's input);
// for a ConvTranspose3d the caller swaps roles (G := the deconv's input, X := its output
// gradient) and the result is directly in nn.ConvTranspose3d's [Cin][Cout][k^3] layout.
//
// GEMM view per tap: M = 32 channels of G, N = 32 channels of X, K = voxels.  With
// v_mfma_f32_32x32x2_f32 the lane index runs over CHANNELS for both operands and the two
// K-slots are two adjacent voxels, so LDS tiles are kept channel-major with an ODD row stride:
// the 32 lanes of a half-wave read one voxel of 32 different channels from 32 different banks.
// A workgroup owns one (G-channel block, X-channel block) pair and one spatial partition; its 4
// waves split the (up to 28) taps, each wave keeping 7 accumulators of 32x32 in registers across
// ALL tiles of the partition.  Partials go to a slab [partition][...] that a second kernel sums
// in a fixed order: the result is deterministic (no float atomics).
#include <algorithm>
#include <cstdlib>

#include "common.hpp"


namespace snvc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
    const float *x;   // big grid  [N, Cx, Di, Hi, Wi]
    const float *g;   // small grid [N, Cg, Do, Ho, Wo]
    float *partial;   // [P][pairs][taps_pad][32][32]
    int N, Cx, Di, Hi, Wi;
    int Cg, Do, Ho, Wo;
    int tiles_h, tiles_w;       // tiles per (n, od) plane
    int64_t ntiles;             // N * Do * tiles_h * tiles_w
    int P;                      // spatial partitions (gridDim.x)
    int cx_blocks;              // pairs = cg_blocks * cx_blocks, blockIdx.y = cgb * cx_blocks + cxb
    int64_t x_bs, g_bs;
    int vec;   // 16-byte aligned rows on both grids: float4 staging with register prefetch
    int pairs, upx;     // Winograd form: 1-D grid of 8 * upx * 3 workgroups, see conv3d_wgrad_wino_kernel
};

// A workgroup covers a CHUNK of taps: KDG kernel depth-slices x KHG kernel rows x all KS columns
// (<= 28 taps = 7 per wave); blockIdx.z enumerates the chunks, so 5^3 and 7^3 kernels only widen
// the grid, not the register or LDS footprint.
// KSPLIT_: the 4 waves split the K-steps (voxel pairs) of a tile and each keeps ALL taps of the chunk (<= 10);
// otherwise they split the chunk's taps (<= 7 each).  K-splitting has no padded tap slots (9 taps over 4 waves
// would issue 12 MFMAs per K-step) and ends with a fixed-order reduction of the four waves through LDS.
template <int KS_, int STRIDE_, int DIL_, int TH_, int KDG_, int KHG_, bool KSPLIT_ = false>
struct WgradCfg {
    static constexpr bool KSPLIT = KSPLIT_;
    static constexpr int KS = KS_, STRIDE = STRIDE_, DIL = DIL_, KDG = KDG_, KHG = KHG_;
    static constexpr int PAD = DIL * (KS - 1) / 2;
    static constexpr int TAPS = KS * KS * KS;
    static constexpr int CHUNK_TAPS = KDG * KHG * KS;
    static constexpr int NT = KSPLIT_ ? CHUNK_TAPS : (CHUNK_TAPS + 3) / 4;    // taps per wave
    static constexpr int CH_D = KS / KDG, CH_H = (KS + KHG - 1) / KHG;   // chunks along kd / kh
    static constexpr int TH = TH_, TW = 32;            // output tile: 1 x TH x 32 voxels = 32*TH K-slots
    static constexpr int IN_D = (KDG - 1) * DIL + 1;
    static constexpr int IN_H = (TH - 1) * STRIDE + (KHG - 1) * DIL + 1;
    static_assert(KS % KDG == 0, "depth chunks must tile the kernel");
    static constexpr int IN_W = (TW - 1) * STRIDE + (KS - 1) * DIL + 1;
    static constexpr int XV = IN_D * IN_H * IN_W;      // staged X voxels per channel
    static constexpr int XS = XV | 1;                  // odd row stride
    static constexpr int GV = TH * TW;
    static constexpr int GS = GV + 1;                  // odd
    // vectorised staging (rows widened to 16-byte aligned global columns, like the forward Stager)
    static constexpr int LPAD = 4, XOFF = LPAD - PAD;
    static constexpr int IN_WV = (XOFF + IN_W + 3) / 4 * 4;
    static constexpr int XVV = IN_D * IN_H * IN_WV, XSV = XVV | 1;
    static constexpr int RQ = IN_WV / 4, XITEMS = 32 * IN_D * IN_H * RQ, XNIT = (XITEMS + 255) / 256;
    static constexpr int GITEMS = 32 * TH * 8, GNIT = (GITEMS + 255) / 256;
    static constexpr int LDS_FLOATS = 32 * (XSV > XS ? XSV : XS) + 32 * GS;
    static_assert(NT <= (KSPLIT_ ? 10 : 7), "accumulators per wave");
    static_assert(PAD <= LPAD, "left halo fits the aligned margin");
};

template <class Cfg, int VEC>   // 0: scalar staging; 4 / 2: float4 X pieces and 16- / 8-byte G pieces (rows with Wo % 4 == 2)
__global__ void __launch_bounds__(256, 2)
conv3d_wgrad_kernel(const WgradArgs a) {
    constexpr int KS = Cfg::KS, S = Cfg::STRIDE, DIL = Cfg::DIL, PAD = Cfg::PAD, NT = Cfg::NT;
    constexpr int KDG = Cfg::KDG, KHG = Cfg::KHG;
    constexpr int IN_H = Cfg::IN_H, IN_W = Cfg::IN_W, XV = Cfg::XV, XS = Cfg::XS, GV = Cfg::GV, GS = Cfg::GS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xl = lds;             // [32][XS]
    float *gl = lds + 32 * XS;   // [32][GS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cgb = blockIdx.y / a.cx_blocks, cxb = blockIdx.y - cgb * a.cx_blocks;
    const int cg0 = cgb * 32, cx0 = cxb * 32;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // this workgroup's tap chunk: kernel depth-slices [kd0, kd0+KDG), rows [kh0, kh0+khn), all columns
    const int kd0 = (blockIdx.z / Cfg::CH_H) * KDG, kh0 = (blockIdx.z % Cfg::CH_H) * KHG;
    const int khn = KS - kh0 < KHG ? KS - kh0 : KHG;
    const int chunk_taps = KDG * khn * KS;
    // per-wave tap table: local tap = wave + 4*t -> LDS offset of its window inside the staged X tile
    int tap_off[NT], tap_glob[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int lt = Cfg::KSPLIT ? t : wave + 4 * t;
        const int kw = lt % KS, kh = (lt / KS) % khn, kd = lt / (KS * khn);
        const bool ok = lt < chunk_taps;
        tap_off[t] = ok ? (kd * DIL * IN_H + kh * DIL) * (VEC != 0 ? Cfg::IN_WV : IN_W) + kw * DIL : 0;
        tap_glob[t] = ok ? ((kd0 + kd) * KS + kh0 + kh) * KS + kw : -1;
    }

    const int64_t in_hw = (int64_t)a.Hi * a.Wi, in_dhw = in_hw * a.Di;
    const int64_t out_hw = (int64_t)a.Ho * a.Wo, out_dhw = out_hw * a.Do;
    const int ch = lane & 31, half = lane >> 5;
    const float *xrow = xl + ch * XS, *grow = gl + ch * GS;

    if constexpr (VEC != 0) {
        // ---- float4 staging with register prefetch: the element -> (channel, row, piece) decomposition of a
        // staging item does not depend on the tile, so it is done once; per tile only the three range tests
        // remain.  Tile t+1 is in registers while tile t is multiplied; LDS is single-buffered (70 KB).
        constexpr int IN_WV = Cfg::IN_WV, XSV = Cfg::XSV, RQ = Cfg::RQ, XNIT = Cfg::XNIT;
        constexpr int GP = VEC, GQ = 32 / GP, GITEMS = 32 * Cfg::TH * GQ, GNIT = (GITEMS + 255) / 256;   // G pieces of GP floats
        typedef float GVec __attribute__((ext_vector_type(GP)));
        constexpr int ROWS = Cfg::IN_D * IN_H;
        float *glv = lds + 32 * XSV;
        const float *xrow_v = xl + ch * XSV, *grow_v = glv + ch * GS;
        // one packed code per staging item: dd | hh << 4 | q << 8 | c << 16  (-1: item beyond the tile / channels)
        int xcode[XNIT], gcode[GNIT];
#pragma unroll
        for (int it = 0; it < XNIT; ++it) {
            const int i = it * 256 + tid;
            const int c = i / (ROWS * RQ), r = i - c * (ROWS * RQ);
            const int row = r / RQ, q = r - row * RQ;
            const int dd = row / IN_H, hh = row - dd * IN_H;
            xcode[it] = (i < Cfg::XITEMS && cx0 + c < a.Cx) ? (dd | (hh << 4) | (q << 8) | (c << 16)) : -1;
        }
#pragma unroll
        for (int it = 0; it < GNIT; ++it) {
            const int i = it * 256 + tid;
            const int c = i / (Cfg::TH * GQ), r = i - c * (Cfg::TH * GQ);
            const int hh = r / GQ, q = r - hh * GQ;
            gcode[it] = (i < GITEMS && cg0 + c < a.Cg) ? (hh | (q << 8) | (c << 16)) : -1;
        }
        static_assert(Cfg::IN_D <= 16 && IN_H <= 16 && RQ <= 256, "packed staging code");
        f32x4 xv[XNIT];
        GVec gv[GNIT];
        unsigned xok = 0, gok = 0;
        auto load_tile = [&](int64_t tile) {
            const int tw = (int)(tile % a.tiles_w);
            const int th = (int)((tile / a.tiles_w) % a.tiles_h);
            const int od = (int)((tile / ((int64_t)a.tiles_w * a.tiles_h)) % a.Do);
            const int64_t n = tile / ((int64_t)a.tiles_w * a.tiles_h * a.Do);
            const int oh0 = th * Cfg::TH, ow0 = tw * 32;
            const int id0 = od * S - PAD + kd0 * DIL, ih0 = oh0 * S - PAD + kh0 * DIL, ix0 = ow0 * S - Cfg::LPAD;
            const float *xn = a.x + n * a.x_bs + (int64_t)cx0 * in_dhw;
            const float *gn = a.g + n * a.g_bs + (int64_t)cg0 * out_dhw;
            const int64_t xorg = (int64_t)id0 * in_hw + (int64_t)ih0 * a.Wi + ix0;    // may be negative; used only when valid
            const int64_t gorg = (int64_t)od * out_hw + (int64_t)oh0 * a.Wo + ow0;
            xok = 0; gok = 0;
#pragma unroll
            for (int it = 0; it < XNIT; ++it) {
                const int dd = xcode[it] & 15, hh = (xcode[it] >> 4) & 15, q = (xcode[it] >> 8) & 255, c = xcode[it] >> 16;
                const bool ok = xcode[it] >= 0 && (unsigned)(id0 + dd) < (unsigned)a.Di &&
                                (unsigned)(ih0 + hh) < (unsigned)a.Hi && (unsigned)(ix0 + 4 * q) < (unsigned)a.Wi;
                const int64_t rel = c * in_dhw + dd * in_hw + (int64_t)hh * a.Wi + 4 * q;
                xv[it] = *reinterpret_cast<const f32x4 *>(ok ? xn + xorg + rel : a.x);
                xok |= (ok ? 1u : 0u) << it;
            }
#pragma unroll
            for (int it = 0; it < GNIT; ++it) {
                const int hh = gcode[it] & 255, q = (gcode[it] >> 8) & 255, c = gcode[it] >> 16;
                const bool ok = gcode[it] >= 0 && oh0 + hh < a.Ho && ow0 + GP * q < a.Wo;
                gv[it] = *reinterpret_cast<const GVec *>(ok ? gn + gorg + c * out_dhw + (int64_t)hh * a.Wo + GP * q : a.g);
                gok |= (ok ? 1u : 0u) << it;
            }
        };
        auto store_tile = [&]() {
#pragma unroll
            for (int it = 0; it < XNIT; ++it) {
                if (Cfg::XITEMS % 256 != 0 && it * 256 + tid >= Cfg::XITEMS) continue;
                const bool ok = (xok >> it) & 1u;
                const int dd = xcode[it] & 15, hh = (xcode[it] >> 4) & 15, q = (xcode[it] >> 8) & 255, c = (xcode[it] >> 16) & 31;
                const int dst = c * XSV + (dd * IN_H + hh) * IN_WV + 4 * q;
#pragma unroll
                for (int j = 0; j < 4; ++j) xl[dst + j] = ok ? xv[it][j] : 0.0f;
            }
#pragma unroll
            for (int it = 0; it < GNIT; ++it) {
                if (GITEMS % 256 != 0 && it * 256 + tid >= GITEMS) continue;
                const bool ok = (gok >> it) & 1u;
                const int hh = gcode[it] & 255, q = (gcode[it] >> 8) & 255, c = (gcode[it] >> 16) & 31;
                const int dst = c * GS + hh * 32 + GP * q;
#pragma unroll
                for (int j = 0; j < GP; ++j) glv[dst + j] = ok ? gv[it][j] : 0.0f;
            }
        };
        int64_t tile = blockIdx.x;
        if (tile < a.ntiles) load_tile(tile);
        for (; tile < a.ntiles; tile += a.P) {
            __syncthreads();   // previous tile fully consumed
            store_tile();
            __syncthreads();
            if (tile + a.P < a.ntiles) load_tile(tile + a.P);   // in flight during the MFMAs below
            for (int kk = Cfg::KSPLIT ? wave : 0; kk < GV / 2; kk += Cfg::KSPLIT ? 4 : 1) {
                const int v = 2 * kk + half;
                const int hh = v >> 5, j = v & 31;
                const float af = grow_v[v];
                const int xbase = (hh * S) * IN_WV + j * S + Cfg::XOFF;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float bf = xrow_v[xbase + tap_off[t]];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[t], 0, 0, 0);
                }
            }
        }
    } else {
    for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += a.P) {
        const int tw = (int)(tile % a.tiles_w);
        const int th = (int)((tile / a.tiles_w) % a.tiles_h);
        const int od = (int)((tile / ((int64_t)a.tiles_w * a.tiles_h)) % a.Do);
        const int64_t n = tile / ((int64_t)a.tiles_w * a.tiles_h * a.Do);
        const int oh0 = th * Cfg::TH, ow0 = tw * 32;
        const int id0 = od * S - PAD + kd0 * DIL, ih0 = oh0 * S - PAD + kh0 * DIL, iw0 = ow0 * S - PAD;
        const float *xn = a.x + n * a.x_bs + (int64_t)cx0 * in_dhw;
        const float *gn = a.g + n * a.g_bs + (int64_t)cg0 * out_dhw;
        __syncthreads();   // previous tile fully consumed
        // ---- stage X: 32 channels x XV voxels (zero padded), branch-free
        for (int e0 = 0; e0 < 32 * XV; e0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int e = e0 + b * 256 + tid;
                const int c = e / XV, rem = e - c * XV;
                const int dd = rem / (IN_H * IN_W), rem2 = rem - dd * (IN_H * IN_W);
                const int hh = rem2 / IN_W, ww = rem2 - hh * IN_W;
                const int gd = id0 + dd, gh = ih0 + hh, gw = iw0 + ww;
                const bool ok = e < 32 * XV && cx0 + c < a.Cx && (unsigned)gd < (unsigned)a.Di &&
                                (unsigned)gh < (unsigned)a.Hi && (unsigned)gw < (unsigned)a.Wi;
                const float t = xn[ok ? (c * in_dhw + gd * in_hw + (int64_t)gh * a.Wi + gw) : 0];
                v[b] = ok ? t : 0.0f;
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int e = e0 + b * 256 + tid;
                if (e < 32 * XV) { const int c = e / XV; xl[c * XS + (e - c * XV)] = v[b]; }
            }
        }
        // ---- stage G: 32 channels x GV voxels
#pragma unroll
        for (int b = 0; b < (32 * GV) / 256; ++b) {
            const int e = b * 256 + tid;
            const int c = e / GV, rem = e - c * GV;
            const int hh = rem / 32, ww = rem - hh * 32;
            const int gh = oh0 + hh, gw = ow0 + ww;
            const bool ok = cg0 + c < a.Cg && gh < a.Ho && gw < a.Wo;
            const float t = gn[ok ? (c * out_dhw + od * out_hw + (int64_t)gh * a.Wo + gw) : 0];
            gl[c * GS + rem] = ok ? t : 0.0f;
        }
        __syncthreads();
        // ---- 32 K-steps (2 voxels each) x NT taps
#pragma unroll 4
        for (int kk = 0; kk < GV / 2; ++kk) {
            const int v = 2 * kk + half;           // output voxel inside the tile: hh = v / 32, j = v % 32
            const int hh = v >> 5, j = v & 31;
            const float af = grow[v];
            const int xbase = (hh * S) * IN_W + j * S;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float bf = xrow[xbase + tap_off[t]];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[t], 0, 0, 0);
            }
        }
    }

    }

    // ---- partial slab: [p][pair][tap (global index)][cg 32][cx 32]
    float *pp = a.partial + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (int64_t)Cfg::TAPS * 1024;
    if constexpr (Cfg::KSPLIT) {
        // the four waves hold partial sums of the SAME taps: add them in wave order through LDS (deterministic)
        constexpr int TG = Cfg::LDS_FLOATS / 1024 < NT ? Cfg::LDS_FLOATS / 1024 : NT;   // taps per reduction round
        static_assert(TG >= 1, "reduction buffer");
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += TG) {
            __syncthreads();   // staging LDS / previous round no longer read
            for (int w = 0; w < 4; ++w) {
                if (wave == w) {
#pragma unroll
                    for (int t = t0; t < t0 + TG && t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float *slot = lds + ((t - t0) * 16 + r) * 64 + lane;
                            *slot = w == 0 ? acc[t][r] : *slot + acc[t][r];
                        }
                }
                __syncthreads();
            }
#pragma unroll
            for (int t = t0; t < t0 + TG && t < NT; ++t) {
                if (tap_glob[t] < 0) continue;
                for (int i = tid; i < 1024; i += 256) {
                    const int r = i >> 6, l = i & 63;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    pp[((int64_t)tap_glob[t] * 32 + row) * 32 + (l & 31)] = lds[(t - t0) * 1024 + i];
                }
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (tap_glob[t] < 0) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;           // cg
                pp[((int64_t)tap_glob[t] * 32 + row) * 32 + ch] = acc[t][r];  // col = cx = lane & 31
            }
        }
    }
}

// dw[cg][cx][tap] = sum_p partial[p][pair][tap][cg%32][cx%32].  One thread per element of a slab, in the slab's own
// order (consecutive lanes read consecutive floats of every partition: the first version walked dw's order and
// touched one cache line per lane and partition, 0.25 ms per layer); four running sums over p mod 4, combined in a
// fixed order: deterministic.
__global__ void wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dw, int Cg, int Cx,
                                    int taps, int taps_pad, int cx_blocks, int pairs, int P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pstride = (int64_t)pairs * taps_pad * 1024;
    if (i >= pstride) return;
    const int cxl = (int)(i & 31), cgl = (int)((i >> 5) & 31);
    const int tap = (int)((i >> 10) % taps_pad);
    const int pair = (int)(i / ((int64_t)taps_pad * 1024));
    const int cg = (pair / cx_blocks) * 32 + cgl, cx = (pair % cx_blocks) * 32 + cxl;
    if (tap >= taps || cg >= Cg || cx >= Cx) return;
    const float *src = partial + i;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int p = 0;
    for (; p + 4 <= P; p += 4) {
        s0 += src[(int64_t)p * pstride];
        s1 += src[(int64_t)(p + 1) * pstride];
        s2 += src[(int64_t)(p + 2) * pstride];
        s3 += src[(int64_t)(p + 3) * pstride];
    }
    for (; p < P; ++p) s0 += src[(int64_t)p * pstride];
    dw[((int64_t)cg * Cx + cx) * taps + tap] = (s0 + s1) + (s2 + s3);
}


// Weight gradient of a 1x1x1 convolution to <= 2 channels (the global model's classifier, Conv3d(C,1,1)):
// dW[cg][cx] = sum over (n, voxels) of g[cg] * x[cx] -- C dot products over the whole grid, HBM-bound.  The MFMA kernel
// above spends 0.8 ms on it at cfg2 (one 32x32 tile for a 1x32 result); here a workgroup owns (chunk of voxels, cx),
// threads stride the chunk with float4 loads, a fixed-shape LDS tree adds the 256 partials, and a second kernel adds
// the chunks in ascending order: deterministic.
template <int CG>
__global__ void __launch_bounds__(256)
wgrad_k1_small_partial(const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ partial, int Cx, int64_t S4,
                       int64_t chunk4, int64_t x_bs, int64_t g_bs) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    // r6: FOUR input channels per workgroup -- the gradient's chunk is read once per four x chunks instead of once per channel
    // (1.47 GB of loads for 0.76 GB of tensors before), five loads in flight per thread instead of two; each channel's sum keeps its
    // order (thread-strided, then the fixed LDS tree): the same bits as the one-channel form
    constexpr int CXB = 4;
    __shared__ float red[CXB * CG][256];
    const int cx0 = blockIdx.y * CXB;
    const int64_t n = blockIdx.z;
    const int64_t v0 = (int64_t)blockIdx.x * chunk4, v1 = v0 + chunk4 < S4 ? v0 + chunk4 : S4;
    const f4 *xp[CXB];
#pragma unroll
    for (int j = 0; j < CXB; ++j) xp[j] = reinterpret_cast<const f4 *>(x + n * x_bs + (int64_t)(cx0 + j < Cx ? cx0 + j : Cx - 1) * S4 * 4);
    const f4 *gp = reinterpret_cast<const f4 *>(g + n * g_bs);
    float acc[CXB][CG];
#pragma unroll
    for (int j = 0; j < CXB; ++j)
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[j][c] = 0.0f;
    for (int64_t v = v0 + threadIdx.x; v < v1; v += 256) {
        f4 xv[CXB], gv[CG];
#pragma unroll
        for (int j = 0; j < CXB; ++j) xv[j] = xp[j][v];
#pragma unroll
        for (int c = 0; c < CG; ++c) gv[c] = gp[(int64_t)c * S4 + v];
#pragma unroll
        for (int j = 0; j < CXB; ++j)
#pragma unroll
            for (int c = 0; c < CG; ++c)
                acc[j][c] += (xv[j][0] * gv[c][0] + xv[j][1] * gv[c][1]) + (xv[j][2] * gv[c][2] + xv[j][3] * gv[c][3]);
    }
#pragma unroll
    for (int j = 0; j < CXB; ++j)
#pragma unroll
        for (int c = 0; c < CG; ++c) red[j * CG + c][threadIdx.x] = acc[j][c];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
#pragma unroll
            for (int c = 0; c < CXB * CG; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < CXB * CG) {
        const int j = threadIdx.x / CG, c = threadIdx.x - j * CG;
        if (cx0 + j < Cx) partial[(((int64_t)n * gridDim.x + blockIdx.x) * Cx + cx0 + j) * CG + c] = red[threadIdx.x][0];
    }
}

__global__ void wgrad_k1_small_final(const float *__restrict__ partial, float *__restrict__ dw, int Cx, int CG, int64_t slabs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (cx, cg)
    if (i >= Cx * CG) return;
    const int cx = i / CG, cg = i - cx * CG;
    float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};          // four chains (slab k to chain k % 4), fixed fold: deterministic, four loads in flight
    int64_t k = 0;
    for (; k + 4 <= slabs; k += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s4[j] += partial[((k + j) * Cx + cx) * CG + cg];
    }
    for (; k < slabs; ++k) s4[0] += partial[(k * Cx + cx) * CG + cg];
    dw[(int64_t)cg * Cx + cx] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// ------------------------------------------------------------------------------------ Winograd-domain weight gradient
// 3x3x3 / stride-1 layers (conv1, conv2, the hourglass's stride-1 layers: 80 % of the training step's wgrad FLOPs).
// The forward runs F(4,3) along W:  y = A^T [ (G g) . (B^T d) ]  per quad of 4 outputs.  Its adjoint w.r.t. the taps is
//     dU_p = sum over quads of (A dy)_p * (B^T d)_p        (6 products per quad and (cout, cin, kd, kh))
//     dg   = G^T dU                                         (applied once, in the reduce kernel)
// so a K-step of two quads (8 output voxels) costs 6 MFMAs per (kd, kh) where the direct form issues 12 (3 taps x 4
// K-steps).  Both operands are transformed in registers right after their LDS reads (8 + 12 VALU per 6 MFMAs):
//     (A dy): p0 = dy0, p1/p2 = (dy0+dy2) +- (dy1+dy3), p3/p4 = (dy0+4dy2) +- (2dy1+8dy3), p5 = dy3
//     (B^T d): as in the forward kernel (conv3d.hip, 4.1c)
// A workgroup (12 waves = the three waves per SIMD that 168 registers allow; a 6-wave workgroup lands 2,2,1,1 on the SIMDs
// and a second one does not fit beside it -- measured: same time with one workgroup per CU forced) owns one kernel depth
// slice kd of one (cout block, cin block) pair and a spatial partition: wave = (row half of the tile, kh, position half)
// keeps three dU_p accumulators over its K-steps of every tile; the tile is double-buffered in LDS (one barrier per
// tile: waves that finish their MFMAs store the next tile while the others still compute).  Partial slabs
// [2P][pair][kd][kh][p][32][32] are summed in a fixed order by wgrad_wino_reduce_kernel, which also applies G^T: deterministic, fp32 throughout (measured against torch autograd in
// tests/test_gpu_parity.py, same 1e-3 bound as the direct form; desc.algo = SNVC_ALGO_DIRECT keeps the direct form).
struct WinoWgradCfg {
    static constexpr int TH = 4, TW = 32, THREADS = 768;
    static constexpr int IN_H = TH + 2, LPAD = 4;
    static constexpr int IN_WV = 40;                       // image columns ow0-4 .. ow0+35
    // Channel strides are 4 * (odd) floats: rows stay 16-byte aligned (whole pieces are stored with one ds_write_b128, no
    // bank conflicts -- the odd strides of the direct kernel force four 4-way conflicting ds_write_b32 per piece) and a
    // lane's quad is ONE ds_read_b128 whose 16-lane groups cover all 64 banks (4 * ch * odd mod 64 is a permutation).
    static constexpr int XVV = IN_H * IN_WV, XSV = XVV + 4;      // 244 = 4 * 61
    static constexpr int GV = TH * TW, GS = GV + 4;              // 132 = 4 * 33
    static_assert((XSV / 4) % 2 == 1 && (GS / 4) % 2 == 1 && XSV % 4 == 0 && GS % 4 == 0, "channel stride = 4 * odd");
    static constexpr int RQ = IN_WV / 4;
    static constexpr int XITEMS = 32 * IN_H * RQ, XNIT = (XITEMS + THREADS - 1) / THREADS;      // 16-byte pieces
    static constexpr int GITEMS = 32 * TH * 8, GNIT = (GITEMS + THREADS - 1) / THREADS;
    static constexpr int BUF_FLOATS = 32 * XSV + 32 * GS;  // 47 KB
    static constexpr int LDS_FLOATS = 2 * BUF_FLOATS;      // double-buffered
    static constexpr int SLABS = 18;                       // (kh, p) per kernel depth slice
};

__global__ void __launch_bounds__(768, 1)
conv3d_wgrad_wino_kernel(const WgradArgs a) {
    using Cfg = WinoWgradCfg;
    constexpr int IN_H = Cfg::IN_H, IN_WV = Cfg::IN_WV, XSV = Cfg::XSV, GS = Cfg::GS, RQ = Cfg::RQ, T = Cfg::THREADS;
    constexpr int XNIT = Cfg::XNIT, GNIT = Cfg::GNIT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // buffer b: X [32][XSV] at lds + b * BUF_FLOATS, G [32][GS] behind it
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = wave / 6, role = wave - 6 * kq;      // this wave: tile rows 2kq, 2kq+1,
    const int kh = role >> 1, ph = role & 1;            // kernel row kh, Winograd positions 3*ph .. 3*ph+2
    // 1-D grid, id -> (XCD id & 7, slot id >> 3), at most one workgroup per CU of each XCD: the three kd workgroups of a
    // (pair, partition) unit sit on ONE XCD, units that are neighbours in depth (tiles run depth-fastest, partition = tile
    // mod P) beside them: the dy tile is fetched into that L2 once for the three kd, and input slice od+1 read by kd = 2
    // of one partition is read at the same time by kd = 1 / kd = 0 of the next two.  (With kd on blockIdx.z the loads
    // alone took 1.48 ms of the layer's 2.45 ms: 6.3 GB over the fabric.)
    const int slot = blockIdx.x >> 3;
    const int kd = slot % 3, unit = (blockIdx.x & 7) * a.upx + slot / 3;
    if (unit >= a.pairs * a.P) return;
    const int part = unit / a.pairs, pair = unit - part * a.pairs;    // pairs of one partition share tiles: same XCD
    const int cgb = pair / a.cx_blocks, cxb = pair - cgb * a.cx_blocks;
    const int cg0 = cgb * 32, cx0 = cxb * 32;

    f32x16 acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

    const int64_t in_hw = (int64_t)a.Hi * a.Wi, in_dhw = in_hw * a.Di;
    const int64_t out_hw = (int64_t)a.Ho * a.Wo, out_dhw = out_hw * a.Do;
    const int ch = lane & 31, half = lane >> 5;
    const int xrow = ch * XSV + (kh + 2 * kq) * IN_WV + 4 * half, grow = 32 * XSV + ch * GS + 2 * kq * 32 + 4 * half;

    // staging: per item, tile-invariant: the 32-bit element offset from the tile's origin, the LDS offset, and the packed
    // code hh | q << 4 for the border tests; bit `it` of xstat / gstat = the item exists (inside the tile's item count and
    // the layer's channels).  Interior tiles (72% of cfg2's conv1) load with no per-item test or address arithmetic: a
    // wave-uniform base plus the item's offset.
    unsigned xoff[XNIT], goff[GNIT];
    int xdst[XNIT], gdst[GNIT], xcode[XNIT], gcode[GNIT];
    unsigned xstat = 0, gstat = 0;
#pragma unroll
    for (int it = 0; it < XNIT; ++it) {
        const int i = it * T + tid;
        const int c = i / (IN_H * RQ), r = i - c * (IN_H * RQ);
        const int hh = r / RQ, q = r - hh * RQ;
        const bool ex = i < Cfg::XITEMS && cx0 + c < a.Cx;
        xstat |= (ex ? 1u : 0u) << it;
        xcode[it] = hh | (q << 4);
        xoff[it] = ex ? (unsigned)(c * in_dhw + (int64_t)hh * a.Wi + 4 * q) : 0u;
        xdst[it] = i < Cfg::XITEMS ? c * XSV + hh * IN_WV + 4 * q : -1;
    }
#pragma unroll
    for (int it = 0; it < GNIT; ++it) {
        const int i = it * T + tid;
        const int c = i / (Cfg::TH * 8), r = i - c * (Cfg::TH * 8);
        const int hh = r / 8, q = r - hh * 8;
        const bool ex = i < Cfg::GITEMS && cg0 + c < a.Cg;
        gstat |= (ex ? 1u : 0u) << it;
        gcode[it] = hh | (q << 4);
        goff[it] = ex ? (unsigned)(c * out_dhw + (int64_t)hh * a.Wo + 4 * q) : 0u;
        gdst[it] = i < Cfg::GITEMS ? 32 * XSV + c * GS + hh * 32 + 4 * q : -1;
    }
    f32x4 xv[XNIT], gv[GNIT];
    unsigned xok = 0, gok = 0;
    const unsigned tiles_dw = (unsigned)a.Do * a.tiles_w, tiles_dwh = tiles_dw * a.tiles_h;
    auto load_tile = [&](unsigned tile) {
        const unsigned n = tile / tiles_dwh, r0 = tile - n * tiles_dwh;           // depth fastest
        const unsigned th = r0 / tiles_dw, r1 = r0 - th * tiles_dw;
        const int tw = (int)(r1 / (unsigned)a.Do), od = (int)(r1 - tw * (unsigned)a.Do);
        const int oh0 = th * Cfg::TH, ow0 = tw * 32;
        const int id = od - 1 + kd, ih0 = oh0 - 1, ix0 = ow0 - Cfg::LPAD;
        const float *gb = a.g + n * a.g_bs + (int64_t)cg0 * out_dhw + (int64_t)od * out_hw + (int64_t)oh0 * a.Wo + ow0;
        const bool d_ok = (unsigned)id < (unsigned)a.Di;
        // the X origin may lie before the tensor on border tiles: it is only dereferenced through items that passed their tests
        const float *xb = a.x + n * a.x_bs + (int64_t)cx0 * in_dhw + (int64_t)id * in_hw + (int64_t)ih0 * a.Wi + ix0;
        const bool interior = d_ok && ih0 >= 0 && ih0 + IN_H <= a.Hi && ix0 >= 0 && ix0 + IN_WV <= a.Wi &&
                              oh0 + Cfg::TH <= a.Ho && ow0 + 32 <= a.Wo;
        if (interior) {
            xok = xstat; gok = gstat;
#pragma unroll
            for (int it = 0; it < XNIT; ++it) xv[it] = *reinterpret_cast<const f32x4 *>(xb + xoff[it]);    // offset 0 (a valid
#pragma unroll                                                                                             // element) where
            for (int it = 0; it < GNIT; ++it) gv[it] = *reinterpret_cast<const f32x4 *>(gb + goff[it]);    // the item is absent
        } else {
            xok = 0; gok = 0;
#pragma unroll
            for (int it = 0; it < XNIT; ++it) {
                const int hh = xcode[it] & 15, q = xcode[it] >> 4;
                const bool ok = ((xstat >> it) & 1u) && d_ok && (unsigned)(ih0 + hh) < (unsigned)a.Hi &&
                                (unsigned)(ix0 + 4 * q) < (unsigned)a.Wi;
                xv[it] = *reinterpret_cast<const f32x4 *>(ok ? xb + xoff[it] : a.x);
                xok |= (ok ? 1u : 0u) << it;
            }
#pragma unroll
            for (int it = 0; it < GNIT; ++it) {
                const int hh = gcode[it] & 15, q = gcode[it] >> 4;
                const bool ok = ((gstat >> it) & 1u) && oh0 + hh < a.Ho && ow0 + 4 * q < a.Wo;
                gv[it] = *reinterpret_cast<const f32x4 *>(ok ? gb + goff[it] : a.g);
                gok |= (ok ? 1u : 0u) << it;
            }
        }
    };
    auto store_tile = [&](float *base) {
#pragma unroll
        for (int it = 0; it < XNIT; ++it) {
            if (Cfg::XITEMS % T != 0 && xdst[it] < 0) continue;
            *reinterpret_cast<f32x4 *>(base + xdst[it]) = ((xok >> it) & 1u) ? xv[it] : f32x4(0.0f);
        }
#pragma unroll
        for (int it = 0; it < GNIT; ++it) {
            if (Cfg::GITEMS % T != 0 && gdst[it] < 0) continue;
            *reinterpret_cast<f32x4 *>(base + gdst[it]) = ((gok >> it) & 1u) ? gv[it] : f32x4(0.0f);
        }
    };
    unsigned tile = part;              // ntiles < 2^31 (host-checked)
    int buf = 0;
    if (tile < a.ntiles) {
        load_tile(tile);
        store_tile(lds);
        if (tile + a.P < a.ntiles) load_tile(tile + a.P);
    }
    __syncthreads();
    for (; tile < a.ntiles; tile += a.P, buf ^= 1) {
        // 8 K-steps of two quads (lanes 0-31: quad 2kk, lanes 32-63: quad 2kk+1) over this wave's two rows; quad q = row q / 8,
        // quad t = q % 8 (q = 2kk + half: row kk >> 2 and quad 2 (kk & 3) + half, so everything but 4 * half is an immediate)
        const float *gq = lds + buf * Cfg::BUF_FLOATS + grow, *xq = lds + buf * Cfg::BUF_FLOATS + xrow;
        // The three LDS reads of K-step kk + 1 are issued between the transforms and the MFMAs of step kk and pinned there (r4:
        // the compiler had sunk every ds_read_b128 to just in front of its consumers, s_waitcnt lgkmcnt(0) per step -- one exposed
        // LDS round trip per 3 MFMAs, hidden only by the SIMD's other two waves: pipe 0.50; the kernel is NOT power-limited, its
        // time is the same on all-zero operands).  Behind the transforms, because the compiler waits with lgkmcnt(0) before a
        // step's first VALU use: there nothing younger is in flight, the reads issued next have the step's 192 MFMA cycles.
        // Both position halves read piece t+1 of the input row; ph == 0 adds the float before it, ph == 1 the float behind it.
        // (the one float a step needs of its second input piece is read as 4 bytes: a ds_read_b128 whose other three lanes
        // are dead gets a destination that overlaps the next read's, and a wait between the two)
        f32x4 dyb[2], xmb[2];
        float xsb[2];
        auto fetch = [&](int kk, int b) {
            dyb[b] = *reinterpret_cast<const f32x4 *>(gq + 8 * kk);                                           // r * 32 + 4 * t == 4 * q
            const float *xp = xq + (kk >> 2) * IN_WV + 8 * (kk & 3);
            xmb[b] = *reinterpret_cast<const f32x4 *>(xp + 4);          // piece t + 1: d1 .. d4
            xsb[b] = xp[ph == 0 ? 3 : 8];                               // d0 (last float of piece t) or d5 (first of piece t + 2)
        };
        fetch(0, 0);
        if (ph == 0) {
#pragma unroll
            for (int kk = 0; kk < Cfg::TH * 2; ++kk) {
                const int b = kk & 1;
                // the quad's inputs x[4t-1 .. 4t+4] are image columns 4t+3 .. 4t+8: last float of piece t, piece t+1, first of t+2
                const f32x4 dy = dyb[b], x1 = xmb[b];
                const float dy0 = dy[0], dy1 = dy[1], dy2 = dy[2], dy3 = dy[3];
                const float d0 = xsb[b], d1 = x1[0], d2 = x1[1], d3 = x1[2], d4 = x1[3];
                const float s02 = dy0 + dy2, s13 = dy1 + dy3;
                const float e = __builtin_fmaf(-4.0f, d2, d4), f = __builtin_fmaf(-4.0f, d1, d3);
                const float v0 = __builtin_fmaf(4.0f, d0, __builtin_fmaf(-5.0f, d2, d4));
                const float a1 = s02 + s13, b1 = e + f, a2 = s02 - s13, b2 = e - f;
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < Cfg::TH * 2) fetch(kk + 1, b ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dy0, v0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc[2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < Cfg::TH * 2; ++kk) {
                const int b = kk & 1;
                const f32x4 dy = dyb[b], x1 = xmb[b];
                const float dy0 = dy[0], dy1 = dy[1], dy2 = dy[2], dy3 = dy[3];
                const float d1 = x1[0], d2 = x1[1], d3 = x1[2], d4 = x1[3], d5 = xsb[b];
                const float t02 = __builtin_fmaf(4.0f, dy2, dy0), t13 = __builtin_fmaf(8.0f, dy3, 2.0f * dy1);
                const float c2 = d4 - d2, e2 = d3 - d1;
                const float v5 = __builtin_fmaf(4.0f, d1, __builtin_fmaf(-5.0f, d3, d5));
                const float a0 = t02 + t13, b0 = __builtin_fmaf(2.0f, e2, c2), a1 = t02 - t13, b1 = __builtin_fmaf(-2.0f, e2, c2);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < Cfg::TH * 2) fetch(kk + 1, b ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(dy3, v5, acc[2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the other buffer was last read before the previous barrier: refill it, then start the loads of the tile after
        if (tile + a.P < a.ntiles) {
            store_tile(lds + (buf ^ 1) * Cfg::BUF_FLOATS);
            if (tile + 2 * a.P < a.ntiles) load_tile(tile + 2 * a.P);
        }
        __syncthreads();
    }
    // ---- partial slab [partition * 2 + row half][pair][kd][kh][pos][cg 32][cx 32]
    float *pp = a.partial + ((((int64_t)part * 2 + kq) * a.pairs + pair) * 3 + kd) * (int64_t)Cfg::SLABS * 1024 +
                (int64_t)(kh * 6 + ph * 3) * 1024;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;            // cg
            pp[(p * 32 + row) * 32 + ch] = acc[p][r];
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3x3 / stride-2 / pad-1 layers (and, with roles swapped, of the transposed layers) in the
// 12-wave form of conv3d_wgrad_wino_kernel: a workgroup owns one kernel depth slice kd of one (cout block, cin block)
// pair and a spatial partition, tiles of 2 output rows x 32 output columns double-buffered in LDS (input rows
// 2*oh0-1 .. +4, columns 2*ow0-4 .. +63 as 17 aligned 16-byte pieces; channel strides 4 * odd floats).
// wave = (kh, quarter of the tile's 16 output quads); a lane reads ITS channel's dy quad (one b128: 4 voxels) and the 12
// input floats under it (three b128: columns 8t .. 8t+11 of row 2r+kh, of which 8t+3 .. 8t+11 are used) and issues the
// 12 MFMAs (4 voxels x kw = 0,1,2) with no arithmetic in between: the direct tap-split kernel this replaces ran at
// 26 % of the matrix pipe on these layers (one output row per tile, scalar LDS reads, 6 % of the cfg4 step each).
// Partial slabs [4 P][pair][27 taps][32][32] in the layout of the direct kernel: summed by wgrad_reduce_kernel in a
// fixed order (deterministic).  GP = floats per staged dy piece (4, or 2 for rows that are only 8-byte aligned: W = 78).
struct S2WgradCfg {
    static constexpr int TH = 2, THREADS = 768;
    static constexpr int IN_H = 2 * TH + 1, IN_WV = 68, RQ = IN_WV / 4;       // 5 rows of 17 pieces
    static constexpr int XSV = IN_H * IN_WV;                                  // 340 = 4 * 85
    static constexpr int GS = TH * 32 + 4;                                    // 68 = 4 * 17
    static_assert((XSV / 4) % 2 == 1 && (GS / 4) % 2 == 1 && XSV % 4 == 0, "channel stride = 4 * odd");
    static constexpr int XITEMS = 32 * IN_H * RQ, XNIT = (XITEMS + THREADS - 1) / THREADS;
    static constexpr int BUF_FLOATS = 32 * XSV + 32 * GS, LDS_FLOATS = 2 * BUF_FLOATS;     // 2 x 51 KB
};

template <int GP>
__global__ void __launch_bounds__(768, 1)
conv3d_wgrad_s2_kernel(const WgradArgs a) {
    using Cfg = S2WgradCfg;
    constexpr int IN_H = Cfg::IN_H, IN_WV = Cfg::IN_WV, XSV = Cfg::XSV, GS = Cfg::GS, RQ = Cfg::RQ, T = Cfg::THREADS;
    constexpr int XNIT = Cfg::XNIT, GQ = 32 / GP, GITEMS = 32 * Cfg::TH * GQ, GNIT = (GITEMS + T - 1) / T;
    typedef float GVec __attribute__((ext_vector_type(GP)));
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = wave / 3, kh = wave - 3 * kq;        // this wave: kernel row kh, quads 4 kq .. 4 kq + 3 of the tile
    const int slot = blockIdx.x >> 3;                   // XCD-grouped units as in conv3d_wgrad_wino_kernel
    const int kd = slot % 3, unit = (blockIdx.x & 7) * a.upx + slot / 3;
    if (unit >= a.pairs * a.P) return;
    const int part = unit / a.pairs, pair = unit - part * a.pairs;    // pairs of one partition share tiles: same XCD
    const int cgb = pair / a.cx_blocks, cxb = pair - cgb * a.cx_blocks;
    const int cg0 = cgb * 32, cx0 = cxb * 32;

    f32x16 acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

    const int64_t in_hw = (int64_t)a.Hi * a.Wi, in_dhw = in_hw * a.Di;
    const int64_t out_hw = (int64_t)a.Ho * a.Wo, out_dhw = out_hw * a.Do;
    const int ch = lane & 31, half = lane >> 5;
    // group j (0, 1) of this wave: output row r = kq >> 1, quad t = 4 (kq & 1) + 2 j + half
    const int r = kq >> 1, t0 = 4 * (kq & 1) + half;
    const int xrow = ch * XSV + (2 * r + kh) * IN_WV + 8 * t0, grow = 32 * XSV + ch * GS + r * 32 + 4 * t0;

    unsigned xoff[XNIT], goff[GNIT];
    int xdst[XNIT], gdst[GNIT], xcode[XNIT], gcode[GNIT];
    unsigned xstat = 0, gstat = 0;
#pragma unroll
    for (int it = 0; it < XNIT; ++it) {
        const int i = it * T + tid;
        const int c = i / (IN_H * RQ), rr = i - c * (IN_H * RQ);
        const int hh = rr / RQ, q = rr - hh * RQ;
        const bool ex = i < Cfg::XITEMS && cx0 + c < a.Cx;
        xstat |= (ex ? 1u : 0u) << it;
        xcode[it] = hh | (q << 4);
        xoff[it] = ex ? (unsigned)(c * in_dhw + (int64_t)hh * a.Wi + 4 * q) : 0u;
        xdst[it] = i < Cfg::XITEMS ? c * XSV + hh * IN_WV + 4 * q : -1;
    }
#pragma unroll
    for (int it = 0; it < GNIT; ++it) {
        const int i = it * T + tid;
        const int c = i / (Cfg::TH * GQ), rr = i - c * (Cfg::TH * GQ);
        const int hh = rr / GQ, q = rr - hh * GQ;
        const bool ex = i < GITEMS && cg0 + c < a.Cg;
        gstat |= (ex ? 1u : 0u) << it;
        gcode[it] = hh | (q << 4);
        goff[it] = ex ? (unsigned)(c * out_dhw + (int64_t)hh * a.Wo + GP * q) : 0u;
        gdst[it] = i < GITEMS ? 32 * XSV + c * GS + hh * 32 + GP * q : -1;
    }
    f32x4 xv[XNIT];
    GVec gv[GNIT];
    unsigned xok = 0, gok = 0;
    const unsigned tiles_dw = (unsigned)a.Do * a.tiles_w, tiles_dwh = tiles_dw * a.tiles_h;
    auto load_tile = [&](unsigned tile) {
        const unsigned n = tile / tiles_dwh, r0 = tile - n * tiles_dwh;           // depth fastest
        const unsigned th = r0 / tiles_dw, r1 = r0 - th * tiles_dw;
        const int tw = (int)(r1 / (unsigned)a.Do), od = (int)(r1 - tw * (unsigned)a.Do);
        const int oh0 = th * Cfg::TH, ow0 = tw * 32;
        const int id = 2 * od - 1 + kd, ih0 = 2 * oh0 - 1, ix0 = 2 * ow0 - 4;
        const float *gb = a.g + n * a.g_bs + (int64_t)cg0 * out_dhw + (int64_t)od * out_hw + (int64_t)oh0 * a.Wo + ow0;
        const bool d_ok = (unsigned)id < (unsigned)a.Di;
        // the X origin may lie before the tensor on border tiles: it is only dereferenced through items that passed their tests
        const float *xb = a.x + n * a.x_bs + (int64_t)cx0 * in_dhw + (int64_t)id * in_hw + (int64_t)ih0 * a.Wi + ix0;
        const bool interior = d_ok && ih0 >= 0 && ih0 + IN_H <= a.Hi && ix0 >= 0 && ix0 + IN_WV <= a.Wi &&
                              oh0 + Cfg::TH <= a.Ho && ow0 + 32 <= a.Wo;
        if (interior) {
            xok = xstat; gok = gstat;
#pragma unroll
            for (int it = 0; it < XNIT; ++it) xv[it] = *reinterpret_cast<const f32x4 *>(xb + xoff[it]);
#pragma unroll
            for (int it = 0; it < GNIT; ++it) gv[it] = *reinterpret_cast<const GVec *>(gb + goff[it]);
        } else {
            xok = 0; gok = 0;
#pragma unroll
            for (int it = 0; it < XNIT; ++it) {
                const int hh = xcode[it] & 15, q = xcode[it] >> 4;
                const bool ok = ((xstat >> it) & 1u) && d_ok && (unsigned)(ih0 + hh) < (unsigned)a.Hi &&
                                (unsigned)(ix0 + 4 * q) < (unsigned)a.Wi;
                xv[it] = *reinterpret_cast<const f32x4 *>(ok ? xb + xoff[it] : a.x);
                xok |= (ok ? 1u : 0u) << it;
            }
#pragma unroll
            for (int it = 0; it < GNIT; ++it) {
                const int hh = gcode[it] & 15, q = gcode[it] >> 4;
                const bool ok = ((gstat >> it) & 1u) && oh0 + hh < a.Ho && ow0 + GP * q < a.Wo;
                gv[it] = *reinterpret_cast<const GVec *>(ok ? gb + goff[it] : a.g);
                gok |= (ok ? 1u : 0u) << it;
            }
        }
    };
    auto store_tile = [&](float *base) {
#pragma unroll
        for (int it = 0; it < XNIT; ++it) {
            if (Cfg::XITEMS % T != 0 && xdst[it] < 0) continue;
            *reinterpret_cast<f32x4 *>(base + xdst[it]) = ((xok >> it) & 1u) ? xv[it] : f32x4(0.0f);
        }
#pragma unroll
        for (int it = 0; it < GNIT; ++it) {
            if (GITEMS % T != 0 && gdst[it] < 0) continue;
            *reinterpret_cast<GVec *>(base + gdst[it]) = ((gok >> it) & 1u) ? gv[it] : GVec(0.0f);
        }
    };
    auto compute = [&](int buf) {
        const float *gq = lds + buf * Cfg::BUF_FLOATS + grow, *xq = lds + buf * Cfg::BUF_FLOATS + xrow;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 dy = *reinterpret_cast<const f32x4 *>(gq + 8 * j);        // quads t0 + 2 j: 4 (t0 + 2 j) floats into the row
            const float *xp = xq + 16 * j;                                        // 8 columns per quad
            const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xp), x1 = *reinterpret_cast<const f32x4 *>(xp + 4),
                        x2 = *reinterpret_cast<const f32x4 *>(xp + 8);
            // output ow = 4t + i reads input columns 8t + 2i + kw + 3 of the staged row (ix0 = 2 ow0 - 4, pad 1)
            const float f[12] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(dy[i], f[2 * i + kw + 3], acc[kw], 0, 0, 0);
        }
    };
    unsigned tile = part;              // ntiles < 2^30 (host-checked)
    int buf = 0;
    if (tile < a.ntiles) {
        load_tile(tile);
        store_tile(lds);
        if (tile + a.P < a.ntiles) load_tile(tile + a.P);
    }
    __syncthreads();
    for (; tile < a.ntiles; tile += a.P, buf ^= 1) {
        compute(buf);
        // the other buffer was last read before the previous barrier: refill it, then start the loads of the tile after
        if (tile + a.P < a.ntiles) {
            store_tile(lds + (buf ^ 1) * Cfg::BUF_FLOATS);
            if (tile + 2 * a.P < a.ntiles) load_tile(tile + 2 * a.P);
        }
        __syncthreads();
    }
    // ---- partial slab [partition * 4 + kq][pair][tap (kd, kh, kw)][cg 32][cx 32]
    float *pp = a.partial + ((((int64_t)part * 4 + kq) * a.pairs + pair) * 27 + (kd * 3 + kh) * 3) * 1024;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * half;          // cg
            pp[(kw * 32 + row) * 32 + ch] = acc[kw][rr];
        }
}

// dw[cg][cx][kd][kh][kw] = sum_p G[p][kw] * (sum over partitions of the dU_p slabs); one thread per (pair, kd, kh, cg, cx)
__global__ void wgrad_wino_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dw, int Cg, int Cx, int cx_blocks,
                                         int pairs, int P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_pair = 9 * 1024;                          // (kd, kh) x 32 x 32
    if (i >= (int64_t)pairs * per_pair) return;
    const int cxl = (int)(i & 31), cgl = (int)((i >> 5) & 31);
    const int kdh = (int)((i >> 10) % 9);
    const int pair = (int)(i / per_pair);
    const int cg = (pair / cx_blocks) * 32 + cgl, cx = (pair % cx_blocks) * 32 + cxl;
    if (cg >= Cg || cx >= Cx) return;
    const int64_t pstride = (int64_t)pairs * 54 * 1024;
    const float *src = partial + ((int64_t)pair * 54 + kdh * 6) * 1024 + cgl * 32 + cxl;
    float u[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        const float *sp = src + p * 1024;
        int k = 0;
        for (; k + 4 <= P; k += 4) {
            s0 += sp[(int64_t)k * pstride];
            s1 += sp[(int64_t)(k + 1) * pstride];
            s2 += sp[(int64_t)(k + 2) * pstride];
            s3 += sp[(int64_t)(k + 3) * pstride];
        }
        for (; k < P; ++k) s0 += sp[(int64_t)k * pstride];
        u[p] = (s0 + s1) + (s2 + s3);
    }
    // G^T (wino tables of F(4,3): U0 = g0/4, U1 = -(g0+g1+g2)/6, U2 = -(g0-g1+g2)/6, U3 = g0/24+g1/12+g2/6,
    // U4 = g0/24-g1/12+g2/6, U5 = g2)
    const float g0 = u[0] * 0.25f - (u[1] + u[2]) * (1.0f / 6.0f) + (u[3] + u[4]) * (1.0f / 24.0f);
    const float g1 = (u[2] - u[1]) * (1.0f / 6.0f) + (u[3] - u[4]) * (1.0f / 12.0f);
    const float g2 = -(u[1] + u[2]) * (1.0f / 6.0f) + (u[3] + u[4]) * (1.0f / 6.0f) + u[5];
    float *o = dw + (((int64_t)cg * Cx + cx) * 9 + kdh) * 3;
    o[0] = g0; o[1] = g1; o[2] = g2;
}

template <class Cfg, int VEC>
void launch_wgrad_variant(const WgradArgs &a, dim3 grid, hipStream_t st) {
    constexpr int bytes = Cfg::LDS_FLOATS * 4;
    static std::atomic<unsigned> attr_done{0};   // one bit per device: the attribute is per device
    if (!allow_large_lds(reinterpret_cast<const void *>(&conv3d_wgrad_kernel<Cfg, VEC>), bytes, attr_done)) return;
    conv3d_wgrad_kernel<Cfg, VEC><<<grid, 256, bytes, st>>>(a);
}

template <class Cfg>
void launch_wgrad(const WgradArgs &a, dim3 grid, hipStream_t st) {
    // the prefetching float4 path is built for the K-split configurations only (the tap-split ones have no
    // registers left for the prefetch: 112 accumulator registers + 60 of staging spill)
    if constexpr (Cfg::KSPLIT) {
        if (a.vec == 2) launch_wgrad_variant<Cfg, 2>(a, grid, st);
        else launch_wgrad_variant<Cfg, 4>(a, grid, st);
    } else {
        launch_wgrad_variant<Cfg, 0>(a, grid, st);
    }
}

// ------------------------------------------------------------------ split-operand (f16x3) weight gradient, 3x3x3 / stride 1 (r6)
// The fp32 forms above run at half of the fp32 matrix pipe (conv2's 32 -> 32 layer on 192 x 96 x 312: 2.07 ms, the largest launch of
// the cfg4 step).  This form puts the same contraction on v_mfma_f32_16x16x32_f16: both operands are split on the
// way into LDS into (hi, lo) pairs of halves -- x * sx = hi + lo with a per-tensor power-of-two scale taken from the tensor's own
// maximum (wgrad_amax2_kernel), 22 significant bits -- and every fp32 product becomes three MFMAs (hi*hi + hi*lo + lo*hi) with
// fp32 accumulation, as in the forward split mode (conv3d_f16.hip).  K of the GEMM is the voxel index, so the natural operand
// layout is the tensors' own NCDHW: a lane's 8 consecutive k are 8 consecutive w of one channel (no layout pass; 16-byte LDS reads
// from channel-major rows).
//   A (16 x 32) = x[ci][w0 + 8 kg + j + kw - 1]   B (32 x 16) = g[co][w0 + 8 kg + j]   D[ci][co] += A B      per tap and 32 voxels
// At f16 rates the layer has 650 flop per algorithmic byte, twice the ridge: every staged byte has to serve ALL 27 taps.  A
// workgroup therefore owns a COLUMN of the grid -- 4 output rows x 32 output columns -- and walks it along d with a ring of
// three input planes in LDS (6 rows x 32 channels x 40 columns each, as halves): per step one new input plane (26 KB) and one
// gradient plane (16 KB) arrive for 4 x 32 voxels x 27 taps x 32 x 32 channels.  The next step's planes travel global -> registers
// under this step's MFMAs and are split into LDS between two barriers.
// Four waves, one per SIMD: wave = (16-channel half of x) x (row pair); it keeps ALL 27 taps x both 16-channel halves of g:
// 216 accumulator registers.  The kw = 0 / 2 operands are the kw = 1 piece shifted by one half: v_alignbit with the neighbouring
// dwords (an unaligned ds_read_b128 is replayed at 64 cycles), computed once and used for both halves of g: 16 VALU per 54 MFMAs.
// Partial slabs [column * 2 + row pair][pair][27][cg 32][cx 32] (the direct kernels' layout) are summed in a fixed order by
// wgrad_reduce_kernel, which also removes the two scales: deterministic.  desc.algo |= SNVC_ALGO_WGRAD_FP32 keeps the fp32 forms.
typedef _Float16 h8w __attribute__((ext_vector_type(8)));
typedef _Float16 h4w __attribute__((ext_vector_type(4)));
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));

struct X3WgCfg {
    static constexpr int TH = 4, XR = TH + 2, THREADS = 512;      // 4 x 32 gradient voxels per plane step
    static constexpr int XCOLS = 32;                         // halves per staged x row: image columns w0 .. w0+31 (64 bytes, chunks swizzled)
    static constexpr int XPLANE = XR * 32 * XCOLS;           // halves of the hi (or lo) half of a slot
    static constexpr int SLOT_BYTES = 2 * XPLANE * 2;        // hi | lo = 24576
    static constexpr int NSLOT = 4;
    static constexpr int GCOLS = 48;                         // halves per staged dy row: columns w0-8 .. w0+39 (96-byte stride: conflict-free b128)
    static constexpr int GPLANE = TH * 32 * GCOLS;
    static constexpr int G_BYTES = 2 * GPLANE * 2;           // 24576
    static constexpr int LDS_BYTES = NSLOT * SLOT_BYTES + 2 * G_BYTES;      // 147456
    static constexpr int LDS_ALLOC = LDS_BYTES + 16384;                      // + a 16 KB sink for the last step's (dead) stores
    static constexpr int XPIECES = XR * 32 * 8, XNIT = XPIECES / THREADS;                        // float4 pieces of an input plane: 1536 = 3 rounds
    static constexpr int GPIECES = TH * 32 * 10, GNIT = (GPIECES + THREADS - 1) / THREADS;       // 1280 -> 3 rounds (the last one half full)
    static_assert(XPIECES % THREADS == 0, "whole rounds");
};

// a float4 piece of a row: one 16-byte load, or -- rows that are only 8-byte aligned (W % 4 == 2: the W = 78 level of the cfg2
// hourglass) -- two 8-byte loads whose second half may lie beyond the row (then it reads element 0 and is zeroed by its own flag)
typedef float f32x2w __attribute__((ext_vector_type(2)));
template <bool A16>
__device__ __forceinline__ f32x4 x3wg_ld4(const float *__restrict__ b, unsigned off, unsigned off_hi) {
    if constexpr (A16) {
        return *reinterpret_cast<const f32x4 *>(b + off);
    } else {
        const f32x2w lo = *reinterpret_cast<const f32x2w *>(b + off), hi = *reinterpret_cast<const f32x2w *>(b + off_hi);
        return f32x4{lo[0], lo[1], hi[0], hi[1]};
    }
}
__device__ __forceinline__ f32x4 x3wg_mask(f32x4 v, bool lo_ok, bool hi_ok) {
    return f32x4{lo_ok ? v[0] : 0.0f, lo_ok ? v[1] : 0.0f, hi_ok ? v[2] : 0.0f, hi_ok ? v[3] : 0.0f};
}

struct X3WgArgs {
    const float *x, *g;
    float *partial;
    const unsigned *amax_x, *amax_g;       // bits of max|x|, max|g| (or of upper bounds)
    int N, Cx, Cg, D, H, W;
    int cx_blocks, pairs, hblocks, wsegs, dparts, dchunk, njobs;
    int pairs32;                // stride-2 form: 32 x 32 channel pairs of the slab layout (its jobs take 64 g channels)
    int64_t x_bs, g_bs;
};

// the maximum of the SNVC_AMAX_SLOTS words a producer pass left (snvc_affine_act_amax ...): every lane reads one, the wave reduces
__device__ __forceinline__ unsigned x3wg_amax(const unsigned *__restrict__ p) {
    unsigned m = p[threadIdx.x & (SNVC_AMAX_SLOTS - 1)];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, off);
        m = m > o ? m : o;
    }
    return m;
}

// power of two s with max * s in [2^13, 2^14) (half: 65504); 1 for an all-zero or non-finite tensor
__device__ __forceinline__ float x3wg_scale(unsigned bits) {
    const float a = __uint_as_float(bits);
    if (!(a > 0.0f) || !(a < __builtin_inff())) return 1.0f;
    int e;
    (void)frexpf(a, &e);
    int k = 14 - e;
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return ldexpf(1.0f, k);
}

__global__ void __launch_bounds__(256)
wgrad_amax2_kernel(const float *__restrict__ x, const float *__restrict__ g, unsigned *__restrict__ out, int64_t n4x, int64_t n4g,
                   int64_t x_bs, int64_t g_bs) {
    const int which = blockIdx.y;
    const f32x4 *p = reinterpret_cast<const f32x4 *>((which ? g : x) + (int64_t)blockIdx.z * (which ? g_bs : x_bs));
    const int64_t n4 = which ? n4g : n4x;
    // |v| as bits: for non-negative floats the unsigned order is the float order; a NaN sorts above infinity
    auto amax4 = [](const f32x4 v) {
        const unsigned b0 = __float_as_uint(v[0]) & 0x7fffffffu, b1 = __float_as_uint(v[1]) & 0x7fffffffu;
        const unsigned b2 = __float_as_uint(v[2]) & 0x7fffffffu, b3 = __float_as_uint(v[3]) & 0x7fffffffu;
        const unsigned m01 = b0 > b1 ? b0 : b1, m23 = b2 > b3 ? b2 : b3;
        return m01 > m23 ? m01 : m23;
    };
    unsigned m = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {           // four 16-byte loads in flight per lane
        const f32x4 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
        const unsigned a0 = amax4(v0), a1 = amax4(v1), a2 = amax4(v2), a3 = amax4(v3);
        const unsigned a01 = a0 > a1 ? a0 : a1, a23 = a2 > a3 ? a2 : a3;
        const unsigned aa = a01 > a23 ? a01 : a23;
        m = m > aa ? m : aa;
    }
    for (; i < n4; i += stride) {
        const unsigned aa = amax4(p[i]);
        m = m > aa ? m : aa;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, off);
        m = m > o ? m : o;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out + which * SNVC_AMAX_SLOTS + (blockIdx.x & (SNVC_AMAX_SLOTS - 1)), m);
}

template <bool A16>      // rows 16-byte aligned (else 8-byte: W % 4 == 2)
__global__ void __launch_bounds__(512, 1)
conv3d_wgrad_x3_kernel(const X3WgArgs a) {
    using Cfg = X3WgCfg;
    constexpr int XNIT = Cfg::XNIT, GNIT = Cfg::GNIT, XCOLS = Cfg::XCOLS, GCOLS = Cfg::GCOLS;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char *const lds = reinterpret_cast<char *>(lds_f);
    char *const gbase = lds + Cfg::NSLOT * Cfg::SLOT_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this wave: x channels 16 cih .. +15, g channels 16 coh .. +15, gradient rows 2 kp, 2 kp + 1.  Waves w and w + 4 share a SIMD
    // and differ in kp: one of them computes while the other one stages (see the loop)
    const int cih = wave & 1, coh = (wave >> 1) & 1, kp = wave >> 2;
    const int i16 = lane & 15, kg = lane >> 4;
    // job -> (column, pair, depth part); the pairs of a column are neighbours on one XCD (they share its planes through that L2)
    const int q8 = a.njobs >> 3, r8 = a.njobs & 7, xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    const int job = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kx;
    if (kx >= q8 + (xcd < r8 ? 1 : 0)) return;
    const int pair = job % a.pairs, col = job / a.pairs;          // col = ((n * dparts + dp) * hblocks + hb) * wsegs + ws
    const int ws = col % a.wsegs, hb = (col / a.wsegs) % a.hblocks, dp = (col / (a.wsegs * a.hblocks)) % a.dparts;
    const int64_t n = col / (a.wsegs * a.hblocks * a.dparts);
    const int cgb = pair / a.cx_blocks, cxb = pair - cgb * a.cx_blocks;
    const int cg0 = cgb * 32, cx0 = cxb * 32;
    const int h0 = hb * Cfg::TH, w0 = ws * 32;
    const int d0 = dp * a.dchunk, d1 = d0 + a.dchunk < a.D ? d0 + a.dchunk : a.D;
    const float sx = x3wg_scale(x3wg_amax(a.amax_x)), sg = x3wg_scale(x3wg_amax(a.amax_g));
    const int64_t hw = (int64_t)a.H * a.W, dhw = hw * a.D;

    // staging tables (plane-invariant): element offset inside a plane of the sample, LDS position (in halves), validity
    unsigned xoff[XNIT], goff[GNIT], xoffh[A16 ? 1 : XNIT], goffh[A16 ? 1 : GNIT];
    int xdst[XNIT], gdst[GNIT];
    unsigned xok = 0, gok = 0, xokh = 0, gokh = 0;        // piece (or its first half) / second half inside the grid
#pragma unroll
    for (int it = 0; it < XNIT; ++it) {
        const int p = it * Cfg::THREADS + tid;
        const int ci = p / 48, rem = p - ci * 48, r = rem >> 3, q4 = rem & 7;
        const int h = h0 - 1 + r, w = w0 + 4 * q4;
        const bool rowok = cx0 + ci < a.Cx && (unsigned)h < (unsigned)a.H;
        const bool ok = rowok && w + (A16 ? 4 : 2) <= a.W, okh = rowok && w + 4 <= a.W;
        xoff[it] = ok ? (unsigned)(ci * dhw + (int64_t)h * a.W + w) : 0u;
        if constexpr (!A16) xoffh[it] = okh ? xoff[it] + 2u : 0u;
        xokh |= (okh ? 1u : 0u) << it;
        // 16-byte chunk (q4 >> 1) of the row sits at position chunk ^ 2 * bit 2 of the channel: conflict-free ds_read_b128 at a 64-byte stride
        xdst[it] = (r * 32 + ci) * XCOLS + (((q4 >> 1) ^ (2 * ((ci >> 2) & 1))) * 8) + (q4 & 1) * 4;
        xok |= (ok ? 1u : 0u) << it;
    }
#pragma unroll
    for (int it = 0; it < GNIT; ++it) {
        const int p = it * Cfg::THREADS + tid;
        const int co = p / 40, rem = p - co * 40, r = rem / 10, q = rem - r * 10;
        const int h = h0 + r, w = w0 - 4 + 4 * q;
        const bool in = p < Cfg::GPIECES;
        const bool rowok = in && cg0 + co < a.Cg && h < a.H && w >= 0;
        const bool ok = rowok && w + (A16 ? 4 : 2) <= a.W, okh = rowok && w + 4 <= a.W;
        goff[it] = ok ? (unsigned)(co * dhw + (int64_t)h * a.W + w) : 0u;
        if constexpr (!A16) goffh[it] = okh ? goff[it] + 2u : 0u;
        gokh |= (okh ? 1u : 0u) << it;
        gdst[it] = in ? (r * 32 + co) * GCOLS + 4 + 4 * q : -1;
        gok |= (ok ? 1u : 0u) << it;
    }
    const float *const xs = a.x + n * a.x_bs + (int64_t)cx0 * dhw;
    const float *const gs = a.g + n * a.g_bs + (int64_t)cg0 * dhw;
    // Loads are UNCONDITIONAL (an absent piece reads element 0 of the sample, a valid address) and the zeros are put in when the
    // registers are split into LDS: a predicated load -- `ok ? *p : 0` -- makes the compiler zero the destination first and wait
    // (s_waitcnt vmcnt(0)) for every earlier load before it may overwrite those registers: the six loads of a step then run one
    // after the other, each paying its whole latency (measured: 5.1 us per step instead of 2.6).
    f32x4 xv[XNIT], gv[GNIT];
    bool x_dok = false, g_dok = false;     // the planes in the registers lie inside the grid
    auto load_x = [&](int id) {           // input plane id -> registers
        x_dok = (unsigned)id < (unsigned)a.D;
        const float *b = xs + (int64_t)(x_dok ? id : 0) * hw;
#pragma unroll
        for (int it = 0; it < XNIT; ++it) xv[it] = x3wg_ld4<A16>(b, xoff[it], xoffh[A16 ? 0 : it]);
    };
    auto load_g = [&](int od) {
        g_dok = od < a.D;
        const float *b = gs + (int64_t)(g_dok ? od : 0) * hw;
#pragma unroll
        for (int it = 0; it < GNIT; ++it) gv[it] = x3wg_ld4<A16>(b, goff[it], goffh[A16 ? 0 : it]);
    };
    auto split_store = [&](char *hi_base, int lo_bytes, int dst_bytes, f32x4 v, float s) {
        const f32x4 t = v * s;
        const h4w hi = __builtin_convertvector(t, h4w);
        const f32x4 back = __builtin_convertvector(hi, f32x4);
        const h4w lo = __builtin_convertvector(t - back, h4w);
        *reinterpret_cast<h4w *>(hi_base + dst_bytes) = hi;
        *reinterpret_cast<h4w *>(hi_base + dst_bytes + lo_bytes) = lo;
    };
    // one piece of the plane held in the registers -> LDS.  `live` false (the walk's last step: nothing behind it): the same
    // instructions write into the 16 KB behind the image instead -- no branch, so that the whole step stays ONE basic block and the
    // compiler can place these ~50 VALU instructions between the step's MFMAs (a wave's own VALU fits the 8 issue cycles an MFMA
    // leaves free; staged as a separate phase they cost their full time: 0.31 ms of the layer's 0.93, measured with debug switches)
    char *const dummy = lds + Cfg::LDS_BYTES;
    auto store_x_piece = [&](int it, int id, bool live) {          // plane id lives in slot (id + 4) & 3
        char *b = live ? lds + ((id + 4) & 3) * Cfg::SLOT_BYTES : dummy;
        const int off = live ? 2 * xdst[it] : ((2 * xdst[it]) & 8191);
        split_store(b, live ? 2 * Cfg::XPLANE : 8192, off, x3wg_mask(xv[it], x_dok && ((xok >> it) & 1u), x_dok && ((xokh >> it) & 1u)), sx);
    };
    auto store_g_piece = [&](int it, int od, bool live) {
        const bool in = gdst[it] >= 0 && live;
        char *b = in ? gbase + (od & 1) * Cfg::G_BYTES : dummy;
        const int off = in ? 2 * gdst[it] : ((2 * gdst[it]) & 8191);
        split_store(b, in ? 2 * Cfg::GPLANE : 8192, off, x3wg_mask(gv[it], g_dok && ((gok >> it) & 1u), g_dok && ((gokh >> it) & 1u)), sg);
    };
    auto load_x_piece = [&](int it, const float *b) { xv[it] = x3wg_ld4<A16>(b, xoff[it], xoffh[A16 ? 0 : it]); };
    auto load_g_piece = [&](int it, const float *b) { gv[it] = x3wg_ld4<A16>(b, goff[it], goffh[A16 ? 0 : it]); };

    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4(0.0f);

    // prologue: planes d0 - 1, d0, d0 + 1 and gradient plane d0 into LDS; plane d0 + 2 and gradient plane d0 + 1 into registers
    for (int k = -1; k <= 1; ++k) {
        load_x(d0 + k);
#pragma unroll
        for (int it = 0; it < XNIT; ++it) store_x_piece(it, d0 + k, true);
    }
    load_g(d0);
#pragma unroll
    for (int it = 0; it < GNIT; ++it) store_g_piece(it, d0, true);
    load_x(d0 + 2);
    load_g(d0 + 1);
    __syncthreads();

    // byte offsets of this lane's operands inside a staged row set
    const int xlane = ((cih * 16 + i16) * XCOLS + ((kg ^ (2 * ((i16 >> 2) & 1))) * 8)) * 2;
    const int glane = ((coh * 16 + i16) * GCOLS + 8 + 8 * kg) * 2;
    static_assert(XNIT == 3 && GNIT == 3, "one piece per (row, kd) of a step: 3 x pieces behind the first row, 3 g pieces behind the second");
    for (int od = d0; od < d1; ++od) {
        const bool live = od + 1 < d1;       // the registers' planes (input od + 2, gradient od + 1) have a step that reads them
        // where the registers are refilled from: input plane od + 3, gradient plane od + 2 (read by the step after next; beyond the
        // walk or the grid the loads read a valid plane and the split writes zeros)
        const bool nx = (unsigned)(od + 3) < (unsigned)a.D, ng = od + 2 < a.D;
        const float *const xnext = xs + (int64_t)(nx ? od + 3 : 0) * hw, *const gnext = gs + (int64_t)(ng ? od + 2 : 0) * hw;
        const char *const gplane = gbase + (od & 1) * Cfg::G_BYTES + glane;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int rg = 2 * kp + rr;
            // the gradient row's three column shifts (hi and lo): kw = 0 -> g[u + 1], kw = 1 -> g[u], kw = 2 -> g[u - 1]; the shifted
            // pieces come from the aligned one and its neighbouring dwords (an unaligned ds_read_b128 is replayed at 64 cycles)
            u32x4w gb[2][3];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const char *q = gplane + (rg * 32 * GCOLS + pl * Cfg::GPLANE) * 2;
                const u32x4w b = *reinterpret_cast<const u32x4w *>(q);
                const unsigned left = *reinterpret_cast<const unsigned *>(q - 4);      // columns u - 2, u - 1
                const unsigned right = *reinterpret_cast<const unsigned *>(q + 16);    // columns u + 8, u + 9
                gb[pl][1] = b;
                gb[pl][0] = u32x4w{__builtin_amdgcn_alignbit(b[1], b[0], 16), __builtin_amdgcn_alignbit(b[2], b[1], 16),
                                   __builtin_amdgcn_alignbit(b[3], b[2], 16), __builtin_amdgcn_alignbit(right, b[3], 16)};
                gb[pl][2] = u32x4w{__builtin_amdgcn_alignbit(b[0], left, 16), __builtin_amdgcn_alignbit(b[1], b[0], 16),
                                   __builtin_amdgcn_alignbit(b[2], b[1], 16), __builtin_amdgcn_alignbit(b[3], b[2], 16)};
            }
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const char *sb = lds + ((od + kd + 3) & 3) * Cfg::SLOT_BYTES + xlane;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const char *xp = sb + (rg + kh) * 32 * XCOLS * 2;
                    const h8w xh = *reinterpret_cast<const h8w *>(xp);
                    const h8w xl = *reinterpret_cast<const h8w *>(xp + 2 * Cfg::XPLANE);
                    // three products per tap; the dependent MFMAs of one accumulator are three instructions apart
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int t = (kd * 3 + kh) * 3 + kw;
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 2 ? xl : xh, __builtin_bit_cast(h8w, gb[pr == 1 ? 1 : 0][kw]),
                                                                            acc[t], 0, 0, 0);
                        }
                    if (kh == 0) {           // one staged piece per (row, kd): split it into LDS, refill its registers for the step after next
                        if (rr == 0) {
                            store_x_piece(kd, od + 2, live);
                            load_x_piece(kd, xnext);
                        } else {
                            store_g_piece(kd, od + 1, live);
                            load_g_piece(kd, gnext);
                        }
                    }
                }
            }
        }
        x_dok = nx;
        g_dok = ng;
        // plane od + 2 / gradient plane od + 1 are in LDS; every read of plane od - 1 is done.  A bare barrier (LDS traffic only:
        // the refills stay in flight across it)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // ---- partial slab [col * 2 + kp][pair][tap][cg 32][cx 32]: D[i = x channel 4 kg + r][j = g channel i16]
    float *pp = a.partial + (((int64_t)(col * 2 + kp) * a.pairs + pair) * 27) * 1024 + (coh * 16 + i16) * 32 + cih * 16 + 4 * kg;
#pragma unroll
    for (int t = 0; t < 27; ++t) *reinterpret_cast<f32x4 *>(pp + t * 1024) = acc[t];
}

// ---- the same for 3x3x3 / stride-2 / pad-1 layers (and, roles swapped, the transposed layers): g on the small grid, x on the big one.
//   dW[kw] = sum_u g[u - (kw == 0)  + 1 ... ]: with the input row de-interleaved into its even and odd columns E[c] = x[2c], O[c] = x[2c+1]
//   kw = 1 -> E[u] * g[u],   kw = 2 -> O[u] * g[u],   kw = 0 -> x[2 ow - 1] * g[ow] = O[u] * g[u + 1]   (u = ow - 1)
// so every x operand is an aligned piece of E or O and only g needs one shifted copy.  A stride-2 layer has an eighth of the
// stride-1 layer's products per input byte: at f16 rates it is bound by the input stream, and the tile is chosen for that -- a
// workgroup owns 2 x 32 gradient voxels per plane step (5 input rows x 64 input columns x 32 channels per input plane, two new
// input planes per step into a ring of three), all 64 gradient channels: wave = (16-channel half of x) x (16-channel quarter of g),
// 27 accumulators of 16 x 16.  Partial slabs in the direct kernels' layout, one per column.
struct X3S2WgCfg {
    static constexpr int TH = 2, XR = 2 * TH + 1, THREADS = 512;
    static constexpr int XCOLS = 64;                         // halves per staged x row: E (32) then O (32), 16-byte chunks swizzled
    static constexpr int XPLANE = XR * 32 * XCOLS;
    static constexpr int SLOT_BYTES = 2 * XPLANE * 2;        // hi | lo = 40960
    static constexpr int GCOLS = 48, GCH = 64;
    static constexpr int GPLANE = TH * GCH * GCOLS;
    static constexpr int G_BYTES = 2 * GPLANE * 2;           // 24576
    static constexpr int LDS_BYTES = 3 * SLOT_BYTES + G_BYTES;      // 147456
    static constexpr int XPIECES = XR * 32 * 16, XNIT = XPIECES / THREADS;                       // float4 pieces of an input plane: 2560 = 5 rounds
    static constexpr int GPIECES = TH * GCH * 10, GNIT = (GPIECES + THREADS - 1) / THREADS;      // 1280 -> 3 rounds
    static_assert(XPIECES % THREADS == 0, "whole rounds");
};

template <bool G16>      // gradient rows 16-byte aligned (else 8-byte: Wout % 4 == 2); the input rows (2 Wout floats) always are
__global__ void __launch_bounds__(512, 1)
conv3d_wgrad_x3s2_kernel(const X3WgArgs a) {      // a.D / H / W: the SMALL grid (g); x lives on 2D x 2H x 2W
    using Cfg = X3S2WgCfg;
    constexpr int XNIT = Cfg::XNIT, GNIT = Cfg::GNIT, XCOLS = Cfg::XCOLS, GCOLS = Cfg::GCOLS;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char *const lds = reinterpret_cast<char *>(lds_f);
    char *const gbase = lds + 3 * Cfg::SLOT_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cih = wave & 1, coq = wave >> 1;             // x channels 16 cih .. +15, g channels 16 coq .. +15 of the job's 64
    const int i16 = lane & 15, kg = lane >> 4;
    const int q8 = a.njobs >> 3, r8 = a.njobs & 7, xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    const int job = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kx;
    if (kx >= q8 + (xcd < r8 ? 1 : 0)) return;
    // pairs here = (64-channel blocks of g) x (32-channel blocks of x)
    const int pair = job % a.pairs, col = job / a.pairs;
    const int ws = col % a.wsegs, hb = (col / a.wsegs) % a.hblocks, dp = (col / (a.wsegs * a.hblocks)) % a.dparts;
    const int64_t n = col / (a.wsegs * a.hblocks * a.dparts);
    const int cg64 = pair / a.cx_blocks, cxb = pair - cg64 * a.cx_blocks;
    const int cg0 = cg64 * 64, cx0 = cxb * 32;
    const int h0 = hb * Cfg::TH, w0 = ws * 32;             // small grid
    const int d0 = dp * a.dchunk, d1 = d0 + a.dchunk < a.D ? d0 + a.dchunk : a.D;
    const float sx = x3wg_scale(x3wg_amax(a.amax_x)), sg = x3wg_scale(x3wg_amax(a.amax_g));
    const int Hi = 2 * a.H, Wi = 2 * a.W, Di = 2 * a.D;
    const int64_t ghw = (int64_t)a.H * a.W, gdhw = ghw * a.D;
    const int64_t xhw = (int64_t)Hi * Wi, xdhw = xhw * Di;

    unsigned xoff[XNIT], goff[GNIT], goffh[G16 ? 1 : GNIT];
    int xdst[XNIT], gdst[GNIT];
    unsigned xok = 0, gok = 0, gokh = 0;
#pragma unroll
    for (int it = 0; it < XNIT; ++it) {
        const int p = it * Cfg::THREADS + tid;
        const int ci = p / 80, rem = p - ci * 80, r = rem >> 4, q = rem & 15;       // 5 rows x 16 float4
        const int h = 2 * h0 - 1 + r, w = 2 * w0 + 4 * q;
        const bool ok = cx0 + ci < a.Cx && (unsigned)h < (unsigned)Hi && w + 4 <= Wi;
        xoff[it] = ok ? (unsigned)(ci * xdhw + (int64_t)h * Wi + w) : 0u;
        // the float4 holds E[2q], O[2q], E[2q+1], O[2q+1]: halves 2 (q & 3) .. +1 of E chunk q >> 2 and of O chunk 4 + (q >> 2); chunk c of
        // the row sits at position c ^ ((ci >> 1) & 7): conflict-free ds_read_b128 at the 128-byte row stride
        const int sw = (ci >> 1) & 7;
        xdst[it] = (r * 32 + ci) * XCOLS + (((q >> 2) ^ sw) * 8) + 2 * (q & 3);    // E position; O: chunk + 4 -> position ^ 4 (see store_x)
        xok |= (ok ? 1u : 0u) << it;
    }
#pragma unroll
    for (int it = 0; it < GNIT; ++it) {
        const int p = it * Cfg::THREADS + tid;
        const int co = p / 20, rem = p - co * 20, r = rem / 10, q = rem - r * 10;
        const int h = h0 + r, w = w0 - 4 + 4 * q;
        const bool in = p < Cfg::GPIECES;
        const bool rowok = in && cg0 + co < a.Cg && h < a.H && w >= 0;
        const bool ok = rowok && w + (G16 ? 4 : 2) <= a.W, okh = rowok && w + 4 <= a.W;
        goff[it] = ok ? (unsigned)(co * gdhw + (int64_t)h * a.W + w) : 0u;
        if constexpr (!G16) goffh[it] = okh ? goff[it] + 2u : 0u;
        gokh |= (okh ? 1u : 0u) << it;
        gdst[it] = in ? (r * Cfg::GCH + co) * GCOLS + 4 + 4 * q : -1;
        gok |= (ok ? 1u : 0u) << it;
    }
    const float *const xs = a.x + n * a.x_bs + (int64_t)cx0 * xdhw;
    const float *const gs = a.g + n * a.g_bs + (int64_t)cg0 * gdhw;
    f32x4 xv[2][XNIT], gv[GNIT];
    bool x_dok[2] = {false, false}, g_dok = false;
    auto load_x = [&](int k, int id) {         // input plane id -> register set k (unconditional loads: see conv3d_wgrad_x3_kernel)
        x_dok[k] = (unsigned)id < (unsigned)Di;
        const float *b = xs + (int64_t)(x_dok[k] ? id : 0) * xhw;
#pragma unroll
        for (int it = 0; it < XNIT; ++it) xv[k][it] = *reinterpret_cast<const f32x4 *>(b + xoff[it]);
    };
    auto load_g = [&](int od) {
        g_dok = od < a.D;
        const float *b = gs + (int64_t)(g_dok ? od : 0) * ghw;
#pragma unroll
        for (int it = 0; it < GNIT; ++it) gv[it] = x3wg_ld4<G16>(b, goff[it], goffh[G16 ? 0 : it]);
    };
    typedef _Float16 h2w __attribute__((ext_vector_type(2)));
    auto store_x = [&](int k, int id) {        // plane id lives in slot (id + 3) % 3
        char *b = lds + ((id + 3) % 3) * Cfg::SLOT_BYTES;
#pragma unroll
        for (int it = 0; it < XNIT; ++it) {
            const f32x4 v = (x_dok[k] && ((xok >> it) & 1u)) ? xv[k][it] : f32x4(0.0f);
            const f32x4 t = v * sx;
            const h4w hi = __builtin_convertvector(t, h4w);
            const f32x4 back = __builtin_convertvector(hi, f32x4);
            const h4w lo = __builtin_convertvector(t - back, h4w);
            char *e = b + 2 * xdst[it];
            char *o = b + 2 * (xdst[it] ^ 32);                 // position ^ 4 in units of 8 halves
            *reinterpret_cast<h2w *>(e) = h2w{hi[0], hi[2]};
            *reinterpret_cast<h2w *>(o) = h2w{hi[1], hi[3]};
            *reinterpret_cast<h2w *>(e + 2 * Cfg::XPLANE) = h2w{lo[0], lo[2]};
            *reinterpret_cast<h2w *>(o + 2 * Cfg::XPLANE) = h2w{lo[1], lo[3]};
        }
    };
    auto store_g = [&]() {
#pragma unroll
        for (int it = 0; it < GNIT; ++it) {
            if (gdst[it] < 0) continue;
            const f32x4 v = x3wg_mask(gv[it], g_dok && ((gok >> it) & 1u), g_dok && ((gokh >> it) & 1u));
            const f32x4 t = v * sg;
            const h4w hi = __builtin_convertvector(t, h4w);
            const f32x4 back = __builtin_convertvector(hi, f32x4);
            const h4w lo = __builtin_convertvector(t - back, h4w);
            *reinterpret_cast<h4w *>(gbase + 2 * gdst[it]) = hi;
            *reinterpret_cast<h4w *>(gbase + 2 * (Cfg::GPLANE + gdst[it])) = lo;
        }
    };

    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4(0.0f);

    // prologue: input planes 2 d0 - 1, 2 d0, 2 d0 + 1 and gradient plane d0 into LDS
    load_x(0, 2 * d0 - 1);
    store_x(0, 2 * d0 - 1);
    load_x(0, 2 * d0);
    load_x(1, 2 * d0 + 1);
    store_x(0, 2 * d0);
    store_x(1, 2 * d0 + 1);
    load_g(d0);
    store_g();
    __syncthreads();

    const int sw = (i16 >> 1) & 7;
    const int xlaneE = ((cih * 16 + i16) * XCOLS + ((kg ^ sw) * 8)) * 2, xlaneO = ((cih * 16 + i16) * XCOLS + (((4 + kg) ^ sw) * 8)) * 2;
    const int glane = ((coq * 16 + i16) * GCOLS + 8 + 8 * kg) * 2;
    for (int od = d0; od < d1; ++od) {
        const bool more = od + 1 < d1;
        if (more) {                          // the next step's two new input planes and its gradient plane travel under this step's MFMAs
            load_x(0, 2 * od + 2);
            load_x(1, 2 * od + 3);
            load_g(od + 1);
        }
        const int s0 = (2 * od - 1 + 3) % 3;     // slot of input plane 2 od - 1 (kd = 0)
#pragma unroll 1      // (unrolled: 60-84 bytes of scratch and 0.585 -> 0.66 ms)
        for (int r = 0; r < Cfg::TH; ++r) {
            u32x4w gb[2][2];                 // [hi | lo][g[u] | g[u + 1]]
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const char *q = gbase + (r * Cfg::GCH * GCOLS + pl * Cfg::GPLANE) * 2 + glane;
                const u32x4w b = *reinterpret_cast<const u32x4w *>(q);
                const unsigned right = *reinterpret_cast<const unsigned *>(q + 16);
                gb[pl][0] = b;
                gb[pl][1] = u32x4w{__builtin_amdgcn_alignbit(b[1], b[0], 16), __builtin_amdgcn_alignbit(b[2], b[1], 16),
                                   __builtin_amdgcn_alignbit(b[3], b[2], 16), __builtin_amdgcn_alignbit(right, b[3], 16)};
            }
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                int slot = s0 + kd;
                slot = slot >= 3 ? slot - 3 : slot;
                const char *sb = lds + slot * Cfg::SLOT_BYTES;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const char *xp = sb + (2 * r + kh) * 32 * XCOLS * 2;
                    const h8w eh = *reinterpret_cast<const h8w *>(xp + xlaneE), el = *reinterpret_cast<const h8w *>(xp + xlaneE + 2 * Cfg::XPLANE);
                    const h8w oh = *reinterpret_cast<const h8w *>(xp + xlaneO), ol = *reinterpret_cast<const h8w *>(xp + xlaneO + 2 * Cfg::XPLANE);
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int t = (kd * 3 + kh) * 3 + kw;
                            const h8w av = kw == 1 ? (pr == 2 ? el : eh) : (pr == 2 ? ol : oh);
                            const h8w bv = __builtin_bit_cast(h8w, gb[pr == 1 ? 1 : 0][kw == 0 ? 1 : 0]);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[t], 0, 0, 0);
                        }
                }
            }
        }
        __syncthreads();                     // every read of planes 2 od - 1, 2 od and of the gradient plane is done
        if (more) {
            store_x(0, 2 * od + 2);          // into the slot of plane 2 od - 1
            store_x(1, 2 * od + 3);          // into the slot of plane 2 od
            store_g();
        }
        __syncthreads();
    }
    // ---- partial slab [col][pair32][tap][cg 32][cx 32] with pair32 = (2 cg64 + (coq >> 1)) * cx_blocks + cxb
    const int cgb32 = 2 * cg64 + (coq >> 1);
    if (cgb32 * a.cx_blocks >= a.pairs32) return;            // a 64-channel job over a layer with an odd number of 32-channel blocks
    const int pair32 = cgb32 * a.cx_blocks + cxb;
    float *pp = a.partial + (((int64_t)col * a.pairs32 + pair32) * 27) * 1024 + ((coq & 1) * 16 + i16) * 32 + cih * 16 + 4 * kg;
#pragma unroll
    for (int t = 0; t < 27; ++t) *reinterpret_cast<f32x4 *>(pp + t * 1024) = acc[t];
}

// wgrad_reduce_kernel for the split-operand form: the same fixed-order sum, then the two power-of-two scales are taken out (exact)
__global__ void wgrad_reduce_x3_kernel(const float *__restrict__ partial, float *__restrict__ dw, int Cg, int Cx, int cx_blocks, int pairs,
                                       int P, const unsigned *__restrict__ amax_x, const unsigned *__restrict__ amax_g) {
    const unsigned ax = x3wg_amax(amax_x), ag = x3wg_amax(amax_g);        // (before any lane leaves: wave shuffles)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pstride = (int64_t)pairs * 27 * 1024;
    if (i >= pstride) return;
    const int cxl = (int)(i & 31), cgl = (int)((i >> 5) & 31);
    const int tap = (int)((i >> 10) % 27);
    const int pair = (int)(i / ((int64_t)27 * 1024));
    const int cg = (pair / cx_blocks) * 32 + cgl, cx = (pair % cx_blocks) * 32 + cxl;
    if (cg >= Cg || cx >= Cx) return;
    const float *src = partial + i;
    // sixteen independent chains (slab p goes to chain p % 16), folded in a fixed tree: deterministic, and sixteen loads in flight per
    // thread instead of four -- the launch has only (pairs * 27 * 1024) / 256 workgroups (108 for conv2) walking 240-480 slabs each:
    // 42.7 us at cfg4's conv2 with four chains
    float sv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) sv[k] = 0.0f;
    int p = 0;
    for (; p + 16 <= P; p += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sv[k] += src[(int64_t)(p + k) * pstride];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (p + k < P) sv[k] += src[(int64_t)(p + k) * pstride];
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
        for (int k = 0; k < w; ++k) sv[k] = sv[k] + sv[k + w];
    const float inv_x = 1.0f / x3wg_scale(ax), inv_g = 1.0f / x3wg_scale(ag);
    dw[((int64_t)cg * Cx + cx) * 27 + tap] = (sv[0] * inv_x) * inv_g;
}

constexpr int kWgradPartitions = 512;   // 2 workgroups per CU
#define SNVC_CFG(...) WgradCfg<__VA_ARGS__>

}  // namespace
}  // namespace snvc

extern "C" {

// Spatial partitions and units per XCD of the two 12-wave forms: one workgroup per CU and (pair, partition, kd) unit, one round
static void wgrad_units(int pairs, int cap, int &P, int &upx) {
    upx = snvc::device_cu_count() / 8 / 3;
    if (upx < 1) upx = 1;
    P = 8 * upx / pairs;
    if (P < 1) P = 1;           // more pairs than units: the units take several rounds
    if (P > cap) P = cap;
    if (P * pairs > 8 * upx) upx = snvc::ceil_div(P * pairs, 8);
}

// Partial slabs [partitions][pairs][taps][32][32] of the form the layer takes; which of a key's forms runs depends on the
// operands' alignment, known only at launch, so the size is the largest of the key's candidates -- each with the partition
// count its launch really uses (r1-r2 reserved 512 partitions for every form: 6x too much for the 12-wave forms).
int64_t snvc_conv3d_wgrad_workspace_bytes(const snvc_conv3d_desc *d) {
    using namespace snvc;
    if (!d || d->Cin <= 0 || d->Cout <= 0) return -1;
    const int64_t taps = (int64_t)d->ksize * d->ksize * d->ksize;
    const int64_t pairs = (int64_t)ceil_div(d->Cout, 32) * ceil_div(d->Cin, 32);
    int64_t slabs = (int64_t)kWgradPartitions * taps;                        // tap-split / K-split forms
    if (pairs <= 65535) {
        int P, upx;
        if (d->ksize == 3 && d->stride == 1 && d->dilation == 1) {            // Winograd-domain form: 2P row-half slabs x 54
            wgrad_units((int)pairs, kWgradPartitions / 2, P, upx);
            if ((int64_t)2 * P * 54 > slabs) slabs = (int64_t)2 * P * 54;
        } else if (d->ksize == 3 && d->stride == 2 && d->dilation == 1) {     // stride-2 form: 4P slabs x 27
            wgrad_units((int)pairs, kWgradPartitions / 4, P, upx);
            if ((int64_t)4 * P * 27 > slabs) slabs = (int64_t)4 * P * 27;
        }
    }
    return slabs * pairs * 1024 * (int64_t)sizeof(float) + 1024;     // + the two amax words of the split-operand form (r6), at the end
}

// desc describes the FORWARD Conv3d (x = its input on the big grid, g = gradient of its output on
// the small grid, dw [Cout][Cin][k^3]).  For a ConvTranspose3d(k3,s2,p1,op1) pass the equivalent
// stride-2 convolution with roles swapped: x := gradient of the deconv's output, g := the deconv's
// input, desc = {Cin := deconv Cout, Cout := deconv Cin, big grid, small grid, k3, s2, p1}: dw is
// then in nn.ConvTranspose3d's [Cin][Cout][27] layout.
int snvc_conv3d_wgrad(const snvc_conv3d_desc *d, const float *x, const float *g, float *dw, void *workspace,
                      void *stream) {
    return snvc_conv3d_wgrad_amax(d, x, g, dw, workspace, nullptr, nullptr, stream);
}

int snvc_conv3d_wgrad_amax(const snvc_conv3d_desc *d, const float *x, const float *g, float *dw, void *workspace,
                           const uint32_t *amax_x, const uint32_t *amax_g, void *stream) {
    using namespace snvc;
    if (!d) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_wgrad: null desc");
    if (d->transposed) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_wgrad: describe the equivalent strided convolution (see header)");
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_wgrad: sizes must be positive");
    if (d->pad != d->dilation * (d->ksize - 1) / 2) return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_wgrad: pad must equal dilation*(ksize-1)/2");
    const int eff = d->dilation * (d->ksize - 1) + 1;
    if (d->Dout != (d->Din + 2 * d->pad - eff) / d->stride + 1 || d->Hout != (d->Hin + 2 * d->pad - eff) / d->stride + 1 ||
        d->Wout != (d->Win + 2 * d->pad - eff) / d->stride + 1)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_wgrad: output size does not match the convolution arithmetic");
    if (!x || !g || !dw || !workspace) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_wgrad: null pointer");
    const int64_t in_sz = (int64_t)d->Cin * d->Din * d->Hin * d->Win, out_sz = (int64_t)d->Cout * d->Dout * d->Hout * d->Wout;
    if (in_sz >= ((int64_t)1 << 31) || out_sz >= ((int64_t)1 << 31))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_wgrad: one sample must stay below 2^31 elements");

    WgradArgs a;
    a.x = x; a.g = g; a.partial = (float *)workspace;
    a.N = d->N; a.Cx = d->Cin; a.Di = d->Din; a.Hi = d->Hin; a.Wi = d->Win;
    a.Cg = d->Cout; a.Do = d->Dout; a.Ho = d->Hout; a.Wo = d->Wout;
    const int th = d->stride == 1 ? 2 : 1;   // rows per tile (see the WgradCfg instantiations below)
    a.tiles_h = ceil_div(d->Hout, th); a.tiles_w = ceil_div(d->Wout, 32);
    a.ntiles = (int64_t)d->N * d->Dout * a.tiles_h * a.tiles_w;
    a.P = kWgradPartitions;
    a.cx_blocks = ceil_div(d->Cin, 32);
    a.x_bs = d->x_batch_stride ? d->x_batch_stride : in_sz;
    a.g_bs = d->y_batch_stride ? d->y_batch_stride : out_sz;
    // float4 staging of X needs 16-byte rows on the big grid; G rows may be 16- or 8-byte ones (the W = 78 level of the
    // cfg2 hourglass: its two layers took the scalar path at 1.0 ms each)
    const bool xv = d->Win % 4 == 0 && a.x_bs % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                    !(d->algo & SNVC_ALGO_SCALAR_STAGING);
    const bool g4 = d->Wout % 4 == 0 && a.g_bs % 4 == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0;
    const bool g2 = d->Wout % 2 == 0 && a.g_bs % 2 == 0 && (reinterpret_cast<uintptr_t>(g) & 7) == 0;
    a.vec = xv ? (g4 ? 4 : (g2 ? 2 : 0)) : 0;
    const int pairs = ceil_div(d->Cout, 32) * a.cx_blocks;
    if (pairs > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_wgrad: too many channel pairs");
    hipStream_t st = as_stream(stream);
    {   // 1x1x1 to <= 2 channels: streaming dot products (see wgrad_k1_small_partial)
        const int64_t S = (int64_t)d->Dout * d->Hout * d->Wout;
        const int64_t chunk4 = 16384;                                 // 64k voxels per workgroup (x four channels, r6; 8192 measured slower)
        const int64_t chunks = ceil_div<int64_t>(S / 4, chunk4);
        if (d->ksize == 1 && d->stride == 1 && d->Cout <= 2 && S % 4 == 0 && a.x_bs % 4 == 0 && a.g_bs % 4 == 0 &&
            ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15) == 0 && d->Cin <= 65535 && d->N <= 65535 &&
            chunks * d->N * d->Cin * d->Cout * 4 <= snvc_conv3d_wgrad_workspace_bytes(d)) {
            dim3 grid((unsigned)chunks, (unsigned)ceil_div(d->Cin, 4), (unsigned)d->N);       // four input channels per workgroup
            if (d->Cout == 1)
                wgrad_k1_small_partial<1><<<grid, 256, 0, st>>>(x, g, (float *)workspace, d->Cin, S / 4, chunk4, a.x_bs, a.g_bs);
            else
                wgrad_k1_small_partial<2><<<grid, 256, 0, st>>>(x, g, (float *)workspace, d->Cin, S / 4, chunk4, a.x_bs, a.g_bs);
            wgrad_k1_small_final<<<ceil_div(d->Cin * d->Cout, 64), 64, 0, st>>>((const float *)workspace, dw, d->Cin, d->Cout,
                                                                               chunks * d->N);
            return check_launch("snvc_conv3d_wgrad(k1 small)");
        }
    }
    const int key = d->ksize * 100 + d->stride * 10 + d->dilation;
    // rows of the split-operand forms: 16-byte aligned, or 8-byte (W % 4 == 2: float2 pieces)
    const bool x8 = d->Win % 2 == 0 && a.x_bs % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0 && !(d->algo & SNVC_ALGO_SCALAR_STAGING);
    // (their own amax pass reads the tensors as flat float4 streams)
    const bool flat16 = in_sz % 4 == 0 && out_sz % 4 == 0 && a.x_bs % 4 == 0 && a.g_bs % 4 == 0 &&
                        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    if (key == 311 && flat16 && (a.vec == 4 || (x8 && g2)) && !(d->algo & SNVC_ALGO_WGRAD_FP32) && (d->algo & SNVC_ALGO_ARITH_MASK) != SNVC_ALGO_DIRECT) {
        // split-operand (f16x3) form, see conv3d_wgrad_x3_kernel: one workgroup per (column of 4 x 32 voxels, depth part, channel pair)
        X3WgArgs b;
        b.x = x; b.g = g; b.partial = (float *)workspace;
        b.N = d->N; b.Cx = d->Cin; b.Cg = d->Cout; b.D = d->Dout; b.H = d->Hout; b.W = d->Wout;
        b.cx_blocks = a.cx_blocks; b.pairs = pairs; b.pairs32 = pairs; b.x_bs = a.x_bs; b.g_bs = a.g_bs;
        b.hblocks = ceil_div(d->Hout, X3WgCfg::TH); b.wsegs = ceil_div(d->Wout, 32);
        const int64_t cols = (int64_t)d->N * b.hblocks * b.wsegs;
        // depth parts: enough jobs for every CU once, columns at least 8 planes long, at most 256 columns x parts (512 slabs, the
        // workspace's size)
        int dparts = 1;
        while (cols * dparts * pairs < device_cu_count() * 3 / 4 && d->Dout / (dparts * 2) >= 8 && cols * dparts * 2 <= 256) dparts *= 2;
        const int64_t slabs_bytes = snvc_conv3d_wgrad_workspace_bytes(d) - 1024;
        if (cols * dparts <= 256 && cols * dparts * pairs < ((int64_t)1 << 24) &&
            (int64_t)2 * cols * dparts * pairs * 27 * 1024 * 4 <= slabs_bytes) {
            b.dparts = dparts; b.dchunk = ceil_div(d->Dout, dparts);
            b.dparts = ceil_div(d->Dout, b.dchunk);
            b.njobs = (int)(cols * b.dparts * pairs);
            unsigned *amax = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + slabs_bytes);
            b.amax_x = amax_x ? amax_x : amax;
            b.amax_g = amax_g ? amax_g : amax + SNVC_AMAX_SLOTS;
            if (!amax_x || !amax_g) {        // a maximum the caller did not bring: one more pass over that tensor
                if (hipMemsetAsync(amax, 0, 2 * SNVC_AMAX_SLOTS * 4, st) != hipSuccess) return fail(SNVC_ERR_HIP, "snvc_conv3d_wgrad: hipMemsetAsync failed");
                const int64_t n4x = amax_x ? 0 : in_sz / 4, n4g = amax_g ? 0 : out_sz / 4;
                const unsigned ab = (unsigned)std::min<int64_t>(ceil_div<int64_t>(std::max(n4x, n4g), 256 * 8), 4096);
                wgrad_amax2_kernel<<<dim3(ab, 2, (unsigned)d->N), 256, 0, st>>>(x, g, amax, n4x, n4g, a.x_bs, a.g_bs);
            }
            static std::atomic<unsigned> attr_x3{0}, attr_x3u{0};
            const unsigned nwg = (unsigned)(8 * ceil_div(b.njobs, 8));
            if (a.vec == 4) {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_wgrad_x3_kernel<true>), X3WgCfg::LDS_ALLOC, attr_x3))
                    conv3d_wgrad_x3_kernel<true><<<dim3(nwg), X3WgCfg::THREADS, X3WgCfg::LDS_ALLOC, st>>>(b);
            } else {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_wgrad_x3_kernel<false>), X3WgCfg::LDS_ALLOC, attr_x3u))
                    conv3d_wgrad_x3_kernel<false><<<dim3(nwg), X3WgCfg::THREADS, X3WgCfg::LDS_ALLOC, st>>>(b);
            }
            int rcx = check_launch("snvc_conv3d_wgrad(split operands)");
            if (rcx) return rcx;
            const int64_t total = (int64_t)pairs * 27 * 1024;
            wgrad_reduce_x3_kernel<<<dim3((unsigned)ceil_div<int64_t>(total, 256)), 256, 0, st>>>(
                (const float *)workspace, dw, d->Cout, d->Cin, a.cx_blocks, pairs, (int)(2 * cols * b.dparts), b.amax_x, b.amax_g);
            return check_launch("snvc_conv3d_wgrad(split operands reduce)");
        }
    }
    const int64_t wino_tiles = (int64_t)d->N * d->Dout * ceil_div(d->Hout, WinoWgradCfg::TH) * a.tiles_w;   // 32-bit tile counter
    if (key == 311 && a.vec == 4 && wino_tiles < ((int64_t)1 << 30) && (d->algo & SNVC_ALGO_ARITH_MASK) != SNVC_ALGO_DIRECT) {
        // Winograd-domain form (see conv3d_wgrad_wino_kernel): half the MFMAs of the direct form
        a.tiles_h = ceil_div(d->Hout, WinoWgradCfg::TH);
        a.ntiles = (int64_t)d->N * d->Dout * a.tiles_h * a.tiles_w;
        // one 12-wave workgroup per CU, one round: upx (pair, partition) units x 3 depth slices on each of the 8 XCDs
        wgrad_units(pairs, kWgradPartitions / 2, a.P, a.upx);
        a.pairs = pairs;
        const unsigned nwg = (unsigned)(8 * a.upx * 3);
        static std::atomic<unsigned> attr_done{0};
        constexpr int bytes = WinoWgradCfg::LDS_FLOATS * 4;
        if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_wgrad_wino_kernel), bytes, attr_done))
            conv3d_wgrad_wino_kernel<<<dim3(nwg), WinoWgradCfg::THREADS, bytes, st>>>(a);
        int rcw = check_launch("snvc_conv3d_wgrad(winograd)");
        if (rcw) return rcw;
        const int64_t total = (int64_t)pairs * 9 * 1024;
        wgrad_wino_reduce_kernel<<<dim3((unsigned)ceil_div<int64_t>(total, 256)), 256, 0, st>>>((const float *)workspace, dw, d->Cout,
                                                                                                 d->Cin, a.cx_blocks, pairs, 2 * a.P);
        return check_launch("snvc_conv3d_wgrad(winograd reduce)");
    }
    if (key == 321 && flat16 && (a.vec == 4 || a.vec == 2) && !(d->algo & SNVC_ALGO_WGRAD_FP32) && (d->algo & SNVC_ALGO_ARITH_MASK) != SNVC_ALGO_DIRECT &&
        d->Din == 2 * d->Dout && d->Hin == 2 * d->Hout && d->Win == 2 * d->Wout) {
        // split-operand (f16x3) form of the stride-2 layers, see conv3d_wgrad_x3s2_kernel
        X3WgArgs b;
        b.x = x; b.g = g; b.partial = (float *)workspace;
        b.N = d->N; b.Cx = d->Cin; b.Cg = d->Cout; b.D = d->Dout; b.H = d->Hout; b.W = d->Wout;
        b.cx_blocks = a.cx_blocks; b.pairs32 = pairs; b.pairs = ceil_div(d->Cout, 64) * a.cx_blocks; b.x_bs = a.x_bs; b.g_bs = a.g_bs;
        b.hblocks = ceil_div(d->Hout, X3S2WgCfg::TH); b.wsegs = ceil_div(d->Wout, 32);
        const int64_t cols = (int64_t)d->N * b.hblocks * b.wsegs;
        int dparts = 1;
        while (cols * dparts * b.pairs < device_cu_count() * 3 / 4 && d->Dout / (dparts * 2) >= 8 && cols * dparts * 2 <= 512) dparts *= 2;
        const int64_t slabs_bytes = snvc_conv3d_wgrad_workspace_bytes(d) - 1024;
        if (cols * dparts <= 512 && cols * dparts * b.pairs < ((int64_t)1 << 24) &&
            cols * dparts * pairs * 27 * 1024 * 4 <= slabs_bytes) {
            b.dchunk = ceil_div(d->Dout, dparts);
            b.dparts = ceil_div(d->Dout, b.dchunk);
            b.njobs = (int)(cols * b.dparts * b.pairs);
            unsigned *amax = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + slabs_bytes);
            b.amax_x = amax_x ? amax_x : amax;
            b.amax_g = amax_g ? amax_g : amax + SNVC_AMAX_SLOTS;
            if (!amax_x || !amax_g) {
                if (hipMemsetAsync(amax, 0, 2 * SNVC_AMAX_SLOTS * 4, st) != hipSuccess) return fail(SNVC_ERR_HIP, "snvc_conv3d_wgrad: hipMemsetAsync failed");
                const int64_t n4x = amax_x ? 0 : in_sz / 4, n4g = amax_g ? 0 : out_sz / 4;
                const unsigned ab = (unsigned)std::min<int64_t>(ceil_div<int64_t>(std::max(n4x, n4g), 256 * 8), 4096);
                wgrad_amax2_kernel<<<dim3(ab, 2, (unsigned)d->N), 256, 0, st>>>(x, g, amax, n4x, n4g, a.x_bs, a.g_bs);
            }
            static std::atomic<unsigned> attr_x3s2{0}, attr_x3s2u{0};
            const unsigned nwg = (unsigned)(8 * ceil_div(b.njobs, 8));
            if (a.vec == 4) {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_wgrad_x3s2_kernel<true>), X3S2WgCfg::LDS_BYTES, attr_x3s2))
                    conv3d_wgrad_x3s2_kernel<true><<<dim3(nwg), X3S2WgCfg::THREADS, X3S2WgCfg::LDS_BYTES, st>>>(b);
            } else {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_wgrad_x3s2_kernel<false>), X3S2WgCfg::LDS_BYTES, attr_x3s2u))
                    conv3d_wgrad_x3s2_kernel<false><<<dim3(nwg), X3S2WgCfg::THREADS, X3S2WgCfg::LDS_BYTES, st>>>(b);
            }
            int rcx = check_launch("snvc_conv3d_wgrad(split operands, stride 2)");
            if (rcx) return rcx;
            const int64_t total = (int64_t)pairs * 27 * 1024;
            wgrad_reduce_x3_kernel<<<dim3((unsigned)ceil_div<int64_t>(total, 256)), 256, 0, st>>>(
                (const float *)workspace, dw, d->Cout, d->Cin, a.cx_blocks, pairs, (int)(cols * b.dparts), b.amax_x, b.amax_g);
            return check_launch("snvc_conv3d_wgrad(split operands, stride 2, reduce)");
        }
    }
    const int64_t s2_tiles = (int64_t)d->N * d->Dout * ceil_div(d->Hout, S2WgradCfg::TH) * a.tiles_w;
    if (key == 321 && flat16 && (a.vec == 4 || a.vec == 2) && s2_tiles < ((int64_t)1 << 30) &&
        (d->algo & SNVC_ALGO_ARITH_MASK) != SNVC_ALGO_DIRECT) {
        // 12-wave form (see conv3d_wgrad_s2_kernel); SNVC_ALGO_DIRECT keeps the tap-split kernel below
        a.tiles_h = ceil_div(d->Hout, S2WgradCfg::TH);
        a.ntiles = s2_tiles;
        wgrad_units(pairs, kWgradPartitions / 4, a.P, a.upx);
        a.pairs = pairs;
        const unsigned nwg = (unsigned)(8 * a.upx * 3);
        constexpr int bytes = S2WgradCfg::LDS_FLOATS * 4;
        static std::atomic<unsigned> attr4{0}, attr2{0};
        if (a.vec == 4) {
            if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_wgrad_s2_kernel<4>), bytes, attr4))
                conv3d_wgrad_s2_kernel<4><<<dim3(nwg), S2WgradCfg::THREADS, bytes, st>>>(a);
        } else {
            if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_wgrad_s2_kernel<2>), bytes, attr2))
                conv3d_wgrad_s2_kernel<2><<<dim3(nwg), S2WgradCfg::THREADS, bytes, st>>>(a);
        }
        int rcs = check_launch("snvc_conv3d_wgrad(stride 2)");
        if (rcs) return rcs;
        const int64_t total = (int64_t)pairs * 27 * 1024;
        wgrad_reduce_kernel<<<dim3((unsigned)ceil_div<int64_t>(total, 256)), 256, 0, st>>>((const float *)workspace, dw, d->Cout, d->Cin, 27,
                                                                                            27, a.cx_blocks, pairs, 4 * a.P);
        return check_launch("snvc_conv3d_wgrad(stride 2 reduce)");
    }
#define SNVC_WGRAD_CASE(CFG)                                                                  \
    {                                                                                         \
        using C_ = CFG;                                                                       \
        a.tiles_h = ceil_div(d->Hout, C_::TH);                                                \
        a.ntiles = (int64_t)d->N * d->Dout * a.tiles_h * a.tiles_w;                           \
        dim3 grid(kWgradPartitions, (unsigned)pairs, (unsigned)(C_::CH_D * C_::CH_H));        \
        launch_wgrad<C_>(a, grid, st);                                                        \
    }
    switch (key) {  //                            KS S  D  TH KDG KHG
        case 111: SNVC_WGRAD_CASE(SNVC_CFG(1, 1, 1, 2, 1, 1)) break;
        case 311:   // float4-staged, register-prefetched, K-split form when the rows are 16-byte aligned
            if (a.vec) SNVC_WGRAD_CASE(SNVC_CFG(3, 1, 1, 2, 1, 3, true))
            else SNVC_WGRAD_CASE(SNVC_CFG(3, 1, 1, 2, 3, 3))
            break;
        case 321:   // same form for the stride-2 layers (one output row per tile keeps it inside 256 VGPRs)
            if (a.vec) SNVC_WGRAD_CASE(SNVC_CFG(3, 2, 1, 1, 1, 3, true))
            else SNVC_WGRAD_CASE(SNVC_CFG(3, 2, 1, 1, 3, 3))
            break;
        case 511: SNVC_WGRAD_CASE(SNVC_CFG(5, 1, 1, 2, 1, 5)) break;
        case 512: SNVC_WGRAD_CASE(SNVC_CFG(5, 1, 2, 2, 1, 5)) break;
        case 711: SNVC_WGRAD_CASE(SNVC_CFG(7, 1, 1, 2, 1, 4)) break;
        default:
            return fail(SNVC_ERR_UNSUPPORTED,
                        "snvc_conv3d_wgrad: (ksize,stride,dilation) not in {(1,1,1),(3,1,1),(3,2,1),(5,1,1),(5,1,2),(7,1,1)}");
    }
#undef SNVC_WGRAD_CASE
    int rc = check_launch("snvc_conv3d_wgrad");
    if (rc) return rc;
    const int taps = d->ksize * d->ksize * d->ksize;
    const int64_t total = (int64_t)pairs * taps * 1024;
    wgrad_reduce_kernel<<<dim3((unsigned)ceil_div<int64_t>(total, 256)), 256, 0, st>>>(
        (const float *)workspace, dw, d->Cout, d->Cin, taps, taps, a.cx_blocks, pairs, kWgradPartitions);
    return check_launch("snvc_conv3d_wgrad(reduce)");
}

}  // extern "C"
