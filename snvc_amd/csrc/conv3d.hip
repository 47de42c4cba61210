// 3D convolution / transposed convolution with fused epilogue for gfx950
// (SURVEY.md section 8 rows a4-a7).
//
// What the reference does: nn.Conv3d / nn.ConvTranspose3d through cuDNN, then separate
// BatchNorm3d, residual-add and ReLU kernels (snvc/models/submodule.py:32-50,85-268;
// snvc/models/vernier.py:249-295,414-438).
//
// What this file does instead -- a direct (im2col-free) convolution shaped for CDNA4:
//   * GEMM view per workgroup:  Y^T[Cout x voxels] = W[Cout x (Cin*taps)] * X[(Cin*taps) x voxels].
//     The MFMA's A operand is the weight (rows = output channels), the B operand is the
//     activation (columns = 32 consecutive output voxels along W).  With NCDHW tensors this
//     orientation makes every operand access unit-stride: B fragments are 32 consecutive
//     floats of an LDS row, and each accumulator register is a 128-byte contiguous run of one
//     output channel, so the epilogue stores whole cache lines without a transpose.
//   * v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate (no reduced precision anywhere; the
//     direct kernels are an exact fp32 FMA chain per output); one VGPR per operand per lane.
//   * The input tile (+halo) of KC input channels is staged once in LDS and re-used by all
//     k^3 taps: a tap is only a constant LDS address offset.  Zero padding is materialised in
//     LDS, so the inner loop has no bounds checks.
//   * Weights are pre-packed (snvc_conv3d_pack_weights) into the exact order the waves
//     stream them, fragment by fragment, from L2.
//   * Epilogue fused in registers: per-channel affine (folded eval BatchNorm), residual add
//     before or after the activation, ReLU / Sigmoid, channel-sliced output (builds the
//     torch.cat of vernier.py:433 in place).
//   * ConvTranspose3d(k3,s2,p1,op1) is computed as 8 parity-class sub-convolutions over the
//     INPUT grid (27 taps per 8 outputs; no zero insertion): a workgroup owns one
//     (depth-parity, height-parity) class, keeps both width parities in registers and
//     stores them interleaved as 8-byte pairs.
//   * blockIdx -> tile mapping is XCD-aware: the 8 XCDs (round-robin dispatch) each get a
//     contiguous run of tiles so that halo re-reads hit that XCD's L2.
//
// Kernel families in this file (the dispatcher in conv3d_forward_impl picks one per launch):
//   conv3d_mfma_kernel      direct form, any k in {1,3,5,7}, stride 1/2, dilation 1/2; also what
//                           desc.algo = SNVC_ALGO_DIRECT and training's k5/k7 layers run on
//   conv3d_wino_kernel /    Winograd F(4,3) along W for k3 / stride 1 (register-staged, and LDS-DMA
//   conv3d_wino_dma_kernel  staged: the default 4x4x32 tile at three workgroups per CU)
//   conv3d_winok_kernel     F(4,5) / F(4,7) for k5 / k7 (also dilation 2 for k5, polyphase) and the
//                           polyphase + F(4,2) form of k3 / stride 2
//   deconv3d_mfma_kernel    ConvTranspose3d(k3,s2,p1,op1), one parity class per workgroup, optional
//                           fused 1x1x1 head (snvc_conv3d_forward_head)
//   pointwise_small_kernel, conv3d_k3_cout1_kernel   VALU kernels for layers with 1-2 output channels
#include "common.hpp"
#include "conv3d_internal.hpp"
#include "elementwise_internal.hpp"
#include "wino_tables.hpp"

namespace snvc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native 16-byte vector (keeps staging arrays in registers)

struct ConvArgs {
    const float *x;
    const float *wp;
    const float *scale;
    const float *bias;
    const float *res;
    const float *plane;   // optional [N][Cout][3][Hout][Wout] depth-class planes added before the affine
    const float *wp_wino; // Winograd-packed weights of a k3/s1 layer (behind the direct packing)
    int nchunks_wino;
    float *y;
    int Cin, Din, Hin, Win;
    int Cout, Dout, Hout, Wout;
    int tiles_d, tiles_h, tiles_w;
    int nchunks, flags;
    const float *head_w; // fused 1x1x1 head (snvc_conv3d_forward_head): [Cout] weights, or nullptr
    float *y_head;       //   its [N,1,Dout,Hout,Wout] output; `y` is then not written
    int fast_epi;        // the launch qualifies for the fast epilogues (see the toolkit comment)
    int dc_planar;       // deconv3d_mfma_kernel: depth-1 layer, only the classes (pd = 0, ph) are launched (2 * ntiles workgroups)
    double *stats;       // XMODE 3: per job [32 channels][sum, sum of squares] of the raw result (train-mode BatchNorm), or nullptr
    int njobs, groups;   // Winograd kernel: jobs = tiles x channel groups x samples
    int vec;  // 1: 16-byte aligned rows (Win % 4 == 0, aligned base and strides) -> float4 staging
    int64_t x_bs, y_bs, r_bs;
};

__device__ __forceinline__ float epilogue_f(float v, float res, int flags) {
    if (flags & SNVC_EPI_ADD_PRE) v += res;
    if (flags & SNVC_EPI_RELU) v = v > 0.0f ? v : 0.0f;
    if (flags & SNVC_EPI_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    if (flags & SNVC_EPI_ADD_POST) v += res;
    return v;
}

// Bijective XCD-aware remap (cdna guide T1): blocks b and b+8 share an XCD; give each XCD a
// contiguous range of logical tiles.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, k = b >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// ------------------------------------------------------------------------------------ staging
// The LDS image of one chunk is [KC][IN_D][IN_H][IN_WV] floats.  IN_WV is the tile's input row
// widened to 16-byte-aligned global columns: column 0 of the image is global column
// (first output column)*STRIDE - LPAD with LPAD = PAD rounded up to a multiple of 4, so that a
// row is RQ = IN_WV/4 aligned float4 pieces, each either entirely inside [0, Win) or entirely
// padding when Win % 4 == 0.  Item i (= 16-byte piece i) lands at LDS float 4*i: lanes write
// consecutive 16-byte slots (no bank conflicts) and the geometry of an item -- its global
// offset and whether it is padding -- does not depend on the channel chunk, so it is computed
// ONCE per thread and kept in registers.  Loads are unconditional (padding items read offset 0
// and are zeroed by a select): no branches, no per-item waits; the whole chunk is in flight at
// once, and it is issued one chunk AHEAD of its use (register prefetch).
template <int KC_, int IN_D_, int IN_H_, int IN_WV_, int PIECE_ = 4>
struct Stager {
    // PIECE = floats per staged piece: 4 (16-byte pieces; rows with Win % 4 == 0) or 2 (8-byte pieces
    // for rows that are only 8-byte aligned, e.g. the W = 78 level of the cfg2 hourglass)
    static constexpr int KC = KC_, IN_D = IN_D_, IN_H = IN_H_, IN_WV = IN_WV_, PIECE = PIECE_;
    static constexpr int RQ = IN_WV / PIECE;
    static constexpr int ROWS = KC * IN_D * IN_H;
    static constexpr int ITEMS = ROWS * RQ;
    static constexpr int NIT = (ITEMS + 255) / 256;
    static constexpr int CH = IN_D * IN_H * IN_WV;   // floats per channel
    static constexpr int TILE = KC * CH;             // floats per chunk image
    typedef float Vec __attribute__((ext_vector_type(PIECE)));
    static_assert(PIECE == 4 || PIECE == 2, "pieces are 16 or 8 bytes");
    static_assert(IN_WV % 4 == 0, "image rows are whole float4 pieces");
    static_assert(NIT <= 32, "validity mask is one 32-bit register");

    unsigned off[NIT];   // float offset of the piece relative to (sample base + c0*in_dhw)
    unsigned kcs;        // 2 bits per item would not fit every config: kc is recomputed when needed
    unsigned vmask;      // bit it: piece is inside the tensor (else padding -> zeros)

    __device__ __forceinline__ void init(int tid, int id0, int ih0, int ix0, int Din, int Hin, int Win,
                                         int in_hw, int in_dhw) {
        vmask = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * 256 + tid;
            const int row = i / RQ, q = i - row * RQ;
            const int kc = row / (IN_D * IN_H), r2 = row - kc * (IN_D * IN_H);
            const int dd = r2 / IN_H, hh = r2 - dd * IN_H;
            const int gd = id0 + dd, gh = ih0 + hh, gw = ix0 + PIECE * q;
            const bool ok = i < ITEMS && (unsigned)gd < (unsigned)Din && (unsigned)gh < (unsigned)Hin &&
                            (unsigned)gw < (unsigned)Win;
            off[it] = ok ? (unsigned)(kc * in_dhw + gd * in_hw + gh * Win + gw) : 0u;
            vmask |= (ok ? 1u : 0u) << it;
        }
    }

    // channels [c0, c0+KC) -> registers.  Only ISSUES the loads (padding pieces read offset 0, and so
    // do pieces of channels >= Cin when the last chunk is partial: nothing outside the tensor is ever
    // touched); nothing here consumes the data, so the wave does not wait for it.
    __device__ __forceinline__ void load(const float *__restrict__ xc, int tid, int cin_left, Vec (&v)[NIT]) const {
        if (cin_left >= KC) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) v[it] = *reinterpret_cast<const Vec *>(xc + off[it]);
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const bool chan_ok = (it * 256 + tid) / (RQ * IN_D * IN_H) < cin_left;
                v[it] = *reinterpret_cast<const Vec *>(xc + (chan_ok ? off[it] : 0u));
            }
        }
    }

    // registers -> LDS image; padding pieces and channels >= Cin (cin_left = Cin - c0 < KC) become
    // zeros here, by a select on the way out.
    __device__ __forceinline__ void store(float *__restrict__ buf, int tid, int cin_left, const Vec (&v)[NIT]) const {
        const unsigned m = store_mask(tid, cin_left);
#pragma unroll
        for (int it = 0; it < NIT; ++it) store_one(buf, tid, it, m, v[it]);
    }

    // The same store, one piece at a time (for kernels that interleave it with their MFMA stream).
    __device__ __forceinline__ unsigned store_mask(int tid, int cin_left) const {
        unsigned m = vmask;
        if (cin_left < KC) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if ((it * 256 + tid) / (RQ * IN_D * IN_H) >= cin_left) m &= ~(1u << it);
        }
        return m;
    }
    static __device__ __forceinline__ void store_one(float *__restrict__ buf, int tid, int it, unsigned m, const Vec &v) {
        const int i = it * 256 + tid;
        Vec z = v;
        if (!((m >> it) & 1u)) z = Vec(0.0f);
        if (ITEMS % 256 == 0 || i < ITEMS) reinterpret_cast<Vec *>(buf)[i] = z;
    }

    // Fallback for tensors whose rows are not 16-byte aligned (Win % 4 != 0, odd strides):
    // same LDS image, one float at a time, synchronous, still branch-free per element.
    static __device__ __forceinline__ void stage_scalar(const float *__restrict__ xc, float *__restrict__ buf,
                                                        int tid, int id0, int ih0, int ix0, int Din, int Hin,
                                                        int Win, int in_hw, int in_dhw, int cin_left) {
        constexpr int BATCH = 8;
        for (int e0 = 0; e0 < TILE; e0 += 256 * BATCH) {
            float v[BATCH];
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const int e = e0 + b * 256 + tid;
                const int kc = e / CH, rem = e - kc * CH;
                const int dd = rem / (IN_H * IN_WV), rem2 = rem - dd * (IN_H * IN_WV);
                const int hh = rem2 / IN_WV, ww = rem2 - hh * IN_WV;
                const int gd = id0 + dd, gh = ih0 + hh, gw = ix0 + ww;
                const bool ok = e < TILE && kc < cin_left && (unsigned)gd < (unsigned)Din &&
                                (unsigned)gh < (unsigned)Hin && (unsigned)gw < (unsigned)Win;
                const float t = xc[ok ? (kc * in_dhw + gd * in_hw + gh * Win + gw) : 0];
                v[b] = ok ? t : 0.0f;
            }
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const int e = e0 + b * 256 + tid;
                if (e < TILE) buf[e] = v[b];
            }
        }
    }
};

// Packed weights of one chunk are a contiguous run of WF floats in exactly the order the
// fragments are consumed, so staging them is a linear float4 copy global -> registers -> LDS,
// prefetched together with the input image.  Keeping the A fragments in LDS (and not in
// per-wave global loads) leaves the vector-memory queue to the prefetch alone: the MFMA loop
// only ever waits on LDS reads (lgkmcnt), and the one vmcnt wait sits at the END of a chunk.
template <int WF_>
struct WeightStager {
    static constexpr int WF = WF_;
    static constexpr int ITEMS = WF / 4;
    static constexpr int NIT = (ITEMS + 255) / 256;
    static_assert(WF % 4 == 0, "weight chunk is whole float4 pieces");
    static __device__ __forceinline__ void load(const float *__restrict__ wc, int tid, f32x4 (&v)[NIT]) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * 256 + tid;
            v[it] = reinterpret_cast<const f32x4 *>(wc)[(ITEMS % 256 == 0 || i < ITEMS) ? i : 0];
        }
    }
    static __device__ __forceinline__ void store(float *__restrict__ wbuf, int tid, const f32x4 (&v)[NIT]) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) store_one(wbuf, tid, it, v[it]);
    }
    static __device__ __forceinline__ void store_one(float *__restrict__ wbuf, int tid, int it, const f32x4 &v) {
        const int i = it * 256 + tid;
        if (ITEMS % 256 == 0 || i < ITEMS) reinterpret_cast<f32x4 *>(wbuf)[i] = v;
    }
};

// Epilogue shared by conv and deconv: acc register r of lane l is output channel
// cbase + (r&3) + 8*(r>>2) + 4*(l>>5); scale / bias for the lane's 16 channels are fetched
// together, residual values are fetched as one batch per accumulator (clamped addresses, no
// branches) and the stores are predicated.
struct ChanAffine {
    float sc[16], bi[16];
};

__device__ __forceinline__ void load_affine(const ConvArgs &a, int cbase, int lane, ChanAffine &f) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        co = co < a.Cout ? co : a.Cout - 1;
        f.sc[r] = a.scale ? a.scale[co] : 1.0f;
        f.bi[r] = a.scale ? a.bias[co] : 0.0f;
    }
}

// ---- fast epilogue toolkit (whole 32-channel groups, 16-byte aligned rows, no Sigmoid; the host
// sets ConvArgs::fast_epi when a launch qualifies, everything else takes the generic epilogues).
// Two measured facts shape it:
//  * vector loads and stores retire through one in-order counter (vmcnt): a load issued behind a
//    store is not usable before that store has been acknowledged, so a load/store/load/store chain
//    costs a write round trip per link (40k cycles per workgroup in the first Winograd epilogue).
//    All loads of the epilogue are therefore issued before its first store, results are formed IN
//    PLACE in the accumulator registers, and the stores go out at the end, fire-and-forget.
//  * stores are issue-bound: a wave-wide store instruction occupies the CU's store path for several
//    hundred cycles whatever its width (32 x 8-byte stores per lane: 18k cycles), so lanes trade
//    values first (DPP, no LDS) until each owns 4 consecutive outputs of one channel = one 16-byte
//    store where it had four 4-byte or two 8-byte ones.
__device__ __forceinline__ float dpp_xor1(float v) {   // value of lane ^ 1 (quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {   // value of lane ^ 2 (quad_perm [2,3,0,1])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

// Registers a0..a3 of lanes 4j..4j+3 form a 4x4 block (register = channel c+k, lane = voxel 4j+i);
// afterwards lane 4j+i holds channel c+i, voxels 4j..4j+3 in (a0..a3).  Two butterfly stages.
__device__ __forceinline__ void quad_transpose(float &a0, float &a1, float &a2, float &a3, bool odd, bool hi) {
    {
        const float s01 = odd ? a0 : a1, s23 = odd ? a2 : a3;
        const float g01 = dpp_xor1(s01), g23 = dpp_xor1(s23);
        a0 = odd ? g01 : a0; a1 = odd ? a1 : g01;
        a2 = odd ? g23 : a2; a3 = odd ? a3 : g23;
    }
    {
        const float s02 = hi ? a0 : a2, s13 = hi ? a1 : a3;
        const float g02 = dpp_xor2(s02), g13 = dpp_xor2(s13);
        a0 = hi ? g02 : a0; a2 = hi ? a2 : g02;
        a1 = hi ? g13 : a1; a3 = hi ? a3 : g13;
    }
}

__device__ __forceinline__ float act_f(float y, float r, bool add_pre, bool relu, bool add_post) {
    y = add_pre ? y + r : y;
    const float t = y > 0.0f ? y : 0.0f;
    y = relu ? t : y;
    return add_post ? y + r : y;
}

// Four consecutive floats; V4: one 16-byte load (the caller guarantees all four exist), else two
// 8-byte loads, the second only where `hi_ok` (it is redirected to the first pair otherwise).
template <bool V4>
__device__ __forceinline__ f32x4 load_quad(const char *p, bool hi_ok) {
    if constexpr (V4) {
        return *reinterpret_cast<const f32x4 *>(p);
    } else {
        const float2 lo = *reinterpret_cast<const float2 *>(p);
        const float2 hi = *reinterpret_cast<const float2 *>(p + (hi_ok ? 8 : 0));
        f32x4 o;
        o[0] = lo.x; o[1] = lo.y; o[2] = hi.x; o[3] = hi.y;
        return o;
    }
}

// Lane owns the output PAIR (v0[r], v1[r]) at columns (2*col, 2*col+1) of channel
// c(r) = (r&3) + 8*(r>>2) + 4*(lane>>5) -- the layout of the Winograd and transposed-convolution
// kernels.  Even lanes end up with columns 4k..4k+3 of channel c(r), odd lanes with the same columns
// of c(r+1).  base: channel-group base (wave-uniform), cs: channel stride in bytes, vo: this lane's
// byte offset of its pair inside channel 4*(lane>>5).
__device__ __forceinline__ void store_pairs_x4(char *base, int64_t cs, unsigned vo, bool ok, const f32x16 &v0,
                                               const f32x16 &v1, int lane) {
    const bool odd = (lane & 1) != 0;
    const unsigned off = vo + (odd ? (unsigned)cs - 8u : 0u);   // odd lane: next channel, one pair to the left
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const int cl = (r & 3) + 8 * (r >> 2);
        const float s0 = odd ? v0[r] : v0[r + 1], s1 = odd ? v1[r] : v1[r + 1];
        const float g0 = dpp_xor1(s0), g1 = dpp_xor1(s1);
        f32x4 o;
        o[0] = odd ? g0 : v0[r];
        o[1] = odd ? g1 : v1[r];
        o[2] = odd ? v0[r + 1] : g0;
        o[3] = odd ? v1[r + 1] : g1;
        if (ok) *reinterpret_cast<f32x4 *>(base + cl * cs + off) = o;
    }
}

// ------------------------------------------------------------------------------------ conv
template <int KS_, int STRIDE_, int DIL_, int MI_, int TD_, int TH_, int KC_, bool DB_, int OCC_ = 2, int KDG_ = 0, int KSD_ = KS_,
          int KSH_ = KS_>
struct ConvCfg {
    // KSH: kernel extent along H (= KS except for the 3 x 7 depth-1 layer of the sheared first convolution, desc.ksize_h)
    static constexpr int KSH = KSH_;
    static constexpr int PAD_H = DIL_ * (KSH_ - 1) / 2;
    // KSD: kernel extent along D.  KSD = KS for the cubic 3D layers; KSD = 1 (with TD = 1) is the depth-1 form that
    // runs the 2D BEV neck's Conv2d layers on [N,C,1,H,W] views (desc.ksize_d = 1).
    // A chunk (KC input channels) is consumed in NPH phases of KDG kernel depth-slices each; the
    // weights of ONE phase are resident in LDS at a time (double buffered), the input image of the
    // whole chunk stays resident across its phases.
    static constexpr int KSD = KSD_;
    static constexpr int KDG = KDG_ == 0 ? KSD_ : KDG_;
    static constexpr int NPH = KSD_ / KDG;
    static_assert(KSD_ % KDG == 0, "phases must tile the kernel depth");
    static constexpr int OCC = OCC_;                // minimum waves per SIMD requested from the register allocator
    static constexpr int KS = KS_, STRIDE = STRIDE_, DIL = DIL_, MI = MI_, TD = TD_, TH = TH_, KC = KC_;
    static constexpr bool DB = DB_;                 // double-buffered LDS image (one barrier per chunk)
    static constexpr int TW = 32;
    static constexpr int PAD = DIL * (KS - 1) / 2;
    static constexpr int PAD_D = DIL * (KSD - 1) / 2;
    static constexpr int LPAD = (PAD + 3) / 4 * 4;  // left halo rounded to a 16-byte boundary
    static constexpr int XOFF = LPAD - PAD;         // image column of the tile's first needed input
    static constexpr int IN_D = (TD - 1) * STRIDE + (KSD - 1) * DIL + 1;
    static constexpr int IN_H = (TH - 1) * STRIDE + (KSH - 1) * DIL + 1;
    static constexpr int IN_W = (TW - 1) * STRIDE + (KS - 1) * DIL + 1;
    static constexpr int IN_WV = (XOFF + IN_W + 3) / 4 * 4;
    using St = Stager<KC, IN_D, IN_H, IN_WV>;
    static constexpr int CH = St::CH, TILE = St::TILE;
    static constexpr int NB = TD * TH / 4;          // 32-voxel rows per wave (4 waves)
    static constexpr int KP = KC / 2;               // MFMA k-steps per chunk
    static constexpr int TAPS = KSD * KSH * KS;
    static constexpr int WF = KDG * KSH * KS * KP * 64 * MI;  // packed weight floats per phase
    using Ws = WeightStager<WF>;
    static constexpr int IMG_BUFS = DB ? 2 : 1;
    static constexpr int LDS_BYTES = (TILE * IMG_BUFS + WF * 2) * 4;   // images, then two weight buffers
    static_assert(TD * TH % 4 == 0, "rows must split over 4 waves");
    static_assert(KC % 2 == 0, "KC must be even (MFMA K = 2)");
};

template <class Cfg>
__device__ __forceinline__ void conv_compute_phase(const float *__restrict__ img, const float *__restrict__ wl,
                                                   int bbase, int wave, f32x16 (&acc)[Cfg::NB][Cfg::MI]) {
    // img: staged input image, already advanced to this phase's first depth slice;
    // wl : staged weights of this phase + lane*MI.  Both in LDS.
    constexpr int KS = Cfg::KS, S = Cfg::STRIDE, DIL = Cfg::DIL, MI = Cfg::MI, TH = Cfg::TH, KP = Cfg::KP;
    constexpr int NB = Cfg::NB, IN_H = Cfg::IN_H, IN_WV = Cfg::IN_WV, CH = Cfg::CH, KDG = Cfg::KDG;
    constexpr int KSH = Cfg::KSH;
    constexpr int UNR = KS <= 3 ? KSH : 1;  // k3: fully unrolled taps; k5/k7: the kh loop stays rolled
#pragma unroll
    for (int kd = 0; kd < KDG; ++kd) {
#pragma unroll UNR
        for (int kh = 0; kh < KSH; ++kh) {
            const float *wrow = wl + ((kd * KSH + kh) * KS) * KP * 64 * MI;
            const int tap_base = bbase + (kd * DIL * IN_H + kh * DIL) * IN_WV;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) {
                    float af[MI];
#pragma unroll
                    for (int m = 0; m < MI; ++m) af[m] = wrow[(kw * KP + kp) * 64 * MI + m];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int row = wave * NB + nb;
                        const int dd = row / TH, hh = row % TH;
                        const float bf = img[tap_base + kp * 2 * CH + (dd * S * IN_H + hh * S) * IN_WV + kw * DIL];
#pragma unroll
                        for (int m = 0; m < MI; ++m)
                            acc[nb][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bf, acc[nb][m], 0, 0, 0);
                    }
                }
            }
        }
    }
}

// Fast epilogue of the direct kernel (toolkit comment above).  A lane holds ONE output voxel of 16
// channels per accumulator; 4x4 register/lane transposes (DPP) turn every 4 registers x 4 lanes into
// "lane = channel, registers = 4 consecutive voxels", i.e. one 16-byte store per lane where the
// generic epilogue issues four 4-byte ones.  No depth-class planes here (host).
template <class Cfg, bool RES>
__device__ __forceinline__ void conv_epilogue_fast(const ConvArgs &a, f32x16 (&acc)[Cfg::NB][Cfg::MI], int od0, int oh0,
                                                   int ow0, int cg, int64_t n, int lane, int wave) {
    constexpr int MI = Cfg::MI, TH = Cfg::TH, NB = Cfg::NB;
    const int half = lane >> 5, li = lane & 3, lj = (lane & 31) >> 2;
    const bool odd = (lane & 1) != 0, hi = (lane & 2) != 0;
    const int out_hw = a.Hout * a.Wout, out_dhw = out_hw * a.Dout;
    const int64_t cs = (int64_t)out_dhw * 4;
    const bool relu = (a.flags & SNVC_EPI_RELU) != 0, add_pre = (a.flags & SNVC_EPI_ADD_PRE) != 0,
               add_post = (a.flags & SNVC_EPI_ADD_POST) != 0;
    const int ow = ow0 + 4 * lj;
    unsigned voff[NB];
    bool ok[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row = wave * NB + nb;
        const int od = od0 + row / TH, oh = oh0 + row % TH;
        ok[nb] = od < a.Dout && oh < a.Hout && ow < a.Wout;     // Wout % 4 == 0 (host): all four voxels exist
        const int sp = ok[nb] ? od * out_hw + oh * a.Wout + ow : 0;
        voff[nb] = 4u * (unsigned)((4 * half + li) * out_dhw + sp);
    }
#pragma unroll
    for (int m = 0; m < MI; ++m) {
        const int cbase = __builtin_amdgcn_readfirstlane((cg * MI + m) * 32);
        const char *const rb = RES ? reinterpret_cast<const char *>(a.res + n * a.r_bs) + cbase * cs : nullptr;
        float sc[4], bi[4];     // this lane's channel of each register quad: cbase + 8q + 4*half + li
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[q] = a.scale ? a.scale[cbase + 8 * q + 4 * half + li] : 1.0f;
            bi[q] = a.scale ? a.bias[cbase + 8 * q + 4 * half + li] : 0.0f;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 rv[RES ? 4 : 1];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (RES) rv[q] = *reinterpret_cast<const f32x4 *>(rb + 8 * q * cs + voff[nb]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v0 = acc[nb][m][4 * q], v1 = acc[nb][m][4 * q + 1], v2 = acc[nb][m][4 * q + 2], v3 = acc[nb][m][4 * q + 3];
                quad_transpose(v0, v1, v2, v3, odd, hi);
                acc[nb][m][4 * q] = act_f(v0 * sc[q] + bi[q], RES ? rv[q][0] : 0.0f, add_pre, relu, add_post);
                acc[nb][m][4 * q + 1] = act_f(v1 * sc[q] + bi[q], RES ? rv[q][1] : 0.0f, add_pre, relu, add_post);
                acc[nb][m][4 * q + 2] = act_f(v2 * sc[q] + bi[q], RES ? rv[q][2] : 0.0f, add_pre, relu, add_post);
                acc[nb][m][4 * q + 3] = act_f(v3 * sc[q] + bi[q], RES ? rv[q][3] : 0.0f, add_pre, relu, add_post);
            }
            // one batch of loads in flight at a time: pin this batch's results before the next loads
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(acc[nb][m][r])::"memory");
        }
    }
#pragma unroll
    for (int m = 0; m < MI; ++m) {
        const int cbase = __builtin_amdgcn_readfirstlane((cg * MI + m) * 32);
        char *const yb = reinterpret_cast<char *>(a.y + n * a.y_bs) + cbase * cs;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o;
                o[0] = acc[nb][m][4 * q]; o[1] = acc[nb][m][4 * q + 1]; o[2] = acc[nb][m][4 * q + 2]; o[3] = acc[nb][m][4 * q + 3];
                if (ok[nb]) *reinterpret_cast<f32x4 *>(yb + 8 * q * cs + voff[nb]) = o;
            }
    }
}

template <class Cfg, int EPI = 0>   // EPI: 0 generic epilogue, 1 fast, 2 fast with residual
__global__ void __launch_bounds__(256, Cfg::OCC)
conv3d_mfma_kernel(const ConvArgs a) {
    constexpr int S = Cfg::STRIDE, MI = Cfg::MI, TD = Cfg::TD, TH = Cfg::TH, KC = Cfg::KC, NB = Cfg::NB;
    constexpr int CH = Cfg::CH, TILE = Cfg::TILE, LPAD = Cfg::LPAD, XOFF = Cfg::XOFF;
    using St = typename Cfg::St;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = a.tiles_d * a.tiles_h * a.tiles_w;
    const int t = xcd_remap(blockIdx.x, ntiles);
    const int tw = t % a.tiles_w, th = (t / a.tiles_w) % a.tiles_h, td = t / (a.tiles_w * a.tiles_h);
    const int cg = blockIdx.y;  // group of 32*MI output channels
    const int64_t n = blockIdx.z;
    const int od0 = td * TD, oh0 = th * TH, ow0 = tw * 32;
    const int id0 = od0 * S - Cfg::PAD_D, ih0 = oh0 * S - Cfg::PAD_H, ix0 = ow0 * S - LPAD;

    f32x16 acc[NB][MI];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][m][r] = 0.0f;

    const int in_hw = a.Hin * a.Win, in_dhw = in_hw * a.Din;
    const float *xn = a.x + n * a.x_bs;
    // B-fragment base: lane&31 = voxel column, lane>>5 = k within the k-pair
    const int bbase = (lane >> 5) * CH + (lane & 31) * S + XOFF;

    constexpr int WF = Cfg::WF, NPH = Cfg::NPH, KDG = Cfg::KDG, DIL = Cfg::DIL;
    constexpr int SLICE = KDG * DIL * Cfg::IN_H * Cfg::IN_WV;   // image floats per phase step (depth advance)
    using Ws = typename Cfg::Ws;
    float *const wlds = lds + TILE * Cfg::IMG_BUFS;          // two weight buffers behind the image(s)
    // packed weights of this channel group: [chunk][phase] blocks of WF floats
    const float *wg = a.wp + (int64_t)cg * a.nchunks * NPH * WF;
    const int nphase = a.nchunks * NPH;
    if (a.vec) {
        St st;
        st.init(tid, id0, ih0, ix0, a.Din, a.Hin, a.Win, in_hw, in_dhw);
        typename St::Vec pre[St::NIT];
        f32x4 wpre[Ws::NIT];
        st.load(xn, tid, a.Cin, pre);
        Ws::load(wg, tid, wpre);
        st.store(lds, tid, a.Cin, pre);
        Ws::store(wlds, tid, wpre);
        __syncthreads();
        int chunk = 0, ph = 0;
        for (int p = 0; p < nphase; ++p) {
            const bool more = p + 1 < nphase;
            const bool last_of_chunk = ph == NPH - 1;
            const bool new_img = more && last_of_chunk;       // the next phase starts a new chunk
            if (more) Ws::load(wg + (int64_t)(p + 1) * WF, tid, wpre);
            if (new_img) st.load(xn + (int64_t)(chunk + 1) * KC * in_dhw, tid, a.Cin - (chunk + 1) * KC, pre);
            const float *img = lds + (Cfg::DB ? (chunk & 1) * TILE : 0) + ph * SLICE;
            conv_compute_phase<Cfg>(img, wlds + (p & 1) * WF + lane * MI, bbase, wave, acc);
            if (more) Ws::store(wlds + ((p + 1) & 1) * WF, tid, wpre);
            if (Cfg::DB) {
                if (new_img) st.store(lds + ((chunk + 1) & 1) * TILE, tid, a.Cin - (chunk + 1) * KC, pre);
                __syncthreads();
            } else {
                __syncthreads();
                if (new_img) {            // uniform: the single image may only be overwritten once all
                    st.store(lds, tid, a.Cin - (chunk + 1) * KC, pre);   // waves have finished reading it
                    __syncthreads();
                }
            }
            if (last_of_chunk) { ph = 0; ++chunk; } else { ++ph; }
        }
    } else {
        for (int p = 0; p < nphase; ++p) {
            const int chunk = p / NPH, ph = p - chunk * NPH;
            __syncthreads();
            if (ph == 0)
                St::stage_scalar(xn + (int64_t)chunk * KC * in_dhw, lds, tid, id0, ih0, ix0, a.Din, a.Hin, a.Win,
                                 in_hw, in_dhw, a.Cin - chunk * KC);
            f32x4 wpre[Ws::NIT];
            Ws::load(wg + (int64_t)p * WF, tid, wpre);
            Ws::store(wlds, tid, wpre);
            __syncthreads();
            conv_compute_phase<Cfg>(lds + ph * SLICE, wlds + lane * MI, bbase, wave, acc);
        }
    }

    // ---- epilogue
    if constexpr (EPI != 0) {
        conv_epilogue_fast<Cfg, EPI == 2>(a, acc, od0, oh0, ow0, cg, n, lane, wave);
        return;
    }
    const int ow = ow0 + (lane & 31);
    const int64_t out_hw = (int64_t)a.Hout * a.Wout, out_dhw = out_hw * a.Dout;
    float *yn = a.y + n * a.y_bs;
    const float *rn = a.res ? a.res + n * a.r_bs : nullptr;
#pragma unroll
    for (int m = 0; m < MI; ++m) {
        const int cbase = cg * 32 * MI + m * 32;
        ChanAffine f;
        load_affine(a, cbase, lane, f);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int row = wave * NB + nb;
            const int od = od0 + row / TH, oh = oh0 + row % TH;
            const bool vox_ok = od < a.Dout && oh < a.Hout && ow < a.Wout;
            const int64_t sp = vox_ok ? od * out_hw + (int64_t)oh * a.Wout + ow : 0;
            float rv[16], pv[16];
            // depth class of this output plane for the factored first convolution (see
            // snvc_conv3d_forward_ex): 0 = first plane, 2 = last plane, 1 = interior
            const int cls = od == 0 ? 0 : (od >= a.Dout - 1 ? 2 : 1);
            const int64_t psp = vox_ok ? (int64_t)cls * out_hw + (int64_t)oh * a.Wout + ow : 0;
            const float *pn = a.plane ? a.plane + n * (int64_t)a.Cout * 3 * out_hw : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                co = co < a.Cout ? co : a.Cout - 1;
                rv[r] = rn ? rn[co * out_dhw + sp] : 0.0f;
                pv[r] = pn ? pn[co * 3 * out_hw + psp] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = epilogue_f((acc[nb][m][r] + pv[r]) * f.sc[r] + f.bi[r], rv[r], a.flags);
                if (vox_ok && co < a.Cout) yn[co * out_dhw + sp] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ conv k3: Winograd F(4,3) along W
// For the 3x3x3 / stride-1 convolutions (all of cfg2's 3D FLOPs outside the hourglass resampling
// layers) the W dimension is computed with the 1-D Winograd minimal-filtering algorithm F(4,3):
// four adjacent outputs from six inputs with 6 multiplications instead of 12,
//     V = B^T d   (d = x[4t-1 .. 4t+4]):  V0 = 4d0-5d2+d4          V5 = 4d1-5d3+d5
//                  V1 = (d4-4d2)+(d3-4d1)   V2 = (d4-4d2)-(d3-4d1)
//                  V3 = (d4-d2)+2(d3-d1)    V4 = (d4-d2)-2(d3-d1)
//     U = G g     : U0 = g0/4, U1 = -(g0+g1+g2)/6, U2 = -(g0-g1+g2)/6,
//                   U3 = g0/24+g1/12+g2/6, U4 = g0/24-g1/12+g2/6, U5 = g2     (at weight-pack time, in fp64)
//     m_p = sum over (cin, kd, kh) of U_p * V_p                                 (the MFMA contraction)
//     y0 = m0+m1+m2+m3+m4        y1 = (m1-m2)+2(m3-m4)
//     y2 = (m1+m2)+4(m3+m4)      y3 = (m1-m2)+8(m3-m4)+m5
// so a (cin-pair, kd, kh) step issues 6 MFMAs for 128 output voxels where the direct form issues 12:
// half of the matrix-core work, still fp32 products and fp32 accumulation (measured error of the fp32
// F(4,3) against an fp64 convolution: 2.2e-6 of the output range on a 576-term contraction, the
// direct fp32 chain 0.7e-6 -- three orders below the 1e-3 contract; cuDNN, the reference's backend,
// picks from the same family of algorithms under its autotuner, tools/inference_agnostic.py:18).
// The input transform costs nothing extra in memory: the LDS image is the raw tile the direct kernel
// stages and V is formed in registers right after the fragment reads (one 16-byte and two 4-byte LDS
// reads + 12 VALU ops per 6 MFMAs).  The 32 MFMA columns are 2 rows x 16 tiles of 4 outputs: a lane
// owns 4 consecutive outputs, so the epilogue stores 16 bytes per lane and channel.
template <int TD_, int TH_, int KC_, int PIECE_ = 4, int TW_ = 64, int OCC_ = 2, int KSD_ = 3>
struct WinoCfg {
    static constexpr int OCC = OCC_;   // workgroups per CU the register allocator is asked to allow
    static constexpr int TD = TD_, TH = TH_, KC = KC_, MI = 1, PIECE = PIECE_;
    static constexpr int KS = 3, NPOS = 6, STRIDE = 1;
    static constexpr int KSD = KSD_;   // kernel extent along D: 3, or 1 for the depth-1 (2D neck) layers: 3 (kd, kh) taps instead of 9
    // The 32 MFMA columns are RPB rows x (TW/4) quads: 2 rows x 64 outputs, or 4 rows x 32 outputs for
    // layers whose width fills 64-wide tiles badly (W = 156 -> 3 x 64 is 81 % full, 5 x 32 is 97 %).
    static constexpr int TW = TW_, QUADS = TW / 4, RPB = 32 / QUADS, LPAD = 4, XOFF = 3;
    static_assert(TW == 64 || TW == 32, "tile width");
    static constexpr int IN_D = TD + KSD - 1, IN_H = TH + 2, IN_W = TW + 2;
    static constexpr int IN_WV = (XOFF + IN_W + 3) / 4 * 4;   // 72
    using St = Stager<KC, IN_D, IN_H, IN_WV, PIECE>;
    static constexpr int CH = St::CH, TILE = St::TILE;
    static constexpr int NB = TD * TH / (4 * RPB);            // row BLOCKS (RPB rows) per wave (4 waves)
    static constexpr int KP = KC / 2;
    static constexpr int NTAP = KSD * 3;                      // (kd, kh) taps
    static constexpr int WF = NTAP * NPOS * KP * 64;          // packed floats per chunk: [tap9][pos][kp][lane]
    using Ws = WeightStager<WF>;
    static constexpr int LDS_BYTES = (TILE * 2 + WF * 2) * 4;
    static_assert(TD * TH % (4 * RPB) == 0 && TH % RPB == 0, "row blocks must split over 4 waves");
    static_assert(CH % 4 == 0 && IN_WV % 4 == 0, "16-byte LDS reads need aligned strides");
};

// One (kd, kh, k-pair) step = 6 weight fragments (positions 0..5) + NB raw input sextuples.
template <int NB>
struct WinoStep {
    float a[6];
    float d0[NB], d5[NB];
    f32x4 d14[NB];
};

template <class Cfg>
__device__ __forceinline__ void wino_load_step(const float *__restrict__ img, const float *__restrict__ wl, int bbase,
                                               int wave, int step, WinoStep<Cfg::NB> &o) {
    constexpr int TH = Cfg::TH, KP = Cfg::KP, NB = Cfg::NB, IN_H = Cfg::IN_H, IN_WV = Cfg::IN_WV, CH = Cfg::CH;
    const int kp = step % KP, tap9 = step / KP;         // compile-time after unrolling
    const int kd = tap9 / 3, kh = tap9 % 3;
#pragma unroll
    for (int p = 0; p < 6; ++p) o.a[p] = wl[((tap9 * 6 + p) * KP + kp) * 64];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int rp = wave * NB + nb;                  // row block: rows (dd, hh0 .. hh0 + RPB - 1)
        const int dd = rp / (TH / Cfg::RPB), hh0 = Cfg::RPB * (rp % (TH / Cfg::RPB));
        const float *px = img + bbase + kp * 2 * CH + ((dd + kd) * IN_H + hh0 + kh) * IN_WV;
        o.d0[nb] = px[0];
        o.d14[nb] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(px + 1, 16));
        o.d5[nb] = px[5];
    }
}

// `mid(s)` runs between steps s and s+1: the kernel uses it to retire the NEXT chunk's prefetch
// (registers -> the idle LDS buffers) a piece at a time in the shadow of the MFMAs, so that a chunk
// ends with a bare barrier instead of a vmcnt wait + a burst of LDS writes.
template <class Cfg, class Mid>
__device__ __forceinline__ void wino_compute_chunk(const float *__restrict__ img, const float *__restrict__ wl,
                                                   int bbase, int wave, f32x16 (&acc)[6][Cfg::NB], Mid &&mid) {
    constexpr int NB = Cfg::NB, NS = Cfg::NTAP * Cfg::KP;
    constexpr bool PIPE = false;   // explicit one-step-ahead fragment reads measured slower (2.93 -> 3.05 ms on cfg2 conv1)
    WinoStep<NB> cur, nxt;
    if (PIPE) wino_load_step<Cfg>(img, wl, bbase, wave, 0, cur);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (PIPE) {
            if (s + 1 < NS) wino_load_step<Cfg>(img, wl, bbase, wave, s + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            wino_load_step<Cfg>(img, wl, bbase, wave, s, cur);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float d0 = cur.d0[nb], d1 = cur.d14[nb][0], d2 = cur.d14[nb][1], d3 = cur.d14[nb][2],
                        d4 = cur.d14[nb][3], d5 = cur.d5[nb];
            const float ta = __builtin_fmaf(-4.0f, d2, d4), tb = __builtin_fmaf(-4.0f, d1, d3), tc = d4 - d2, te = d3 - d1;
            const float v0 = __builtin_fmaf(4.0f, d0, __builtin_fmaf(-5.0f, d2, d4));
            const float v5 = __builtin_fmaf(4.0f, d1, __builtin_fmaf(-5.0f, d3, d5));
            const float v1 = ta + tb, v2 = ta - tb, v3 = __builtin_fmaf(2.0f, te, tc), v4 = __builtin_fmaf(-2.0f, te, tc);
            acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[0], v0, acc[0][nb], 0, 0, 0);
            acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[1], v1, acc[1][nb], 0, 0, 0);
            acc[2][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[2], v2, acc[2][nb], 0, 0, 0);
            acc[3][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[3], v3, acc[3][nb], 0, 0, 0);
            acc[4][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[4], v4, acc[4][nb], 0, 0, 0);
            acc[5][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[5], v5, acc[5][nb], 0, 0, 0);
        }
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
        mid(s);
        if (PIPE) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < NS) cur = nxt;
        }
    }
}

// A job = one output tile of one 32-channel group of one sample; one workgroup per job.
struct WinoJob {
    int od0, oh0, ow0, cg;
    int id;              // (n * tiles + tile) * groups + cg: the job's slot in per-job outputs (ConvArgs::stats)
    int64_t n;
};

__device__ __forceinline__ WinoJob wino_decode_job(const ConvArgs &a, int v, int total, int TD, int TH, int TW) {
    const int ntiles = a.tiles_d * a.tiles_h * a.tiles_w;
    const int j = xcd_remap(v, total);
    // channel group fastest: the groups of one tile run back to back on one XCD, so the tile's input is read from HBM
    // once and from that XCD's L2 by the other groups (r3: hg conv1, 2 groups, read 2.1x its algorithmic bytes when the
    // groups were whole passes over the volume apart -- profiles/r3/traffic.json)
    const int cg = j % a.groups, rest = j / a.groups;
    const int t = rest % ntiles;
    WinoJob o;
    o.ow0 = (t % a.tiles_w) * TW;
    o.oh0 = ((t / a.tiles_w) % a.tiles_h) * TH;
    o.od0 = (t / (a.tiles_w * a.tiles_h)) * TD;
    o.cg = cg;
    o.id = j;
    o.n = rest / ntiles;
    return o;
}

// ------------------------------------------------------------------------------------ conv k5 / k7: Winograd F(4,5) / F(4,7) along W
// The local model's first layers (7x7x7 64->32 and 5x5x5 32->32 on the whole voxel grid) carry two
// thirds of its FLOPs.  Along W they use F(4,KS): 4 outputs from KS+3 inputs with KS+3 multiplications
// instead of 4*KS (k5: 8 vs 20, k7: 10 vs 28).  The Cook-Toom matrices come from
// tools/gen_wino_tables.py (exact rationals, symmetric point sets 0, +-1, +-2, +-1/2 (, +-3), inf);
// the fp32 algorithm differs from an fp64 convolution by 3e-6 (k5) / 3e-5 (k7) of the output range
// on a 448-term contraction.  B^T rows of a +-p pair are (even part) +- (odd part), so the input
// transform is 27 (k5) / 40 (k7) VALU operations per step of 8 / 10 MFMAs.
template <int KS> struct WinoTables;
template <> struct WinoTables<5> {
    static constexpr int P = 8;
    static constexpr double bt(int r, int c) { return wino::BT5[r][c]; }
    static constexpr double at(int r, int c) { return wino::AT5[r][c]; }
    static constexpr double g(int r, int c) { return wino::G5[r][c]; }
};
template <> struct WinoTables<7> {
    static constexpr int P = 10;
    static constexpr double bt(int r, int c) { return wino::BT7[r][c]; }
    static constexpr double at(int r, int c) { return wino::AT7[r][c]; }
    static constexpr double g(int r, int c) { return wino::G7[r][c]; }
};

// v = B^T d.  Row 0 and row P-1 are plain dot products; rows (2j+1, 2j+2) share an even-column part E
// and an odd-column part O:  v[2j+1] = E + O,  v[2j+2] = E - O.
template <int KS>
__device__ __forceinline__ void wino_input_transform(const float (&d)[WinoTables<KS>::P], float (&v)[WinoTables<KS>::P]) {
    using T = WinoTables<KS>;
    constexpr int P = T::P;
    {
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int c = 0; c < P; ++c) {
            if (T::bt(0, c) != 0.0) s0 = __builtin_fmaf((float)T::bt(0, c), d[c], s0);
            if (T::bt(P - 1, c) != 0.0) s1 = __builtin_fmaf((float)T::bt(P - 1, c), d[c], s1);
        }
        v[0] = s0;
        v[P - 1] = s1;
    }
#pragma unroll
    for (int j = 0; 2 * j + 2 < P - 1; ++j) {
        float e = 0.0f, o = 0.0f;
#pragma unroll
        for (int c = 0; c < P; ++c) {
            if (T::bt(2 * j + 1, c) == 0.0) continue;
            if (c % 2 == 0) e = __builtin_fmaf((float)T::bt(2 * j + 1, c), d[c], e);
            else o = __builtin_fmaf((float)T::bt(2 * j + 1, c), d[c], o);
        }
        v[2 * j + 1] = e + o;
        v[2 * j + 2] = e - o;
    }
}

// y = A^T m (4 outputs), with the same pair structure: s = m+ + m-, t = m+ - m-.
template <int KS>
__device__ __forceinline__ void wino_output_transform(const float (&m)[WinoTables<KS>::P], float (&y)[4]) {
    using T = WinoTables<KS>;
    constexpr int P = T::P;
    y[0] = m[0]; y[1] = 0.0f; y[2] = 0.0f; y[3] = m[P - 1];
#pragma unroll
    for (int j = 0; 2 * j + 2 < P - 1; ++j) {
        const float sj = m[2 * j + 1] + m[2 * j + 2], tj = m[2 * j + 1] - m[2 * j + 2];
        y[0] = __builtin_fmaf((float)T::at(0, 2 * j + 1), sj, y[0]);
        y[1] = __builtin_fmaf((float)T::at(1, 2 * j + 1), tj, y[1]);
        y[2] = __builtin_fmaf((float)T::at(2, 2 * j + 1), sj, y[2]);
        y[3] = __builtin_fmaf((float)T::at(3, 2 * j + 1), tj, y[3]);
    }
}

// Tile 4 x 4 x 32 voxels x 32 channels (a wave = one depth slice: 4 rows x 8 quads), KC = 2 channels
// per chunk, one kernel depth-slice (kd) of weights resident at a time (double buffered), the input
// image single-buffered; everything staged by LDS-DMA (no prefetch registers: the 8 or 10 position
// accumulators take 128 / 160 of them).
// Dilation 2 (the local model's third convolution, k5): in D and H the taps are simply two rows apart;
// along W the convolution splits into two independent dense ones over the even and the odd columns
// (polyphase), so a "quad" is 4 outputs of ONE parity, 2 columns apart, its 8 inputs are every other
// element of four 16-byte reads, and the epilogue first re-interleaves the two parities between
// neighbouring lanes so that every lane again owns 4 consecutive outputs.
template <int KS_, int KC_, int DIL_ = 1>
struct WinoKCfg {
    static constexpr int KS = KS_, KC = KC_, DIL = DIL_, STRIDE = 1, MI = 1, TD = 4, TH = 4, PIECE = 4;
    static constexpr int NPOS = WinoTables<KS>::P;
    static constexpr int TW = 32, QUADS = 8, RPB = 4, NB = 1;
    static constexpr int PAD = DIL * (KS - 1) / 2, LPAD = 4, XOFF = LPAD - PAD;
    static constexpr int IN_D = TD + DIL * (KS - 1), IN_H = TH + DIL * (KS - 1), IN_W = TW + DIL * (KS - 1);
    static constexpr int IN_WV = (XOFF + IN_W + 3) / 4 * 4;
    static_assert(DIL == 1 || (DIL == 2 && XOFF == 0 && NPOS == 8), "dilation 2 is built for k5");
    using St = Stager<KC, IN_D, IN_H, IN_WV, 4>;
    static constexpr int CH = St::CH, TILE = St::TILE;
    static constexpr int KP = KC / 2;
    static constexpr int WF = KS * NPOS * KP * 64;            // packed floats per (chunk, kd): [kh][pos][kp][lane]
    static constexpr int WPH = 1;                              // kernel depth slices of weights resident at a time
    static constexpr int LDS_BYTES = (TILE + 2 * WF) * 4;
    static_assert(DIL == 2 || (XOFF + 4 * (QUADS - 1) + NPOS <= IN_WV && XOFF + NPOS <= 12),
                  "three 16-byte reads cover a quad's inputs");
    static_assert(DIL == 1 || 8 * 3 + 16 <= IN_WV, "four 16-byte reads cover a dilated quad's inputs");
};

// Stride-2 3x3x3 convolution (the hourglass down-sampling layers): in D and H the stride only changes
// which input rows a tap reads; along W the layer splits into its two polyphase components,
//     y[o] = g1 * xe[o]  +  ( g0 * xo[o-1] + g2 * xo[o] ),     xe[i] = x[2i], xo[i] = x[2i+1]:
// a pointwise product on the even columns (4 MFMAs per 4 outputs, one shared weight fragment) and a
// 2-tap stride-1 convolution on the odd ones, done with F(4,2) (5 MFMAs per 4 outputs instead of 8):
// 9 MFMAs per quad and (cin pair, kd, kh) where the direct kernel issues 12.  The 12 floats
// x[2o-4 .. 2o+7] of a quad come from three 16-byte LDS reads; xe and xo are their even / odd elements.
template <int KC_, int STORE_PIECE_ = 4>
struct WinoS2Cfg {
    static constexpr int KS = 3, KC = KC_, DIL = 1, STRIDE = 2, MI = 1, TD = 4, TH = 4, PIECE = 4;
    static constexpr int STORE_PIECE = STORE_PIECE_;
    static constexpr int NPOS = 9;                            // 4 even-phase + 5 F(4,2) positions
    static constexpr int NA = 6;                              // weight fragments per step: g1, U0..U4
    static constexpr int TW = 32, QUADS = 8, RPB = 4, NB = 1;
    static constexpr int PAD = 1, LPAD = 4, XOFF = 3;
    static constexpr int IN_D = 2 * TD + 1, IN_H = 2 * TH + 1, IN_W = 2 * (TW - 1) + 3;
    static constexpr int IN_WV = (XOFF + IN_W + 3) / 4 * 4;   // 68
    using St = Stager<KC, IN_D, IN_H, IN_WV, 4>;
    static constexpr int CH = St::CH, TILE = St::TILE;
    static constexpr int KP = KC / 2;
    static constexpr int WF = KS * NA * KP * 64;              // packed floats per (chunk, kd): [kh][frag][kp][lane]
    // the weights of a WHOLE chunk (3 depth slices) are resident: one barrier per 81 MFMAs instead of one per 27
    static constexpr int WPH = KS;
    static constexpr int LDS_BYTES = (TILE + 2 * WPH * WF) * 4;
    static_assert(8 * (QUADS - 1) + 12 <= IN_WV, "three 16-byte reads cover a quad's inputs");
};

template <class Cfg>
__device__ __forceinline__ void winos2_compute_phase(const float *__restrict__ img, const float *__restrict__ wl, int bbase,
                                                     f32x16 (&acc)[9][1]) {
    constexpr int KP = Cfg::KP, IN_WV = Cfg::IN_WV, CH = Cfg::CH, NA = Cfg::NA;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            float af[NA];
#pragma unroll
            for (int q = 0; q < NA; ++q) af[q] = wl[((kh * NA + q) * KP + kp) * 64];
            const float *px = img + bbase + kp * 2 * CH + kh * IN_WV;
            f32x4 t[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) t[j] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(px + 4 * j, 16));
            // element e of the 12 = x[2o - 4 + e]:  xe[o+j] = e 4+2j,  xo[o-1+i] = e 3+2i
            const float e0 = t[1][0], e1 = t[1][2], e2 = t[2][0], e3 = t[2][2];
            const float d0 = t[0][3], d1 = t[1][1], d2 = t[1][3], d3 = t[2][1], d4 = t[2][3];
            // F(4,2) input transform (wino::BT2)
            const float v0 = __builtin_fmaf(2.0f, d0 - d2, d3 - d1);
            const float v1 = (d3 - d2) - 2.0f * d1;
            const float v2 = __builtin_fmaf(2.0f, d1, __builtin_fmaf(-3.0f, d2, d3));
            const float v3 = d3 - d1;
            const float v4 = __builtin_fmaf(2.0f, d1 - d3, d4 - d2);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], e0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], e1, acc[1][0], 0, 0, 0);
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], e2, acc[2][0], 0, 0, 0);
            acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], e3, acc[3][0], 0, 0, 0);
            acc[4][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], v0, acc[4][0], 0, 0, 0);
            acc[5][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], v1, acc[5][0], 0, 0, 0);
            acc[6][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], v2, acc[6][0], 0, 0, 0);
            acc[7][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[4], v3, acc[7][0], 0, 0, 0);
            acc[8][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[5], v4, acc[8][0], 0, 0, 0);
        }
    }
}

template <class Cfg>
__device__ __forceinline__ void winok_compute_phase(const float *__restrict__ img, const float *__restrict__ wl, int bbase,
                                                    int lane, f32x16 (&acc)[Cfg::NPOS][1]) {
    constexpr int KS = Cfg::KS, P = Cfg::NPOS, KP = Cfg::KP, IN_WV = Cfg::IN_WV, CH = Cfg::CH, XOFF = Cfg::XOFF;
    constexpr int DIL = Cfg::DIL;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            float af[P];
#pragma unroll
            for (int q = 0; q < P; ++q) af[q] = wl[((kh * P + q) * KP + kp) * 64];
            const float *px = img + bbase + kp * 2 * CH + kh * DIL * IN_WV;
            float d[P], v[P];
            if constexpr (DIL == 1) {
                f32x4 t[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) t[j] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(px + 4 * j, 16));
#pragma unroll
                for (int i = 0; i < P; ++i) d[i] = t[(XOFF + i) / 4][(XOFF + i) % 4];
            } else {   // inputs 2i + parity of the 16 floats at px (px already points at this quad pair)
                f32x4 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(px + 4 * j, 16));
                const bool odd = (lane & 1) != 0;
#pragma unroll
                for (int i = 0; i < P; ++i) d[i] = odd ? t[(2 * i + 1) / 4][(2 * i + 1) % 4] : t[(2 * i) / 4][(2 * i) % 4];
            }
            wino_input_transform<KS>(d, v);
#pragma unroll
            for (int q = 0; q < P; ++q) acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], v[q], acc[q][0], 0, 0, 0);
        }
    }
}

// Width of the epilogue's loads / stores in floats: a configuration may stage 16-byte pieces (input rows) and
// still store 8-byte halves (output rows with W % 4 == 2), e.g. the stride-2 layer 156 -> 78.
template <class Cfg, class = void> struct EpiloguePiece { static constexpr int value = Cfg::PIECE; };
template <class Cfg> struct EpiloguePiece<Cfg, decltype((void)Cfg::STORE_PIECE)> { static constexpr int value = Cfg::STORE_PIECE; };
template <class Cfg> constexpr int epilogue_piece() { return EpiloguePiece<Cfg>::value; }

// Epilogue of one job (fast-epilogue toolkit above): output transform in place (y0..y3 -> acc[0..3],
// a third of the accumulator registers become free), per-channel scale and bias through LDS (parked
// at kernel start; lgkmcnt), every vector load (residual or depth-class planes) before the first
// store, addresses = wave-uniform channel base (SGPR pair) + one 32-bit per-lane byte offset per row
// block (the host routes layers with more than 2^27 output voxels per channel elsewhere).
// HEAD (side head, snvc_conv3d_forward_side_head): besides y, the launch writes the 1x1x1 projection of its own
// 32-channel result to one channel, y_head = sum_c head_w[c] * y[c] -- 16 channels per lane, the other 16 in
// lane ^ 32 -- so that a later consumer of head(y) does not have to read y again (the global model's
// `classifier(v)` term of the folded hourglass tail, models/stereo_volume.py).
// XMODE 2 (SNVC_EPI_AVGPOOL_D4): AvgPool3d((4,1,1),(4,1,1)) of the layer's own result (snvc/models/vernier.py:289,436: the
// pool between conv4 and the BEV reshape) folded in: a 4 x 4 x 32 tile holds exactly the four depths of one pooled plane,
// one per wave, so the waves trade registers through the (now idle) LDS buffers -- wave w collects channels 8w .. 8w+3
// (+4 for the upper half-wave) of all four depths, sums them in depth order, scales by 1/4 and stores the pooled
// [N, C, D/4, H, W] tensor; the full-resolution result is never written.
template <int W, class Cfg>
__device__ __forceinline__ void wino_pool_store(const ConvArgs &a, const WinoJob &job, f32x16 (&acc)[Cfg::NPOS][Cfg::NB],
                                                const f32x4 *__restrict__ X, int lane) {
    const int half = lane >> 5, rowsel = (lane & 31) / Cfg::QUADS;
    const int ow = job.ow0 + 4 * (lane & (Cfg::QUADS - 1)), oh = job.oh0 + rowsel;
    const int out_hw = a.Hout * a.Wout;
    const int64_t pooled_dhw = (int64_t)(a.Dout / 4) * out_hw;
    const bool ok = oh < a.Hout && ow < a.Wout;
    float *yb = a.y + job.n * a.y_bs + (int64_t)(job.cg * 32 + 8 * W + 4 * half) * pooled_dhw +
                (ok ? (int64_t)(job.od0 / 4) * out_hw + oh * a.Wout + ow : 0);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        f32x4 v[4];
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) {
            if (sw == W) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[sw][j] = acc[j][0][4 * W + rr];
            } else {
                v[sw] = X[((W * 3 + (sw < W ? sw : sw - 1)) * 4 + rr) * 64 + lane];
            }
        }
        const f32x4 sum = ((v[0] + v[1]) + v[2]) + v[3];
        if (ok) *reinterpret_cast<f32x4 *>(yb + rr * pooled_dhw) = sum * 0.25f;
    }
}

template <class Cfg, bool RES, bool PLANE, int XMODE = 0>
__device__ __forceinline__ void wino_epilogue(const ConvArgs &a, const WinoJob &job, f32x16 (&acc)[Cfg::NPOS][Cfg::NB],
                                              const float *__restrict__ aff, int lane, int wave, float *xch = nullptr) {
    constexpr bool HEAD = XMODE == 1;
    constexpr int NB = Cfg::NB, TH = Cfg::TH;
    constexpr bool V4 = epilogue_piece<Cfg>() == 4;   // 16-byte loads / stores (Wout % 4 == 0); else 8-byte halves
    const int ow = job.ow0 + 4 * (lane & (Cfg::QUADS - 1)), rowsel = (lane & 31) / Cfg::QUADS;
    const int out_hw = a.Hout * a.Wout, out_dhw = out_hw * a.Dout;
    const int cbase = __builtin_amdgcn_readfirstlane(job.cg * 32);
    const int half = lane >> 5;                     // accumulator register r of this lane is channel
                                                    //   cbase + (r&3) + 8*(r>>2) + 4*half
    // Cout % 32 == 0 (host): every channel of the group exists, no channel predicates anywhere
    const bool relu = (a.flags & SNVC_EPI_RELU) != 0, add_pre = (a.flags & SNVC_EPI_ADD_PRE) != 0,
               add_post = (a.flags & SNVC_EPI_ADD_POST) != 0;
    const int64_t cs = (int64_t)out_dhw * 4, ps = (int64_t)out_hw * 12;   // channel strides in bytes
    char *const yb = reinterpret_cast<char *>(a.y + job.n * a.y_bs) + cbase * cs;
    const char *const rb = RES ? reinterpret_cast<const char *>(a.res + job.n * a.r_bs) + cbase * cs : nullptr;
    const char *const pb =
        PLANE ? reinterpret_cast<const char *>(a.plane + job.n * (int64_t)a.Cout * 3 * out_hw) + cbase * ps : nullptr;

#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (Cfg::STRIDE == 2) {   // even phase (positions 0..3) + A^T of F(4,2) (positions 4..8, wino::AT2)
                const float m0 = acc[4][nb][r], m1 = acc[5][nb][r], m2 = acc[6][nb][r], m3 = acc[7][nb][r], m4 = acc[8][nb][r];
                const float s12 = m1 + m2, d12 = m1 - m2;
                acc[0][nb][r] += (m0 + s12) + m3;
                acc[1][nb][r] += __builtin_fmaf(2.0f, m3, d12);
                acc[2][nb][r] += __builtin_fmaf(4.0f, m3, s12);
                acc[3][nb][r] += __builtin_fmaf(8.0f, m3, d12) + m4;
            } else if constexpr (Cfg::KS == 3) {
                const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r],
                            m4 = acc[4][nb][r], m5 = acc[5][nb][r];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                acc[0][nb][r] = (m0 + s12) + s34;
                acc[1][nb][r] = d12 + 2.0f * d34;
                acc[2][nb][r] = s12 + 4.0f * s34;
                acc[3][nb][r] = (d12 + 8.0f * d34) + m5;
            } else {
                float m[Cfg::NPOS], y[4];
#pragma unroll
                for (int q = 0; q < Cfg::NPOS; ++q) m[q] = acc[q][nb][r];
                wino_output_transform<Cfg::KS>(m, y);
                if constexpr (Cfg::DIL == 2) {
                    // even lane: outputs o, o+2, o+4, o+6; its odd neighbour: o+1, o+3, o+5, o+7.  Afterwards the
                    // even lane owns o .. o+3 and the odd lane o+4 .. o+7 (its usual 4 consecutive columns).
                    const bool odd = (lane & 1) != 0;
                    const float s0 = odd ? y[0] : y[2], s1 = odd ? y[1] : y[3];
                    const float g0 = dpp_xor1(s0), g1 = dpp_xor1(s1);
                    const float a0 = odd ? g0 : y[0], a1 = odd ? y[2] : g0, a2 = odd ? g1 : y[1], a3 = odd ? y[3] : g1;
                    y[0] = a0; y[1] = a1; y[2] = a2; y[3] = a3;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][nb][r] = y[j];
            }
        }
    __builtin_amdgcn_sched_barrier(0);

    unsigned voff[NB];
    bool ok0[NB], ok1[NB];     // outputs (ow, ow+1) and (ow+2, ow+3) in range (Wout is even)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int rp = wave * NB + nb;
        const int od = job.od0 + rp / (TH / Cfg::RPB), oh = job.oh0 + Cfg::RPB * (rp % (TH / Cfg::RPB)) + rowsel;
        const bool row_ok = od < a.Dout && oh < a.Hout;
        ok0[nb] = row_ok && ow < a.Wout;
        ok1[nb] = row_ok && ow + 2 < a.Wout;
        const int sp = ok0[nb] ? od * out_hw + oh * a.Wout + ow : 0;
        voff[nb] = 4u * (unsigned)(4 * half * out_dhw + sp);
        const int cls = od == 0 ? 0 : (od >= a.Dout - 1 ? 2 : 1);
        const unsigned poff = 4u * (unsigned)(4 * half * 3 * out_hw + (ok0[nb] ? cls * out_hw + oh * a.Wout + ow : 0));
        // addends are fetched EB channels at a time (every batch is one exposed memory round trip: as large
        // as the registers allow -- after the output transform only 4 of the position accumulators are live)
        constexpr int EB = (RES && PLANE) ? 4 : (Cfg::NPOS * NB >= 12 ? 8 : 16);
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += EB) {
            f32x4 rv[RES ? EB : 1], pv[PLANE ? EB : 1];
#pragma unroll
            for (int q = 0; q < EB; ++q) {
                const int cl = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);     // channel within the group
                if (RES) rv[q] = load_quad<V4>(rb + cl * cs + voff[nb], ok1[nb]);
                if (PLANE) pv[q] = load_quad<V4>(pb + cl * ps + poff, ok1[nb]);
            }
#pragma unroll
            for (int q = 0; q < EB; ++q) {
                const int r = r0 + q;
                const float sc = aff[(r & 3) + 8 * (r >> 2) + 4 * half], bi = aff[32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float y = acc[j][nb][r];
                    if (PLANE) y += pv[q][j];
                    acc[j][nb][r] = act_f(y * sc + bi, RES ? rv[q][j] : 0.0f, add_pre, relu, add_post);
                }
            }
            // one batch of loads in flight at a time: pin this batch's results before the next loads
#pragma unroll
            for (int q = 0; q < EB; ++q)
                asm volatile("" : "+v"(acc[0][nb][r0 + q]), "+v"(acc[1][nb][r0 + q]), "+v"(acc[2][nb][r0 + q]),
                             "+v"(acc[3][nb][r0 + q])::"memory");
        }
    }
    if constexpr (XMODE == 3) {
        // Batch statistics of the layer's result, taken while it is still in registers (train-mode BatchNorm reads the
        // tensor once less): every lane's fp32 (sum, sum of squares) of its <= 4*NB in-range values per channel go through
        // the idle LDS buffers, 64 threads add them in fp64 in a fixed order -> stats[job][channel][2].
        float *const P = xch;                       // [16 registers x 2][256 threads]
        const int tid = wave * 64 + lane;
        __syncthreads();                            // every wave is done with the image / weight buffers
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float su = 0.0f, sq = 0.0f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float v0 = ok0[nb] ? acc[0][nb][r] : 0.0f, v1 = ok0[nb] ? acc[1][nb][r] : 0.0f;
                const float v2 = ok1[nb] ? acc[2][nb][r] : 0.0f, v3 = ok1[nb] ? acc[3][nb][r] : 0.0f;
                su += (v0 + v1) + (v2 + v3);
                sq += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
            }
            P[(2 * r) * 256 + tid] = su;
            P[(2 * r + 1) * 256 + tid] = sq;
        }
        __syncthreads();
        if (tid < 64) {
            const int ch = tid >> 1, q = tid & 1;                       // channel (r & 3) + 8 * (r >> 2) + 4 * half
            const int r = (ch & 3) + 4 * (ch >> 3), hf = (ch >> 2) & 1;
            const float *src = P + (2 * r + q) * 256 + hf * 32;
            double t = 0.0;
            for (int w = 0; w < 4; ++w)
                for (int l = 0; l < 32; ++l) t += (double)src[w * 64 + ((l + tid) & 31)];   // skewed: no bank conflicts
            a.stats[(int64_t)job.id * 64 + tid] = t;
        }
    }
    if constexpr (HEAD) {
        static_assert(V4, "the side head stores 16-byte pieces");
        float hw[16];     // parked in LDS behind scale | bias at kernel start (a global load here is an exposed round trip)
#pragma unroll
        for (int r = 0; r < 16; ++r) hw[r] = aff[64 + (r & 3) + 8 * (r >> 2) + 4 * half];
        char *const yh = reinterpret_cast<char *>(a.y_head + job.n * (int64_t)out_dhw);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 dsum = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) dsum[j] = __builtin_fmaf(hw[r], acc[j][nb][r], dsum[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) dsum[j] += __shfl_xor(dsum[j], 32);
            if (half == 0 && ok0[nb]) *reinterpret_cast<f32x4 *>(yh + voff[nb]) = dsum;   // half 0: voff = 4 * voxel
        }
    }
    if constexpr (XMODE == 2) {
        static_assert(V4 && NB == 1 && Cfg::TD == 4 && Cfg::TH == Cfg::RPB, "one depth per wave, 16-byte stores");
        f32x4 *const X = reinterpret_cast<f32x4 *>(xch);
#pragma unroll
        for (int dw = 0; dw < 4; ++dw) {
            if (dw == wave) continue;                   // wave-uniform
            const int slot = wave < dw ? wave : wave - 1;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[j][0][4 * dw + rr];
                X[((dw * 3 + slot) * 4 + rr) * 64 + lane] = o;
            }
        }
        __syncthreads();
        switch (wave) {
            case 0: wino_pool_store<0, Cfg>(a, job, acc, X, lane); break;
            case 1: wino_pool_store<1, Cfg>(a, job, acc, X, lane); break;
            case 2: wino_pool_store<2, Cfg>(a, job, acc, X, lane); break;
            default: wino_pool_store<3, Cfg>(a, job, acc, X, lane); break;
        }
        return;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            char *const dst = yb + ((r & 3) + 8 * (r >> 2)) * cs + voff[nb];
            if constexpr (V4) {
                f32x4 o;
                o[0] = acc[0][nb][r]; o[1] = acc[1][nb][r]; o[2] = acc[2][nb][r]; o[3] = acc[3][nb][r];
                if (ok0[nb]) *reinterpret_cast<f32x4 *>(dst) = o;
            } else {   // rows only 8-byte aligned (W % 4 == 2): the last tile of a row is half valid
                if (ok0[nb]) *reinterpret_cast<float2 *>(dst) = make_float2(acc[0][nb][r], acc[1][nb][r]);
                if (ok1[nb]) *reinterpret_cast<float2 *>(dst + 8) = make_float2(acc[2][nb][r], acc[3][nb][r]);
            }
        }
}

template <class Cfg, bool RES, bool PLANE>
__global__ void __launch_bounds__(256, 2)
conv3d_wino_kernel(const ConvArgs a) {
    constexpr int TD = Cfg::TD, TH = Cfg::TH, KC = Cfg::KC, NB = Cfg::NB, CH = Cfg::CH, TILE = Cfg::TILE, WF = Cfg::WF;
    using St = typename Cfg::St;
    using Ws = typename Cfg::Ws;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const WinoJob job = wino_decode_job(a, blockIdx.x, a.njobs, TD, TH, Cfg::TW);

    f32x16 acc[6][NB];
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][nb][r] = 0.0f;

    const int in_hw = a.Hin * a.Win, in_dhw = in_hw * a.Din;
    const float *xn = a.x + job.n * a.x_bs;
    // B-fragment base: lane&15 = output quad t (inputs 4t-1..4t+4 = image columns 4t+3..4t+8),
    // (lane>>4)&1 = row of the row pair, lane>>5 = k
    const int bbase = (lane >> 5) * CH + ((lane & 31) / Cfg::QUADS) * Cfg::IN_WV + 4 * (lane & (Cfg::QUADS - 1)) + Cfg::XOFF;
    float *const wlds = lds + 2 * TILE;
    float *const aff = wlds + 2 * WF;      // scale | bias of this job's 32 channels
    const int nchunks = a.nchunks_wino;
    const float *wg = a.wp_wino + (int64_t)job.cg * nchunks * WF;
    constexpr int NS = Cfg::NTAP * Cfg::KP;
    // prefetch pieces are retired over the last steps of a chunk (weights first: they were requested
    // first); the first FIRST steps (>= 2.5k cycles of MFMAs) cover the global-memory latency.
    // Measured on cfg2 conv2: FIRST = 2 -> 1.92 ms, 3 -> 1.88, 5 -> 1.85, 7 -> 1.85.
    constexpr int NPIECE = Ws::NIT + St::NIT;
    constexpr int FIRST = NS > 5 ? 5 : NS - 1;

    St st;
    st.init(tid, job.od0 - Cfg::KSD / 2, job.oh0 - 1, job.ow0 - Cfg::LPAD, a.Din, a.Hin, a.Win, in_hw, in_dhw);
    typename St::Vec pre[St::NIT];
        f32x4 wpre[Ws::NIT];
    st.load(xn, tid, a.Cin, pre);
    Ws::load(wg, tid, wpre);
    if (tid < 64) {
        float v = tid < 32 ? 1.0f : 0.0f;
        if (a.scale) v = (tid < 32 ? a.scale : a.bias)[job.cg * 32 + (tid & 31)];   // Cout % 32 == 0 (host)
        aff[tid] = v;
    }
    st.store(lds, tid, a.Cin, pre);
    Ws::store(wlds, tid, wpre);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        const int cin_left = a.Cin - (chunk + 1) * KC;
        if (more) {
            Ws::load(wg + (int64_t)(chunk + 1) * WF, tid, wpre);
            st.load(xn + (int64_t)(chunk + 1) * KC * in_dhw, tid, cin_left, pre);
        }
        float *const wnext = wlds + ((chunk + 1) & 1) * WF;
        float *const inext = lds + ((chunk + 1) & 1) * TILE;
        const unsigned m = st.store_mask(tid, cin_left);
        wino_compute_chunk<Cfg>(lds + (chunk & 1) * TILE, wlds + (chunk & 1) * WF + lane, bbase, wave, acc, [&](int s) {
            if (!more) return;
#pragma unroll
            for (int piece = 0; piece < NPIECE; ++piece) {
                // piece -> step: spread over steps FIRST..NS-1 (several per step when NPIECE > NS)
                const int at = FIRST + piece * (NS - FIRST) / NPIECE;
                if (at != s) continue;
                if (piece < Ws::NIT) Ws::store_one(wnext, tid, piece, wpre[piece]);
                else st.store_one(inext, tid, piece - Ws::NIT, m, pre[piece - Ws::NIT]);
            }
        });
        __syncthreads();
    }
    wino_epilogue<Cfg, RES, PLANE>(a, job, acc, aff, lane, wave);
}

// The same convolution with the chunks staged by LDS-DMA (`global_load_lds_dwordx4`: global -> LDS
// without a register hop).  That frees the 28 prefetch registers and the LDS write pass, which is what
// lets a wave carry TWO row pairs (12 accumulators = 192 registers): 108 MFMAs between barriers
// instead of 54, and a 4x4x64 tile whose halo is 2.25x instead of 3x.  A DMA wave-instruction writes
// 64 x 16 bytes to consecutive LDS addresses (wave-uniform base in M0), exactly the Stager's
// item order; padding pieces and channels beyond Cin read a 16-byte zero constant instead.
// `__syncthreads()` drains the DMA (hipcc emits vmcnt(0) in front of the barrier while one is in
// flight), so chunk c+1 lands while chunk c is multiplied and is visible after the chunk's barrier.
__device__ const float g_zero16[4] __attribute__((aligned(16))) = {0.0f, 0.0f, 0.0f, 0.0f};

// LDS-DMA of one chunk of conv3d_wino_dma_kernel (weights -> wbuf, input rows -> ibuf) from the per-thread source
// pointers xsrc[St::NIT] / wsrc[WNIT], which then advance to the next chunk; cin_left = channels from this chunk's first
// one to the layer's last (an odd Cin leaves the last chunk half empty: those pieces read the zero constant).
template <class Cfg>
__device__ __forceinline__ void wino_dma_issue(const float **xsrc, const float **wsrc, unsigned vmask, int64_t xstep, int cin_left,
                                               float *ibuf, float *wbuf, int tid) {
    using St = typename Cfg::St;
    constexpr int WITEMS = Cfg::WF / 4, WNIT = (WITEMS + 255) / 256;
    const int wbase = tid & ~63;           // first item of this wave in a 256-item round
#pragma unroll
    for (int it = 0; it < WNIT; ++it) {
        if (WITEMS % 256 == 0 || it * 256 + tid < WITEMS)
            __builtin_amdgcn_global_load_lds(wsrc[it], wbuf + 4 * (it * 256 + wbase), 16, 0, 0);
        wsrc[it] += Cfg::WF;
    }
#pragma unroll
    for (int it = 0; it < St::NIT; ++it) {
        const float *src = xsrc[it];
        if (cin_left < Cfg::KC && (it * 256 + tid) / (St::RQ * Cfg::IN_D * Cfg::IN_H) >= cin_left) src = g_zero16;
        if (St::ITEMS % 256 == 0 || it * 256 + tid < St::ITEMS)
            __builtin_amdgcn_global_load_lds(src, ibuf + 4 * (it * 256 + wbase), 16, 0, 0);
        if ((vmask >> it) & 1u) xsrc[it] += xstep;
    }
}

template <class Cfg, bool RES, bool PLANE, int XMODE = 0>
__global__ void __launch_bounds__(256, Cfg::OCC)
conv3d_wino_dma_kernel(const ConvArgs a) {
    constexpr bool HEAD = XMODE == 1;
    constexpr int TD = Cfg::TD, TH = Cfg::TH, KC = Cfg::KC, NB = Cfg::NB, CH = Cfg::CH, TILE = Cfg::TILE, WF = Cfg::WF;
    using St = typename Cfg::St;
    static_assert(Cfg::PIECE == 4, "LDS-DMA moves 16-byte pieces");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const WinoJob job = wino_decode_job(a, blockIdx.x, a.njobs, TD, TH, Cfg::TW);

    f32x16 acc[6][NB];
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][nb][r] = 0.0f;

    const int in_hw = a.Hin * a.Win, in_dhw = in_hw * a.Din;
    const float *xn = a.x + job.n * a.x_bs;
    const int bbase = (lane >> 5) * CH + ((lane & 31) / Cfg::QUADS) * Cfg::IN_WV + 4 * (lane & (Cfg::QUADS - 1)) + Cfg::XOFF;
    float *const wlds = lds + 2 * TILE;
    float *const aff = wlds + 2 * WF;      // scale | bias of this job's 32 channels
    const int nchunks = a.nchunks_wino;
    const float *wg = a.wp_wino + (int64_t)job.cg * nchunks * WF;

    St st;
    st.init(tid, job.od0 - Cfg::KSD / 2, job.oh0 - 1, job.ow0 - Cfg::LPAD, a.Din, a.Hin, a.Win, in_hw, in_dhw);
    constexpr int WITEMS = WF / 4, WNIT = (WITEMS + 255) / 256;
    // Per-thread DMA sources live in registers for the whole job and advance by one chunk per issue (chunks are issued in
    // order): beside fp32 MFMAs a VALU instruction costs ~7 cycles of matrix-pipe time
    // (profiles/r2/mfma_f32_issue_microbench.txt), and recomputing seven 64-bit addresses with their zero-fill selects
    // was ~35 of them per chunk.  Padding pieces keep pointing at the zero constant.
    const float *xsrc[St::NIT], *wsrc[WNIT];
#pragma unroll
    for (int it = 0; it < St::NIT; ++it) xsrc[it] = ((st.vmask >> it) & 1u) ? xn + st.off[it] : g_zero16;
#pragma unroll
    for (int it = 0; it < WNIT; ++it) wsrc[it] = wg + 4 * (it * 256 + tid);
    const int64_t xstep = (int64_t)KC * in_dhw;
    wino_dma_issue<Cfg>(xsrc, wsrc, st.vmask, xstep, a.Cin, lds, wlds, tid);
    if (tid < 64) {
        float v = tid < 32 ? 1.0f : 0.0f;
        if (a.scale) v = (tid < 32 ? a.scale : a.bias)[job.cg * 32 + (tid & 31)];   // Cout % 32 == 0 (host)
        aff[tid] = v;
    } else if (HEAD && tid < 96) {
        aff[tid] = a.head_w[job.cg * 32 + (tid - 64)];
    }
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk + 1 < nchunks)
            wino_dma_issue<Cfg>(xsrc, wsrc, st.vmask, xstep, a.Cin - (chunk + 1) * KC, lds + ((chunk + 1) & 1) * TILE,
                                wlds + ((chunk + 1) & 1) * WF, tid);
        wino_compute_chunk<Cfg>(lds + (chunk & 1) * TILE, wlds + (chunk & 1) * WF + lane, bbase, wave, acc, [](int) {});
        __syncthreads();
    }
    static_assert(XMODE != 2 || 48 * 64 * 16 <= Cfg::LDS_BYTES, "the pooled epilogue trades 48 KB through the image / weight buffers");
    static_assert(XMODE != 3 || 32 * 256 * 4 <= Cfg::LDS_BYTES, "the statistics epilogue trades 32 KB through the image / weight buffers");
    wino_epilogue<Cfg, RES, PLANE, XMODE>(a, job, acc, aff, lane, wave, lds);
}

// k5 / k7 Winograd kernel (WinoKCfg above).  Phase p = (chunk, kd): weights of phase p+1 are DMA'd
// while phase p is multiplied; the input image of the next chunk is DMA'd after the chunk's last
// phase (single buffer: that load is exposed to this workgroup and hidden by the co-resident one; a
// chunk is 7 x 70 = 490 MFMAs per wave for k7).
// (the plain k5 form fits 168 VGPRs and 41 KB of LDS: three workgroups per CU, 0.78 -> 0.72 ms on the local conv2)
template <class Cfg, bool RES>
__global__ void __launch_bounds__(256, (Cfg::KS == 5 && Cfg::DIL == 1) ? 3 : 2)
conv3d_winok_kernel(const ConvArgs a) {
    constexpr int KS = Cfg::KS, KC = Cfg::KC, CH = Cfg::CH, TILE = Cfg::TILE, WF = Cfg::WF, P = Cfg::NPOS;
    using St = typename Cfg::St;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const WinoJob job = wino_decode_job(a, blockIdx.x, a.njobs, Cfg::TD, Cfg::TH, Cfg::TW);

    f32x16 acc[P][1];
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][0][r] = 0.0f;

    const int in_hw = a.Hin * a.Win, in_dhw = in_hw * a.Din;
    const float *xn = a.x + job.n * a.x_bs;
    // B-fragment base: lane&7 = output quad t, (lane&31)>>3 = row of the wave's depth slice, lane>>5 = k;
    // a quad's inputs x[4t-PAD .. 4t+3+PAD] sit at image columns 4t+XOFF .. inside three 16-byte reads at 4t
    // (dilation 2: lanes 2q and 2q+1 are the even / odd parity of the 8 outputs at 8q, both read at 8q)
    // (stride 2: output row (wave, r) reads input rows 2*wave + kd, 2*r + kh; a quad's 12 inputs start at 8t)
    const int bbase = (lane >> 5) * CH +
                      Cfg::STRIDE * (wave * Cfg::IN_H + ((lane & 31) >> 3)) * Cfg::IN_WV +
                      (Cfg::DIL == 2 ? 8 * ((lane & 7) >> 1) : 4 * Cfg::STRIDE * (lane & 7));
    constexpr int WPH = Cfg::WPH, WBLK = WPH * WF;             // weights resident per phase: WPH kernel depth slices
    float *const wlds = lds + TILE;
    float *const aff = wlds + 2 * WBLK;
    const int nchunks = a.nchunks_wino, nphase = nchunks * (KS / WPH);
    const float *wg = a.wp_wino + (int64_t)job.cg * nphase * WBLK;
    constexpr int SLICE = Cfg::DIL * Cfg::IN_H * Cfg::IN_WV;     // image floats per kernel depth step

    St st;
    st.init(tid, job.od0 * Cfg::STRIDE - Cfg::PAD, job.oh0 * Cfg::STRIDE - Cfg::PAD, job.ow0 * Cfg::STRIDE - Cfg::LPAD,
            a.Din, a.Hin, a.Win, in_hw, in_dhw);
    constexpr int WITEMS = WBLK / 4, WNIT = (WITEMS + 255) / 256;
    const int wbase = tid & ~63;
    auto issue_w = [&](int ph, int b) {
        const float *wc = wg + (int64_t)ph * WBLK;
        float *const wbuf = wlds + b * WBLK;
#pragma unroll
        for (int it = 0; it < WNIT; ++it) {
            const int i = it * 256 + tid;
            if (WITEMS % 256 == 0 || i < WITEMS)
                __builtin_amdgcn_global_load_lds(wc + 4 * i, wbuf + 4 * (it * 256 + wbase), 16, 0, 0);
        }
    };
    auto issue_img = [&](int chunk) {
        const float *xc = xn + (int64_t)chunk * KC * in_dhw;
        const unsigned m = st.store_mask(tid, a.Cin - chunk * KC);
#pragma unroll
        for (int it = 0; it < St::NIT; ++it) {
            const int i = it * 256 + tid;
            const float *src = ((m >> it) & 1u) ? xc + st.off[it] : g_zero16;
            if (St::ITEMS % 256 == 0 || i < St::ITEMS)
                __builtin_amdgcn_global_load_lds(src, lds + 4 * (it * 256 + wbase), 16, 0, 0);
        }
    };
    issue_img(0);
    issue_w(0, 0);
    if (tid < 64) {
        float v = tid < 32 ? 1.0f : 0.0f;
        if (a.scale) v = (tid < 32 ? a.scale : a.bias)[job.cg * 32 + (tid & 31)];   // Cout % 32 == 0 (host)
        aff[tid] = v;
    }
    __syncthreads();
    int ph = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        for (int kd0 = 0; kd0 < KS; kd0 += WPH, ++ph) {
            if (ph + 1 < nphase) issue_w(ph + 1, (ph + 1) & 1);
#pragma unroll
            for (int k = 0; k < WPH; ++k) {
                const float *wl = wlds + (ph & 1) * WBLK + k * WF + lane;
                if constexpr (Cfg::STRIDE == 2) winos2_compute_phase<Cfg>(lds + (kd0 + k) * SLICE, wl, bbase, acc);
                else winok_compute_phase<Cfg>(lds + (kd0 + k) * SLICE, wl, bbase, lane, acc);
            }
            __syncthreads();
        }
        if (chunk + 1 < nchunks) {
            issue_img(chunk + 1);
            __syncthreads();
        }
    }
    wino_epilogue<Cfg, RES, false>(a, job, acc, aff, lane, wave);
}

// ------------------------------------------------------------------------------------ k3 / stride 2, slice-pipelined refill (r3)
// conv3d_winok_kernel<WinoS2Cfg> refills its single-buffered 44 KB image once per 2-channel chunk, after the chunk's last
// MFMA and in front of a barrier that drains the DMA: ~2 us of exposed refill per 2.2 us of MFMAs, which the second
// resident workgroup can only partly fill (pipe 0.54-0.59).  A second image buffer would cost that second workgroup.
// This form keeps ONE buffer and refills it SLICE BY SLICE as the depth phases release the slices:
//   wave w owns output depth w and reads input depth 2w + kd, so phase kd = 0 reads depth slices {0,2,4,6}, kd = 2 reads
//   {2,4,6,8} and kd = 1 reads {1,3,5,7}.  With the phases ordered A = kd 0, B = kd 2, C = kd 1:
//     start of A(c): issue the weights of chunk c+1 and the ODD slices of chunk c (last read in C(c-1));
//     start of B(c): issue slice 0 of chunk c+1 (only A reads it);
//     start of C(c): issue slices 2, 4, 6 and then 8 of chunk c+1 (B was their last reader);
//   so every slice has at least one full phase (27 MFMAs per wave = 1.7k cycles) of MFMAs between its issue and its first
//   use, the odd ones two.  LDS-DMA stays in flight across s_barrier; a phase ends with a COUNTED `s_waitcnt vmcnt(N)`
//   that only waits for the pieces the next phase reads, followed by s_barrier (hipcc's __syncthreads() would wait for
//   vmcnt(0)).  vmcnt(N) guarantees everything but the wave's N youngest operations: N is the SMALLEST number of DMA
//   instructions any wave issues after the pieces in question (waves that issued more just wait for a few of those too).
// DMA instructions are whole: the LDS image is [depth slice][channel][640 floats] (a 9 x 68 channel plane padded from
// 612 to 640 floats), so a slice is exactly 5 x 64 pieces of 16 bytes -- wave w issues instruction w of every slice,
// wave dd % 4 also instruction 4 of slice dd -- and the 864-piece weight block of a chunk goes as 14 whole instructions
// (the last one overlaps its predecessor by half: same bytes to the same place).  No lane predicates anywhere near a DMA
// (a divergent `if` around one lets the compiler duplicate its neighbours into both arms, which changes the count).
template <int STORE_PIECE_ = 4>
struct WinoS2PipeCfg : WinoS2Cfg<2, STORE_PIECE_> {
    using Base = WinoS2Cfg<2, STORE_PIECE_>;
    static constexpr int PLANE = Base::IN_H * Base::IN_WV;      // 612 floats of a channel's rows in one depth slice
    static constexpr int CH = 640;                               // channel stride inside a depth slice (padded)
    static constexpr int DSLICE = Base::KC * CH;                 // 1280 floats = 320 pieces = 5 DMA instructions
    static constexpr int WBLK = 3 * Base::WF;                    // 3456 floats = 864 pieces
    static constexpr int LDS_BYTES = (Base::IN_D * DSLICE + 2 * WBLK) * 4;
    static_assert(CH >= PLANE && CH % 4 == 0 && DSLICE == 5 * 256 && WBLK / 4 > 13 * 64 - 64 && WBLK / 4 <= 14 * 64, "piece maps");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt_then_barrier() {
    // all but the N youngest vector-memory operations of this wave (here: LDS-DMA instructions) have landed, then the
    // workgroup barrier makes every wave's landed pieces visible; "memory": the compiler moves no LDS access across
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <class Cfg, bool RES, int XMODE = 0>
__global__ void __launch_bounds__(256, 2)
conv3d_winos2_pipe_kernel(const ConvArgs a) {
    constexpr int KC = Cfg::KC, WF = Cfg::WF, IN_WV = Cfg::IN_WV, RQ = IN_WV / 4, IN_D = Cfg::IN_D;
    constexpr int CHS = Cfg::CH, PLANE = Cfg::PLANE, DSLICE = Cfg::DSLICE, WBLK = Cfg::WBLK, WPIECES = WBLK / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WinoJob job = wino_decode_job(a, blockIdx.x, a.njobs, Cfg::TD, Cfg::TH, Cfg::TW);

    f32x16 acc[9][1];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][0][r] = 0.0f;

    const int in_hw = a.Hin * a.Win, in_dhw = in_hw * a.Din;
    const float *xn = a.x + job.n * a.x_bs;
    const int id0 = job.od0 * 2 - 1, ih0 = job.oh0 * 2 - 1, ix0 = job.ow0 * 2 - Cfg::LPAD;
    // B fragments: lane>>5 = channel of the pair, (lane&31)>>3 = output row, lane&7 = quad; wave = output depth
    const int bbase = (lane >> 5) * CHS + 2 * wave * DSLICE + 2 * ((lane & 31) >> 3) * IN_WV + 8 * (lane & 7);
    float *const wlds = lds + IN_D * DSLICE;
    float *const aff = wlds + 2 * WBLK;
    const int nchunks = a.nchunks_wino;
    const float *wg = a.wp_wino + (int64_t)job.cg * nchunks * WBLK;

    // ---- piece maps (constant over slices and chunks).  Piece p of a slice = (channel kc = p / 160, then 16-byte piece
    // r = p % 160 of that channel's padded plane: row r / 17, column piece r % 17; r >= 153 is padding).
    // This thread moves piece 64*wave + lane of every slice and piece 256 + lane of the slices with dd % 4 == wave.
    auto piece_geom = [&](int p, unsigned &off, bool &ok, bool &second) {
        const int kc = p / (CHS / 4), r = p - kc * (CHS / 4), hh = r / RQ, q = r - hh * RQ;
        const int gh = ih0 + hh, gw = ix0 + 4 * q;
        second = kc != 0;
        ok = r < PLANE / 4 && (unsigned)gh < (unsigned)a.Hin && (unsigned)gw < (unsigned)a.Win;
        off = ok ? (unsigned)(kc * in_dhw + gh * a.Win + gw) : 0u;
    };
    unsigned off_a, off_b;
    bool ok_a, ok_b, second_a, second_b;
    piece_geom(64 * wave + lane, off_a, ok_a, second_a);
    piece_geom(256 + lane, off_b, ok_b, second_b);
    unsigned dmask = 0;      // depth slices inside the tensor (wave-uniform)
#pragma unroll
    for (int dd = 0; dd < IN_D; ++dd) dmask |= ((unsigned)(id0 + dd) < (unsigned)a.Din ? 1u : 0u) << dd;

    auto issue_slice = [&](int chunk, int dd) {
        const float *xc = xn + (int64_t)chunk * KC * in_dhw + (int64_t)(id0 + dd) * in_hw;
        const bool dok = (dmask >> dd) & 1u;
        const bool odd_tail = a.Cin - chunk * KC < KC;        // odd Cin: the last chunk's second channel is zeros
        float *const dst = lds + dd * DSLICE;
        const float *sa = (dok && ok_a && !(odd_tail && second_a)) ? xc + off_a : g_zero16;
        __builtin_amdgcn_global_load_lds(sa, dst + 4 * (64 * wave), 16, 0, 0);
        if (wave == (dd & 3)) {      // wave-uniform: a scalar branch, every lane of the wave takes part
            const float *sb = (dok && ok_b && !(odd_tail && second_b)) ? xc + off_b : g_zero16;
            __builtin_amdgcn_global_load_lds(sb, dst + 4 * 256, 16, 0, 0);
        }
    };
    // weights: instructions 0..13 of 64 pieces; wave w issues w, w+4, w+8 and (w = 0: 12, w = 2: 13); instruction 13
    // starts at piece WPIECES - 64 (overlapping 12)
    auto issue_w = [&](int chunk) {
        const float *wc = wg + (int64_t)chunk * WBLK;
        float *const wbuf = wlds + (chunk & 1) * WBLK;
#pragma unroll
        for (int it = 0; it < 3; ++it)
            __builtin_amdgcn_global_load_lds(wc + 4 * (it * 256 + tid), wbuf + 4 * (it * 256 + 64 * wave), 16, 0, 0);
        if (wave == 0) __builtin_amdgcn_global_load_lds(wc + 4 * (768 + lane), wbuf + 4 * 768, 16, 0, 0);
        if (wave == 2) __builtin_amdgcn_global_load_lds(wc + 4 * (WPIECES - 64 + lane), wbuf + 4 * (WPIECES - 64), 16, 0, 0);
    };
    // DMA instructions per wave.  weights: {4,3,4,3}.  slice dd: 1, +1 for wave dd % 4.
    //   odd slices {1,3,5,7}: {4,6,4,6};  slice 0: {2,1,1,1};  slices {2,4,6}: {4,3,5,3};  slice 8: {2,1,1,1}
    constexpr int AFTER_SLICE8_MORE = 3 + 4;      // min over waves of weights + odd slices   (end of A, next chunk exists)
    constexpr int AFTER_SLICE8_LAST = 4;          // min over waves of the odd slices         (end of A, last chunk)
    constexpr int AFTER_ODD = 1;                  // min over waves of slice 0                (end of B)
    constexpr int AFTER_EVEN = 1;                 // min over waves of slice 8                (end of C)

    // prologue: weights of chunk 0 and its EVEN slices (the odd ones are issued at the start of phase A like in every chunk)
    issue_w(0);
#pragma unroll
    for (int dd = 0; dd < IN_D; dd += 2) issue_slice(0, dd);
    if (tid < 64) {
        float v = tid < 32 ? 1.0f : 0.0f;
        if (a.scale) v = (tid < 32 ? a.scale : a.bias)[job.cg * 32 + (tid & 31)];   // Cout % 32 == 0 (host)
        aff[tid] = v;
    }
    __syncthreads();       // drains the DMA (vmcnt(0)) and publishes aff

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        const float *wl = wlds + (chunk & 1) * WBLK + lane;
        // ---- phase A: kd = 0 on slices {0,2,4,6}
        if (more) issue_w(chunk + 1);
#pragma unroll
        for (int dd = 1; dd < IN_D; dd += 2) issue_slice(chunk, dd);
        winos2_compute_phase<Cfg>(lds + 0 * DSLICE, wl + 0 * WF, bbase, acc);
        // B reads slice 8 (issued last in C(chunk-1), or by the prologue): everything older than this phase's issues
        if (more) wait_vmcnt_then_barrier<AFTER_SLICE8_MORE>(); else wait_vmcnt_then_barrier<AFTER_SLICE8_LAST>();
        // ---- phase B: kd = 2 on slices {2,4,6,8}
        if (more) issue_slice(chunk + 1, 0);
        winos2_compute_phase<Cfg>(lds + 2 * DSLICE, wl + 2 * WF, bbase, acc);
        // C reads the odd slices (issued at the start of A): all but slice 0 of the next chunk
        if (more) wait_vmcnt_then_barrier<AFTER_ODD>(); else wait_vmcnt_then_barrier<0>();
        // ---- phase C: kd = 1 on slices {1,3,5,7}
        if (more) {
#pragma unroll
            for (int dd = 2; dd < IN_D; dd += 2) issue_slice(chunk + 1, dd);       // 2, 4, 6, then 8
        }
        winos2_compute_phase<Cfg>(lds + 1 * DSLICE, wl + 1 * WF, bbase, acc);
        // A(chunk+1) reads slices {0,2,4,6} and the next weights: all but slice 8
        if (more) wait_vmcnt_then_barrier<AFTER_EVEN>();
    }
    static_assert(XMODE != 3 || 32 * 256 * 4 <= Cfg::LDS_BYTES, "the statistics epilogue trades 32 KB through the image buffer");
    wino_epilogue<Cfg, RES, false, XMODE>(a, job, acc, aff, lane, wave, lds);
}

// ------------------------------------------------------------------------------------ deconv
// ConvTranspose3d(k=3, s=2, p=1, op=1): out[o] = sum_i sum_t x[i] * w[t], o = 2i - 1 + t.
// Per dimension, output parity 0 uses tap 1 at input offset 0; parity 1 uses tap 2 at offset 0
// and tap 0 at offset +1.
template <int MI_, int TD_, int TH_, int KC_, int PIECE_ = 4>
struct DeconvCfg {
    static constexpr int MI = MI_, TD = TD_, TH = TH_, KC = KC_, PIECE = PIECE_;
    static constexpr int IN_D = TD + 1, IN_H = TH + 1, IN_WV = 36;   // 33 needed columns -> 9 float4 pieces
    using St = Stager<KC, IN_D, IN_H, IN_WV, PIECE>;
    static constexpr int CH = St::CH, TILE = St::TILE;
    static constexpr int NB = TD * TH / 4;
    static constexpr int KP = KC / 2;
    static constexpr int WF_MAX = 12 * KP * 64 * MI;                 // heaviest parity class: 12 taps
    static constexpr int BUF = TILE + WF_MAX;
    static constexpr int LDS_BYTES = BUF * 4 * 2;                    // double buffered
    static_assert(TD * TH % 4 == 0, "rows must split over 4 waves");
};

__host__ __device__ constexpr int deconv_class_offset(int cls) {  // cls = pd*2 + ph
    return cls == 0 ? 0 : cls == 1 ? 3 : cls == 2 ? 9 : 15;
}
__host__ __device__ constexpr int deconv_class_ntaps(int cls) { return cls == 0 ? 3 : cls == 3 ? 12 : 6; }

template <class Cfg, int PD, int PH>
__device__ __forceinline__ void deconv_compute_chunk(const float *__restrict__ buf, const float *__restrict__ wc,
                                                     int bbase, int wave, f32x16 (&acc)[2][Cfg::NB][Cfg::MI]) {
    constexpr int MI = Cfg::MI, TH = Cfg::TH, KP = Cfg::KP, NB = Cfg::NB, IN_H = Cfg::IN_H, IN_WV = Cfg::IN_WV;
    constexpr int CH = Cfg::CH, ND = PD ? 2 : 1, NH = PH ? 2 : 1;
    int tix = 0;  // running tap index inside this class (matches the packing order)
#pragma unroll
    for (int jd = 0; jd < ND; ++jd) {
        const int dld = (PD && jd == 1) ? 1 : 0;  // input offset (+1 for tap 0)
#pragma unroll
        for (int jh = 0; jh < NH; ++jh) {
            const int dlh = (PH && jh == 1) ? 1 : 0;
#pragma unroll
            for (int jw = 0; jw < 3; ++jw, ++tix) {
                // jw = 0: width parity 0 (tap 1, offset 0); 1: parity 1 (tap 2, offset 0);
                // 2: parity 1 (tap 0, offset +1)
                const int pw = jw == 0 ? 0 : 1;
                const int dlw = jw == 2 ? 1 : 0;
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) {
                    float af[MI];
#pragma unroll
                    for (int m = 0; m < MI; ++m) af[m] = wc[(tix * KP + kp) * 64 * MI + m];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int row = wave * NB + nb;
                        const int dd = row / TH, hh = row % TH;
                        const float bf = buf[bbase + kp * 2 * CH + ((dd + dld) * IN_H + hh + dlh) * IN_WV + dlw];
#pragma unroll
                        for (int m = 0; m < MI; ++m)
                            acc[pw][nb][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bf, acc[pw][nb][m], 0, 0, 0);
                    }
                }
            }
        }
    }
}

// Fast epilogue of one parity class (toolkit comment above): lane holds the output pair
// (2*iw, 2*iw+1) of row (2*id+PD, 2*ih+PH) for 16 channels.
// HEAD: the layer's output feeds a 1x1x1 convolution to ONE channel and nothing else (the global model's
// classifier after the hourglass, stereo_volume.py): the 32-channel dot product is formed here -- 16 channels
// per lane, the other 16 in lane ^ 32 -- and only the single-channel result is stored (1/32 of the bytes; the
// full-resolution tensor and the classifier's pass over it disappear).
template <class Cfg, int PD, int PH, bool RES, bool HEAD = false, bool STATS = false>
__device__ __forceinline__ void deconv_epilogue_fast(const ConvArgs &a, f32x16 (&acc)[2][Cfg::NB][Cfg::MI], int id0,
                                                     int ih0, int iw0, int cg, int64_t n, int lane, int wave,
                                                     float *xch = nullptr) {
    constexpr int MI = Cfg::MI, TH = Cfg::TH, NB = Cfg::NB;
    const int iw = iw0 + (lane & 31), half = lane >> 5;
    const int out_hw = a.Hout * a.Wout, out_dhw = out_hw * a.Dout;
    const int64_t cs = (int64_t)out_dhw * 4;
    const bool relu = (a.flags & SNVC_EPI_RELU) != 0, add_pre = (a.flags & SNVC_EPI_ADD_PRE) != 0,
               add_post = (a.flags & SNVC_EPI_ADD_POST) != 0;
    unsigned voff[NB];
    bool ok[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row = wave * NB + nb;
        const int id = id0 + row / TH, ih = ih0 + row % TH;
        ok[nb] = id < a.Din && ih < a.Hin && iw < a.Win;      // Win is even (host): the pair lane shares it
        const int sp = ok[nb] ? (2 * id + PD) * out_hw + (2 * ih + PH) * a.Wout + 2 * iw : 0;
        voff[nb] = 4u * (unsigned)(4 * half * out_dhw + sp);
    }
#pragma unroll
    for (int m = 0; m < MI; ++m) {
        const int cbase = __builtin_amdgcn_readfirstlane((cg * MI + m) * 32);
        const char *const rb = RES ? reinterpret_cast<const char *>(a.res + n * a.r_bs) + cbase * cs : nullptr;
        ChanAffine f;
        load_affine(a, cbase, lane, f);
        // residual batches: a whole accumulator (16 pairs) when the kernel has registers to spare (MI = 1), half
        // of one otherwise; every batch is one exposed memory round trip, so fewer and larger is better
        constexpr int RBATCH = MI == 1 ? 16 : 8;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += RBATCH) {
                float2 rv[RES ? RBATCH : 1];
#pragma unroll
                for (int q = 0; q < RBATCH; ++q) {
                    const int cl = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                    if (RES) rv[q] = *reinterpret_cast<const float2 *>(rb + cl * cs + voff[nb]);
                }
#pragma unroll
                for (int q = 0; q < RBATCH; ++q) {
                    const int r = r0 + q;
                    acc[0][nb][m][r] = act_f(acc[0][nb][m][r] * f.sc[r] + f.bi[r], RES ? rv[q].x : 0.0f, add_pre, relu, add_post);
                    acc[1][nb][m][r] = act_f(acc[1][nb][m][r] * f.sc[r] + f.bi[r], RES ? rv[q].y : 0.0f, add_pre, relu, add_post);
                }
                // one batch of loads in flight at a time (register pressure): pin this batch's results
                // before the next batch's loads may be issued
#pragma unroll
                for (int q = 0; q < RBATCH; ++q)
                    asm volatile("" : "+v"(acc[0][nb][m][r0 + q]), "+v"(acc[1][nb][m][r0 + q])::"memory");
            }
    }
    if constexpr (STATS) {
        // Batch statistics of the layer's result while it is in registers (train-mode BatchNorm, as wino_epilogue's XMODE 3):
        // fp32 (sum, sum of squares) of a lane's <= 2*NB in-range values per channel, added over the 32 lanes of its half-wave
        // with xor shuffles, over the four waves through 1 KB of LDS, in fp64 -> a.stats[workgroup slot][32 channels][2].
        static_assert(MI == 1, "one 32-channel group per workgroup");
        float *const P = xch;                       // [4 waves][64]: channel (r, half) x (sum | sum of squares)
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float su = 0.0f, sq = 0.0f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float v0 = ok[nb] ? acc[0][nb][0][r] : 0.0f, v1 = ok[nb] ? acc[1][nb][0][r] : 0.0f;
                su += v0 + v1;
                sq += v0 * v0 + v1 * v1;
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { su += __shfl_xor(su, o); sq += __shfl_xor(sq, o); }
            if ((lane & 31) == 0) {
                const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;
                P[wave * 64 + 2 * ch] = su;
                P[wave * 64 + 2 * ch + 1] = sq;
            }
        }
        __syncthreads();
        const int tid = wave * 64 + lane;
        if (tid < 64) {
            double t = 0.0;
            for (int w = 0; w < 4; ++w) t += (double)P[w * 64 + tid];
            const int64_t slot = ((n * gridDim.x + blockIdx.x) * gridDim.y + cg);
            a.stats[slot * 64 + tid] = t;
        }
    }
    if constexpr (HEAD) {
        static_assert(MI == 1, "the fused head covers one 32-channel group");
        float hw[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) hw[r] = a.head_w[(r & 3) + 8 * (r >> 2) + 4 * half];
        float *const yh = a.y_head + n * (int64_t)out_dhw;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                d0 = __builtin_fmaf(hw[r], acc[0][nb][0][r], d0);
                d1 = __builtin_fmaf(hw[r], acc[1][nb][0][r], d1);
            }
            d0 += __shfl_xor(d0, 32);
            d1 += __shfl_xor(d1, 32);
            const int row = wave * NB + nb;
            const int id = id0 + row / TH, ih = ih0 + row % TH;
            if (ok[nb] && half == 0)
                *reinterpret_cast<float2 *>(yh + (int64_t)(2 * id + PD) * out_hw + (2 * ih + PH) * a.Wout + 2 * iw) =
                    make_float2(d0, d1);
        }
    } else {
#pragma unroll
        for (int m = 0; m < MI; ++m) {
            const int cbase = __builtin_amdgcn_readfirstlane((cg * MI + m) * 32);
            char *const yb = reinterpret_cast<char *>(a.y + n * a.y_bs) + cbase * cs;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) store_pairs_x4(yb, cs, voff[nb], ok[nb], acc[0][nb][m], acc[1][nb][m], lane);
        }
    }
}

template <class Cfg, int EPI, int PD, int PH>   // EPI: 0 generic epilogue, 1 fast, 2 fast with residual, 3 / 4 = 1 / 2 + fused head
__device__ __forceinline__ void deconv_class_body(const ConvArgs &a, float *lds, int tile, int cg, int64_t n) {
    constexpr int MI = Cfg::MI, TD = Cfg::TD, TH = Cfg::TH, KC = Cfg::KC, KP = Cfg::KP, NB = Cfg::NB;
    constexpr int CH = Cfg::CH, TILE = Cfg::TILE;
    constexpr int CLS_OFF = deconv_class_offset(PD * 2 + PH);
    using St = typename Cfg::St;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tw = tile % a.tiles_w, th = (tile / a.tiles_w) % a.tiles_h, td = tile / (a.tiles_w * a.tiles_h);
    const int id0 = td * TD, ih0 = th * TH, iw0 = tw * 32;

    f32x16 acc[2][NB][MI];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int m = 0; m < MI; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][nb][m][r] = 0.0f;

    const int in_hw = a.Hin * a.Win, in_dhw = in_hw * a.Din;
    const float *xn = a.x + n * a.x_bs;
    constexpr int WF = deconv_class_ntaps(PD * 2 + PH) * KP * 64 * MI;   // this class's weight floats per chunk
    constexpr int WSTRIDE = 27 * KP * 64 * MI;                            // all classes, per chunk
    constexpr int BUF = Cfg::BUF;
    using Ws = WeightStager<WF>;
    const float *wg = a.wp + ((int64_t)cg * a.nchunks * 27 + CLS_OFF) * KP * 64 * MI;
    const int bbase = (lane >> 5) * CH + (lane & 31);

    if (a.vec) {
        St st;
        st.init(tid, id0, ih0, iw0, a.Din, a.Hin, a.Win, in_hw, in_dhw);
        typename St::Vec pre[St::NIT];
        f32x4 wpre[Ws::NIT];
        st.load(xn, tid, a.Cin, pre);
        Ws::load(wg, tid, wpre);
        st.store(lds, tid, a.Cin, pre);
        Ws::store(lds + TILE, tid, wpre);
        __syncthreads();
        for (int chunk = 0; chunk < a.nchunks; ++chunk) {
            const bool more = chunk + 1 < a.nchunks;
            const float *cur = lds + (chunk & 1) * BUF;
            if (more) {
                st.load(xn + (int64_t)(chunk + 1) * KC * in_dhw, tid, a.Cin - (chunk + 1) * KC, pre);
                Ws::load(wg + (int64_t)(chunk + 1) * WSTRIDE, tid, wpre);
            }
            deconv_compute_chunk<Cfg, PD, PH>(cur, cur + TILE + lane * MI, bbase, wave, acc);
            if (more) {
                float *nxt = lds + ((chunk + 1) & 1) * BUF;
                st.store(nxt, tid, a.Cin - (chunk + 1) * KC, pre);
                Ws::store(nxt + TILE, tid, wpre);
            }
            __syncthreads();
        }
    } else {
        for (int chunk = 0; chunk < a.nchunks; ++chunk) {
            __syncthreads();
            St::stage_scalar(xn + (int64_t)chunk * KC * in_dhw, lds, tid, id0, ih0, iw0, a.Din, a.Hin, a.Win, in_hw,
                             in_dhw, a.Cin - chunk * KC);
            f32x4 wpre[Ws::NIT];
            Ws::load(wg + (int64_t)chunk * WSTRIDE, tid, wpre);
            Ws::store(lds + TILE, tid, wpre);
            __syncthreads();
            deconv_compute_chunk<Cfg, PD, PH>(lds, lds + TILE + lane * MI, bbase, wave, acc);
        }
    }

    if constexpr (EPI != 0) {
        if constexpr (EPI == 5 && Cfg::MI == 1)
            deconv_epilogue_fast<Cfg, PD, PH, false, false, true>(a, acc, id0, ih0, iw0, cg, n, lane, wave, lds);
        else if constexpr (EPI >= 3 && Cfg::MI == 1)
            deconv_epilogue_fast<Cfg, PD, PH, EPI == 4, true>(a, acc, id0, ih0, iw0, cg, n, lane, wave);
        else
            deconv_epilogue_fast<Cfg, PD, PH, EPI == 2>(a, acc, id0, ih0, iw0, cg, n, lane, wave);
    } else {
    // generic epilogue: outputs (2*id + PD, 2*ih + PH, 2*iw + {0,1}) -> one 8-byte store per lane
    const int iw = iw0 + (lane & 31);
    const int64_t out_hw = (int64_t)a.Hout * a.Wout, out_dhw = out_hw * a.Dout;
    float *yn = a.y + n * a.y_bs;
    const float *rn = a.res ? a.res + n * a.r_bs : nullptr;
#pragma unroll
    for (int m = 0; m < MI; ++m) {
        const int cbase = cg * 32 * MI + m * 32;
        ChanAffine f;
        load_affine(a, cbase, lane, f);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int row = wave * NB + nb;
            const int id = id0 + row / TH, ih = ih0 + row % TH;
            const bool vox_ok = id < a.Din && ih < a.Hin && iw < a.Win;
            const int64_t sp = vox_ok ? (int64_t)(2 * id + PD) * out_hw + (int64_t)(2 * ih + PH) * a.Wout + 2 * iw : 0;
            float2 rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                co = co < a.Cout ? co : a.Cout - 1;
                rv[r] = rn ? *reinterpret_cast<const float2 *>(rn + co * out_dhw + sp) : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v0 = epilogue_f(acc[0][nb][m][r] * f.sc[r] + f.bi[r], rv[r].x, a.flags);
                const float v1 = epilogue_f(acc[1][nb][m][r] * f.sc[r] + f.bi[r], rv[r].y, a.flags);
                if (vox_ok && co < a.Cout) *reinterpret_cast<float2 *>(yn + co * out_dhw + sp) = make_float2(v0, v1);
            }
        }
    }
    }
}

// The single-group form with a fast epilogue fits 128 VGPRs and 30 KB of LDS: four workgroups per CU (its chunks
// are short, 12..48 MFMAs per wave, so resident waves are what hides the staging latency).
template <class Cfg, int EPI>
__global__ void __launch_bounds__(256, (Cfg::MI == 1 && EPI != 0) ? 4 : 2)
deconv3d_mfma_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ntiles = a.tiles_d * a.tiles_h * a.tiles_w;
    // heaviest class first (12 taps), lightest last: blockIdx.x = cls_order * ntiles + tile
    const int slot = blockIdx.x / ntiles;
    const int tile = xcd_remap(blockIdx.x - slot * ntiles, ntiles);
    const int cg = blockIdx.y;
    const int64_t n = blockIdx.z;
    if (a.dc_planar) {   // depth-1 layer: the two classes of depth parity 0 (output plane 0 is the only one)
        if (slot == 0) deconv_class_body<Cfg, EPI, 0, 1>(a, lds, tile, cg, n);
        else deconv_class_body<Cfg, EPI, 0, 0>(a, lds, tile, cg, n);
        return;
    }
    switch (slot) {  // wave-uniform
        case 0: deconv_class_body<Cfg, EPI, 1, 1>(a, lds, tile, cg, n); break;
        case 1: deconv_class_body<Cfg, EPI, 1, 0>(a, lds, tile, cg, n); break;
        case 2: deconv_class_body<Cfg, EPI, 0, 1>(a, lds, tile, cg, n); break;
        default: deconv_class_body<Cfg, EPI, 0, 0>(a, lds, tile, cg, n); break;
    }
}

// ------------------------------------------------------------------------------------ 1x1x1, tiny Cout
// The classifier (C -> 1, kernel 1) would waste 31/32 of an MFMA tile and is bound by reading its
// input once: a streaming kernel does it -- each lane owns 4 consecutive voxels (16-byte loads of
// every input channel, channel stride = D*H*W), weights in scalar registers, fused epilogue.
template <int COUT>
__global__ void __launch_bounds__(256)
pointwise_small_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ scale,
                       const float *__restrict__ bias, const float *__restrict__ res, float *__restrict__ y,
                       int Cin, int64_t S, int64_t x_bs, int64_t y_bs, int64_t r_bs, int flags) {
    const int64_t n = blockIdx.y;
    const float *xn = x + n * x_bs;
    float *yn = y + n * y_bs;
    const float *rn = res ? res + n * r_bs : nullptr;
    const int64_t S4 = S >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int ci = 0; ci < Cin; ++ci) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(xn + ci * S)[i];
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float wv = w[co * Cin + ci];   // uniform -> scalar load
                acc[co] += v * wv;
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float sc = scale ? scale[co] : 1.0f, bi = scale ? bias[co] : 0.0f;
            f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rn) r = reinterpret_cast<const f32x4 *>(rn + co * S)[i];
            f32x4 o;
            o.x = epilogue_f(acc[co].x * sc + bi, r.x, flags);
            o.y = epilogue_f(acc[co].y * sc + bi, r.y, flags);
            o.z = epilogue_f(acc[co].z * sc + bi, r.z, flags);
            o.w = epilogue_f(acc[co].w * sc + bi, r.w, flags);
            reinterpret_cast<f32x4 *>(yn + co * S)[i] = o;
        }
    }
}

// 1x1x1 FROM <= 2 channels (r6): the data gradient of the classifier (1 -> C: gx[c] = w[c] * gy, an outer product) ran on the MFMA tile
// kernel with 31/32 of its K empty (0.275 ms for the 736 MB it writes at cfg4); streamed instead: a lane owns 4 consecutive voxels,
// reads its CIN values once and writes every output channel's 16 bytes.
template <int CIN>
__global__ void __launch_bounds__(256)
pointwise_expand_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ scale,
                        const float *__restrict__ bias, const float *__restrict__ res, float *__restrict__ y,
                        int Cout, int64_t S, int64_t x_bs, int64_t y_bs, int64_t r_bs, int flags) {
    const int64_t n = blockIdx.y;
    const float *xn = x + n * x_bs;
    float *yn = y + n * y_bs;
    const float *rn = res ? res + n * r_bs : nullptr;
    const int64_t S4 = S >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) v[ci] = reinterpret_cast<const f32x4 *>(xn + ci * S)[i];
#pragma unroll 8
        for (int co = 0; co < Cout; ++co) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) acc += v[ci] * w[co * CIN + ci];      // uniform -> scalar loads
            const float sc = scale ? scale[co] : 1.0f, bi = scale ? bias[co] : 0.0f;
            f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rn) r = reinterpret_cast<const f32x4 *>(rn + co * S)[i];
            f32x4 o;
            o.x = epilogue_f(acc.x * sc + bi, r.x, flags);
            o.y = epilogue_f(acc.y * sc + bi, r.y, flags);
            o.z = epilogue_f(acc.z * sc + bi, r.z, flags);
            o.w = epilogue_f(acc.w * sc + bi, r.w, flags);
            reinterpret_cast<f32x4 *>(yn + co * S)[i] = o;
        }
    }
}

// ------------------------------------------------------------------------------------ 3x3x3 to ONE channel
// The occupancy head of the local model ends in Conv3d(32, 1, 3) + Sigmoid over the whole voxel grid
// (vernier.py:262-270): 27 x Cin multiply-adds per voxel, far too little for a 32-channel MFMA tile (the MFMA
// kernel spent 0.35 ms per crop with 31/32 of its tile empty).  A VALU kernel does it: a workgroup owns
// 4 x 8 x 32 output voxels, a thread one (h, w) column of 4 depths; 4 input channels at a time are staged in
// LDS (6 x 10 x 34 tile each); for every (channel, kh, kw) a thread reads its 6 depth values once and feeds the
// 3 kd taps of its 4 outputs (12 FMAs); weights come in through scalar loads (wave-uniform addresses).
constexpr int K3C1_TD = 4, K3C1_TH = 8, K3C1_KC = 4;
constexpr int K3C1_IND = K3C1_TD + 2, K3C1_INH = K3C1_TH + 2, K3C1_INW = 34;
constexpr int K3C1_CH = K3C1_IND * K3C1_INH * K3C1_INW;

__global__ void __launch_bounds__(256)
conv3d_k3_cout1_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ scale,
                       const float *__restrict__ bias, const float *__restrict__ res, float *__restrict__ y, int Cin,
                       int D, int H, int W, int tiles_h, int tiles_w, int64_t x_bs, int64_t y_bs, int64_t r_bs, int flags) {
    __shared__ float img[K3C1_KC * K3C1_CH];
    const int tid = threadIdx.x, ww = tid & 31, hh = tid >> 5;
    const int t = blockIdx.x;
    const int tw = t % tiles_w, th = (t / tiles_w) % tiles_h, td = t / (tiles_w * tiles_h);
    const int od0 = td * K3C1_TD, oh0 = th * K3C1_TH, ow0 = tw * 32;
    const int64_t n = blockIdx.y;
    const int64_t hw = (int64_t)H * W, dhw = hw * D;
    const float *xn = x + n * x_bs;
    float acc[K3C1_TD] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < Cin; c0 += K3C1_KC) {
        __syncthreads();   // previous channels consumed
        for (int e = tid; e < K3C1_KC * K3C1_CH; e += 256) {
            const int c = e / K3C1_CH, r = e - c * K3C1_CH;
            const int dd = r / (K3C1_INH * K3C1_INW), r2 = r - dd * (K3C1_INH * K3C1_INW);
            const int h2 = r2 / K3C1_INW, w2 = r2 - h2 * K3C1_INW;
            const int gd = od0 - 1 + dd, gh = oh0 - 1 + h2, gw = ow0 - 1 + w2;
            const bool ok = c0 + c < Cin && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const float v = xn[ok ? (c0 + c) * dhw + gd * hw + (int64_t)gh * W + gw : 0];
            img[e] = ok ? v : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < K3C1_KC; ++c) {
            if (c0 + c >= Cin) break;                     // uniform
            const float *wc = w + (int64_t)(c0 + c) * 27;  // [kd][kh][kw], wave-uniform -> scalar loads
            const float *pc = img + c * K3C1_CH + hh * K3C1_INW + ww;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    float v[K3C1_IND];
#pragma unroll
                    for (int dd = 0; dd < K3C1_IND; ++dd) v[dd] = pc[(dd * K3C1_INH + kh) * K3C1_INW + kw];
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
                        const float wv = wc[(kd * 3 + kh) * 3 + kw];
#pragma unroll
                        for (int o = 0; o < K3C1_TD; ++o) acc[o] = __builtin_fmaf(wv, v[o + kd], acc[o]);
                    }
                }
        }
    }
    const float sc = scale ? scale[0] : 1.0f, bi = scale ? bias[0] : 0.0f;
    const int oh = oh0 + hh, ow = ow0 + ww;
    if (oh < H && ow < W) {
#pragma unroll
        for (int o = 0; o < K3C1_TD; ++o) {
            const int od = od0 + o;
            if (od >= D) break;
            const int64_t sp = od * hw + (int64_t)oh * W + ow;
            const float r = res ? res[n * r_bs + sp] : 0.0f;
            y[n * y_bs + sp] = epilogue_f(acc[o] * sc + bi, r, flags);
        }
    }
}

// ------------------------------------------------------------------------------------ packing
// Conv:    packed[cg][chunk][tap][kp][half][i][m] = W[co = cg*32*MI + m*32 + i][ci = chunk*KC + 2kp + half][tap]
// Deconv:  same with tap enumerated class by class (see deconv_class_body) and
//          W indexed [ci][co][tap] (nn.ConvTranspose3d layout).
__global__ void pack_conv_weights_kernel(const float *__restrict__ w, float *__restrict__ packed,
                                         int Cout, int Cin, int taps, int MI, int KC, int nchunks,
                                         int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int KP = KC / 2;
    int64_t r = i;
    const int m = (int)(r % MI); r /= MI;
    const int ii = (int)(r % 32); r /= 32;
    const int half = (int)(r % 2); r /= 2;
    const int kp = (int)(r % KP); r /= KP;
    const int tap = (int)(r % taps); r /= taps;
    const int chunk = (int)(r % nchunks); r /= nchunks;
    const int cg = (int)r;
    const int co = cg * 32 * MI + m * 32 + ii, ci = chunk * KC + 2 * kp + half;
    packed[i] = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * taps + tap] : 0.0f;
}

__device__ __forceinline__ int deconv_tap_of_slot(int slot) {
    // slot in [0,27): class-major order used by deconv_class_body
    int cls, t;
    if (slot < 3) { cls = 0; t = slot; } else if (slot < 9) { cls = 1; t = slot - 3; }
    else if (slot < 15) { cls = 2; t = slot - 9; } else { cls = 3; t = slot - 15; }
    const int pd = cls >> 1, ph = cls & 1;
    const int nh = ph ? 2 : 1;
    const int jw = t % 3, jh = (t / 3) % nh, jd = t / (3 * nh);
    const int kd = pd ? (jd == 0 ? 2 : 0) : 1;
    const int kh = ph ? (jh == 0 ? 2 : 0) : 1;
    const int kw = jw == 0 ? 1 : (jw == 1 ? 2 : 0);
    return (kd * 3 + kh) * 3 + kw;
}

__global__ void pack_deconv_weights_kernel(const float *__restrict__ w, float *__restrict__ packed,
                                           int Cout, int Cin, int MI, int KC, int nchunks, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int KP = KC / 2;
    int64_t r = i;
    const int m = (int)(r % MI); r /= MI;
    const int ii = (int)(r % 32); r /= 32;
    const int half = (int)(r % 2); r /= 2;
    const int kp = (int)(r % KP); r /= KP;
    const int slot = (int)(r % 27); r /= 27;
    const int chunk = (int)(r % nchunks); r /= nchunks;
    const int cg = (int)r;
    const int co = cg * 32 * MI + m * 32 + ii, ci = chunk * KC + 2 * kp + half;
    packed[i] = (co < Cout && ci < Cin) ? w[((int64_t)ci * Cout + co) * 27 + deconv_tap_of_slot(slot)] : 0.0f;
}

// Winograd packing (k3, s1): packed[cg][chunk][tap9 = kd*3+kh][pos][kp][half][i] = U_pos of the three kw taps of
// W[co = cg*32 + i][ci = chunk*KC + 2kp + half][kd][kh][:]   (F(4,3): U = G g, formed in fp64, rounded once)
__global__ void pack_wino_weights_kernel(const float *__restrict__ w, float *__restrict__ packed, int Cout, int Cin,
                                         int KC, int nchunks, int64_t total, int ntap = 9) {   // ntap = 3: depth-1 layer [Cout][Cin][1][3][3]
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int KP = KC / 2;
    int64_t r = i;
    const int ii = (int)(r % 32); r /= 32;
    const int half = (int)(r % 2); r /= 2;
    const int kp = (int)(r % KP); r /= KP;
    const int pos = (int)(r % 6); r /= 6;
    const int tap9 = (int)(r % ntap); r /= ntap;
    const int chunk = (int)(r % nchunks); r /= nchunks;
    const int cg = (int)r;
    const int co = cg * 32 + ii, ci = chunk * KC + 2 * kp + half;
    double u = 0.0;
    if (co < Cout && ci < Cin) {
        const float *g = w + (((int64_t)co * Cin + ci) * ntap + tap9) * 3;
        const double g0 = g[0], g1 = g[1], g2 = g[2];
        switch (pos) {
            case 0: u = g0 / 4.0; break;
            case 1: u = -(g0 + g1 + g2) / 6.0; break;
            case 2: u = -(g0 - g1 + g2) / 6.0; break;
            case 3: u = g0 / 24.0 + g1 / 12.0 + g2 / 6.0; break;
            case 4: u = g0 / 24.0 - g1 / 12.0 + g2 / 6.0; break;
            default: u = g2; break;
        }
    }
    packed[i] = (float)u;
}

// F(4,KS) packing (k5 / k7, s1): packed[cg][chunk][kd][kh][pos][kp][half][i] = (G g)_pos of the KS kw taps of
// W[co = cg*32 + i][ci = chunk*KC + 2kp + half][kd][kh][:]
template <int KS>
__global__ void pack_winok_weights_kernel(const float *__restrict__ w, float *__restrict__ packed, int Cout, int Cin,
                                          int KC, int nchunks, int64_t total) {
    using T = WinoTables<KS>;
    constexpr int P = T::P;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int KP = KC / 2;
    int64_t r = i;
    const int ii = (int)(r % 32); r /= 32;
    const int half = (int)(r % 2); r /= 2;
    const int kp = (int)(r % KP); r /= KP;
    const int pos = (int)(r % P); r /= P;
    const int kh = (int)(r % KS); r /= KS;
    const int kd = (int)(r % KS); r /= KS;
    const int chunk = (int)(r % nchunks); r /= nchunks;
    const int cg = (int)r;
    const int co = cg * 32 + ii, ci = chunk * KC + 2 * kp + half;
    double u = 0.0;
    if (co < Cout && ci < Cin) {
        const float *g = w + ((((int64_t)co * Cin + ci) * KS + kd) * KS + kh) * KS;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            if (q != pos) continue;
#pragma unroll
            for (int k = 0; k < KS; ++k) u += T::g(q, k) * (double)g[k];
        }
    }
    packed[i] = (float)u;
}

// Stride-2 k3 packing: packed[cg][chunk][kd][kh][frag][kp][half][i], frag 0 = g1 (even phase), frag 1..5 = G2 (g0, g2)
__global__ void pack_winos2_weights_kernel(const float *__restrict__ w, float *__restrict__ packed, int Cout, int Cin,
                                           int KC, int nchunks, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int KP = KC / 2;
    int64_t r = i;
    const int ii = (int)(r % 32); r /= 32;
    const int half = (int)(r % 2); r /= 2;
    const int kp = (int)(r % KP); r /= KP;
    const int frag = (int)(r % 6); r /= 6;
    const int kh = (int)(r % 3); r /= 3;
    const int kd = (int)(r % 3); r /= 3;
    const int chunk = (int)(r % nchunks); r /= nchunks;
    const int cg = (int)r;
    const int co = cg * 32 + ii, ci = chunk * KC + 2 * kp + half;
    double u = 0.0;
    if (co < Cout && ci < Cin) {
        const float *g = w + ((((int64_t)co * Cin + ci) * 3 + kd) * 3 + kh) * 3;
        if (frag == 0) u = g[1];
        else u = wino::G2[frag - 1][0] * (double)g[0] + wino::G2[frag - 1][1] * (double)g[2];
    }
    packed[i] = (float)u;
}

// ------------------------------------------------------------------------------------ dispatch
struct Plan {
    int MI, KC, TD, TH;  // tile choice
    int groups, nchunks;
    int tiles_d, tiles_h, tiles_w;
    int kind;  // index into the instantiation table
};

enum Kind {
    K1_M1, K1_M2,
    K3_M1, K3_M2,
    K3S2_M1, K3S2_M2,
    K5_M1, K5_M2,
    K5D2_M1, K5D2_M2,
    K7_M1, K7_M2,
    DC_M1, DC_M2, DCP_M1,
    P1_M1S, P1S2_M1S, P3_M1S, P3S2_M1S, P7_M1S,                          // depth-1 (2D) layers, 1 x 4 x 32 tiles
    KIND_NONE
};

//                       KS S  D  MI TD TH KC  DB   OCC KDG
using CfgK1M1   = ConvCfg<1, 1, 1, 1, 4, 4, 8, true>;
using CfgK1M2   = ConvCfg<1, 1, 1, 2, 4, 4, 8, true>;
using CfgK3M1   = ConvCfg<3, 1, 1, 1, 4, 4, 4, true>;
using CfgK3M2   = ConvCfg<3, 1, 1, 2, 4, 4, 4, true>;
using CfgK3S2M1 = ConvCfg<3, 2, 1, 1, 2, 4, 2, false>;
using CfgK3S2M2 = ConvCfg<3, 2, 1, 2, 2, 4, 2, false>;
using CfgK5M1   = ConvCfg<5, 1, 1, 1, 4, 4, 2, true, 2, 1>;
using CfgK5M2   = ConvCfg<5, 1, 1, 2, 4, 4, 2, true, 2, 1>;
using CfgK5D2M1 = ConvCfg<5, 1, 2, 1, 4, 4, 2, false, 2, 1>;
using CfgK5D2M2 = ConvCfg<5, 1, 2, 2, 4, 4, 2, false, 2, 1>;
using CfgK7M1   = ConvCfg<7, 1, 1, 1, 4, 4, 2, false, 2, 1>;
using CfgK7M2   = ConvCfg<7, 1, 1, 2, 4, 4, 2, false, 2, 1>;
// depth-1 layers (Conv2d of the BEV neck on [N,C,1,H,W] views): tile 1 x 8 x 32
//                        KS S  D  MI TD TH KC  DB   OCC KDG KSD
// r3: the BEV neck's images are small (128 x 192 down to 8 x 12 pixels): on 1 x 8 x 32 tiles with 64 output channels per
// workgroup a whole level is 2 .. 48 workgroups, each a serial chain of up to 1152 MFMAs per wave -- 85-100 us per layer
// whatever its size (profiles/r3/heads_kernel_stats_v1.csv).  Depth-1 layers therefore always run one 32-channel group
// per workgroup (MI = 1; the packing depends on MI and KC only), and on 1 x 4 x 32 tiles while that leaves the launch
// under two workgroups per CU: 4x the workgroups, a quarter of the chain.
// ... and with 16-32 input channels per chunk: a chunk costs one exposed global-memory round trip (~2-4 us on these
// launches: 36 MFMAs per wave cannot cover it), so a 64-channel layer took 8 of them at KC = 8 (40-60 us per layer,
// profiles/r3/heads_kernel_stats_v2.csv) and takes 4 at KC = 16.
using CfgP1M1s   = ConvCfg<1, 1, 1, 1, 1, 4, 32, true, 2, 0, 1>;
using CfgP1S2M1s = ConvCfg<1, 2, 1, 1, 1, 4, 16, true, 2, 0, 1>;
using CfgP3M1s   = ConvCfg<3, 1, 1, 1, 1, 4, 16, true, 2, 0, 1>;
using CfgP3S2M1s = ConvCfg<3, 2, 1, 1, 1, 4, 8, true, 2, 0, 1>;
// 3 (H) x 7 (W), stride 1: the sheared first convolution of the global model (sheared_conv.hip; desc.ksize_h = 3, ksize = 7)
using CfgP7M1s   = ConvCfg<7, 1, 1, 1, 1, 4, 8, true, 2, 0, 1, 3>;
using CfgWino   = WinoCfg<2, 4, 2>;          // k3/s1 fast path: 2 x 4 rows x 64 voxels, 2 input channels per chunk
using CfgWinoBig = WinoCfg<4, 4, 2>;        // LDS-DMA staged, two row pairs per wave: large layers
using CfgWino8  = WinoCfg<2, 4, 2, 2>;
using CfgWinoN  = WinoCfg<4, 4, 2, 4, 32>;   // 32-wide tile (4 rows x 8 quads per MFMA column block): narrow layers
using CfgWinoN8 = WinoCfg<4, 4, 2, 2, 32>;
using CfgWinoN3 = WinoCfg<4, 4, 2, 4, 32, 3>;
using CfgWinoP  = WinoCfg<1, 16, 2, 4, 32, 3, 1>;   // depth-1 (2D neck) layers: 1 x 16 x 32 tile, 3 taps per chunk, 20 KB of LDS   // the 32-wide tile LDS-DMA staged: <= 168 VGPRs, 3 workgroups per CU       // the same for rows that are only 8-byte aligned (W % 4 == 2)
using CfgWinoS2 = WinoS2Cfg<2>;
using CfgWinoS2v8 = WinoS2Cfg<2, 2>;   // output rows only 8-byte aligned
using CfgWinoK5 = WinoKCfg<5, 2>;
using CfgWinoK7 = WinoKCfg<7, 2>;
using CfgWinoK5D2 = WinoKCfg<5, 2, 2>;
using CfgDCM1   = DeconvCfg<1, 2, 4, 4>;
using CfgDCM2   = DeconvCfg<2, 2, 4, 4>;
using CfgDCM1v8 = DeconvCfg<1, 2, 4, 4, 2>;
using CfgDCM2v8 = DeconvCfg<2, 2, 4, 4, 2>;
// depth-1 transposed layer (nn.ConvTranspose2d(k3,s2,p1,op1) of the 2D neck on [N,C,1,H,W] views): 1 x 8 x 32 input tiles,
// only the two depth-parity-0 classes exist (their taps are the kd = 1 plane of the embedded 3x3x3 kernel)
using CfgDCP    = DeconvCfg<1, 1, 8, 4>;
using CfgDCPv8  = DeconvCfg<1, 1, 8, 4, 2>;

template <class Cfg>
constexpr Plan plan_of(int kind) { return Plan{Cfg::MI, Cfg::KC, Cfg::TD, Cfg::TH, 0, 0, 0, 0, 0, kind}; }

int make_plan(const snvc_conv3d_desc &d, Plan &p) {
    if (d.N < 0 || d.Cin <= 0 || d.Cout <= 0 || d.Din <= 0 || d.Hin <= 0 || d.Win <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d: sizes must be positive");
    const bool wide = d.Cout > 32;  // MI = 2 handles 64 output channels per workgroup
    if (d.transposed) {
        if (d.ksize != 3 || d.stride != 2 || d.pad != 1 || d.dilation != 1)
            return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d: transposed conv supports k3,s2,p1,op1 only");
        if (d.ksize_d == 1) {   // depth-1 form: [N,C,1,H,W] -> [N,C,1,2H,2W]; the weight is the 2D kernel on the kd = 1 plane of a 3x3x3 one
            if (d.Din != 1 || d.Dout != 1 || d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win || (d.ksize_h != 0 && d.ksize_h != 3))
                return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d: a depth-1 transposed layer maps [1,H,W] to [1,2H,2W]");
            p = plan_of<CfgDCP>(DCP_M1);
            p.tiles_d = 1; p.tiles_h = ceil_div(d.Hin, p.TH); p.tiles_w = ceil_div(d.Win, 32);
        } else {
        if (d.Dout != 2 * d.Din || d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win)
            return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d: transposed output must be 2x the input");
        // whole 32-channel groups run the single-group form (128 VGPRs: four workgroups per CU); the 64-channel
        // form only remains for channel counts that are not multiples of 32
        p = (wide && d.Cout % 32 != 0) ? plan_of<CfgDCM2>(DC_M2) : plan_of<CfgDCM1>(DC_M1);
        p.tiles_d = ceil_div(d.Din, p.TD); p.tiles_h = ceil_div(d.Hin, p.TH); p.tiles_w = ceil_div(d.Win, 32);
        }
    } else if (d.ksize_d == 1) {
        // depth-1 layer: nn.Conv2d(k, stride, padding=(k-1)/2) on an [N,C,1,H,W] view; the stride applies to H and W only
        // (k = 1, stride 2 is BasicBlock's downsample path, hrnet.py:56-69)
        if (d.Din != 1 || d.Dout != 1 || d.dilation != 1 || d.pad != (d.ksize - 1) / 2)
            return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d: ksize_d = 1 needs Din = Dout = 1, dilation 1, pad = (ksize-1)/2");
        const int kh_ = d.ksize_h ? d.ksize_h : d.ksize;      // kernel extent along H (its padding is (ksize_h-1)/2)
        if (kh_ != d.ksize && !(d.ksize == 7 && kh_ == 3 && d.stride == 1))
            return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d: ksize_h != ksize is built for the 3 x 7 depth-1 layer only");
        if (d.ksize == 7 && kh_ != 3)
            return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d: depth-1 ksize 7 is built as the 3 x 7 layer (ksize_h = 3)");
        const int eH = (d.Hin + 2 * ((kh_ - 1) / 2) - kh_) / d.stride + 1, eW = (d.Win + 2 * d.pad - d.ksize) / d.stride + 1;
        if (d.Hout != eH || d.Wout != eW)
            return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d: output size does not match the convolution arithmetic");
        const int key = d.ksize * 10 + d.stride;
        // one 32-channel group per workgroup on 1 x 4 x 32 tiles whatever the layer (see CfgP*s above)
        (void)wide;
        switch (key) {
            case 11: p = plan_of<CfgP1M1s>(P1_M1S); break;
            case 12: p = plan_of<CfgP1S2M1s>(P1S2_M1S); break;
            case 31: p = plan_of<CfgP3M1s>(P3_M1S); break;
            case 32: p = plan_of<CfgP3S2M1s>(P3S2_M1S); break;
            case 71: p = plan_of<CfgP7M1s>(P7_M1S); break;
            default: return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d: depth-1 layers are built for ksize 1 and 3 (stride 1 and 2) and 3 x 7 (stride 1)");
        }
        p.tiles_d = 1; p.tiles_h = ceil_div(d.Hout, p.TH); p.tiles_w = ceil_div(d.Wout, 32);
    } else {
        if (d.ksize_d != 0 && d.ksize_d != d.ksize)
            return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d: ksize_d must be 0 (cubic), ksize or 1");
        if (d.ksize_h != 0 && d.ksize_h != d.ksize)
            return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d: ksize_h != ksize needs a depth-1 layer (ksize_d = 1)");
        if (d.pad != d.dilation * (d.ksize - 1) / 2)
            return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d: pad must equal dilation*(ksize-1)/2");
        const int eff = d.dilation * (d.ksize - 1) + 1;
        const int eD = (d.Din + 2 * d.pad - eff) / d.stride + 1, eH = (d.Hin + 2 * d.pad - eff) / d.stride + 1,
                  eW = (d.Win + 2 * d.pad - eff) / d.stride + 1;
        if (d.Dout != eD || d.Hout != eH || d.Wout != eW)
            return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d: output size does not match the convolution arithmetic");
        const int key = d.ksize * 100 + d.stride * 10 + d.dilation;
        switch (key) {
            case 111: p = wide ? plan_of<CfgK1M2>(K1_M2) : plan_of<CfgK1M1>(K1_M1); break;
            case 311: p = wide ? plan_of<CfgK3M2>(K3_M2) : plan_of<CfgK3M1>(K3_M1); break;
            case 321: p = wide ? plan_of<CfgK3S2M2>(K3S2_M2) : plan_of<CfgK3S2M1>(K3S2_M1); break;
            case 511: p = wide ? plan_of<CfgK5M2>(K5_M2) : plan_of<CfgK5M1>(K5_M1); break;
            case 512: p = wide ? plan_of<CfgK5D2M2>(K5D2_M2) : plan_of<CfgK5D2M1>(K5D2_M1); break;
            case 711: p = wide ? plan_of<CfgK7M2>(K7_M2) : plan_of<CfgK7M1>(K7_M1); break;
            default:
                return fail(SNVC_ERR_UNSUPPORTED,
                            "snvc_conv3d: (ksize,stride,dilation) not in {(1,1,1),(3,1,1),(3,2,1),(5,1,1),(5,1,2),(7,1,1)}");
        }
        p.tiles_d = ceil_div(d.Dout, p.TD); p.tiles_h = ceil_div(d.Hout, p.TH); p.tiles_w = ceil_div(d.Wout, 32);
    }
    p.groups = ceil_div(d.Cout, 32 * p.MI);
    p.nchunks = ceil_div(d.Cin, p.KC);
    if (p.groups > 65535 || d.N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d: too many channel groups or samples");
    return SNVC_OK;
}

template <class Cfg, int EPI>
void launch_conv_variant(const ConvArgs &a, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};   // one bit per device: the attribute is per device
    if (!allow_large_lds(reinterpret_cast<const void *>(&conv3d_mfma_kernel<Cfg, EPI>), Cfg::LDS_BYTES, attr_done)) return;
    conv3d_mfma_kernel<Cfg, EPI><<<grid, 256, Cfg::LDS_BYTES, st>>>(a);
}

// FAST: also build the fast-epilogue variants of this configuration (the ones on a measured path)
template <class Cfg, bool FAST = false>
void launch_conv(const ConvArgs &a, dim3 grid, hipStream_t st) {
    if constexpr (FAST) {
        if (a.fast_epi && !a.plane) {
            if (a.res) launch_conv_variant<Cfg, 2>(a, grid, st);
            else launch_conv_variant<Cfg, 1>(a, grid, st);
            return;
        }
    }
    launch_conv_variant<Cfg, 0>(a, grid, st);
}

// depth-1 k3 / stride-1 layers big enough for the 1 x 16 x 32 Winograd tile (the small levels of the neck stay on the direct
// 1 x 4 x 32 form: a handful of workgroups either way)
// (every depth-1 k3 / stride-1 layer CARRIES the Winograd weights: what is packed depends on the layer, never on the extents
// of the probe it was packed with)
inline bool planar_wino_weights(const snvc_conv3d_desc &d) {
    return d.ksize_d == 1 && !d.transposed && d.ksize == 3 && (d.ksize_h == 0 || d.ksize_h == 3) && d.stride == 1 && d.dilation == 1;
}
inline bool planar_wino_layer(const snvc_conv3d_desc &d) { return planar_wino_weights(d) && d.Hout >= 32 && d.Wout >= 32; }

inline int64_t wino_packed_count(const snvc_conv3d_desc &d) {
    if (d.ksize_d == 1)
        return planar_wino_weights(d) ? (int64_t)ceil_div(d.Cout, 32) * ceil_div(d.Cin, 2) * CfgWinoP::WF : 0;
    if (!d.transposed && d.stride == 2 && d.ksize == 3 && d.dilation == 1)
        return (int64_t)ceil_div(d.Cout, 32) * ceil_div(d.Cin, 2) * 3 * CfgWinoS2::WF;
    if (d.transposed || d.stride != 1 || !(d.dilation == 1 || (d.dilation == 2 && d.ksize == 5))) return 0;
    const int64_t gc = (int64_t)ceil_div(d.Cout, 32) * ceil_div(d.Cin, 2);     // groups x chunks (KC = 2 everywhere)
    if (d.ksize == 3) return gc * CfgWino::WF;
    if (d.ksize == 5) return gc * 5 * CfgWinoK5::WF;
    if (d.ksize == 7) return gc * 7 * CfgWinoK7::WF;
    return 0;
}

template <class Cfg, bool RES>
void launch_winok_variant(const ConvArgs &a, dim3 grid, hipStream_t st) {
    constexpr int BYTES = Cfg::LDS_BYTES + 256;   // + (scale | bias) of 32 channels
    static std::atomic<unsigned> attr_done{0};   // one bit per device: the attribute is per device
    if (!allow_large_lds(reinterpret_cast<const void *>(&conv3d_winok_kernel<Cfg, RES>), BYTES, attr_done)) return;
    conv3d_winok_kernel<Cfg, RES><<<grid, 256, BYTES, st>>>(a);
}

template <class Cfg>
void launch_winos2_pipe(const ConvArgs &a, dim3 grid, hipStream_t st) {
    constexpr int BYTES = Cfg::LDS_BYTES + 256;   // + (scale | bias) of 32 channels
    if (a.res) {
        static std::atomic<unsigned> attr_done{0};
        if (!allow_large_lds(reinterpret_cast<const void *>(&conv3d_winos2_pipe_kernel<Cfg, true>), BYTES, attr_done)) return;
        conv3d_winos2_pipe_kernel<Cfg, true><<<grid, 256, BYTES, st>>>(a);
    } else {
        static std::atomic<unsigned> attr_done{0};
        if (!allow_large_lds(reinterpret_cast<const void *>(&conv3d_winos2_pipe_kernel<Cfg, false>), BYTES, attr_done)) return;
        conv3d_winos2_pipe_kernel<Cfg, false><<<grid, 256, BYTES, st>>>(a);
    }
}

template <class Cfg>
void launch_winok(const ConvArgs &a, dim3 grid, hipStream_t st) {
    if (a.res) launch_winok_variant<Cfg, true>(a, grid, st);
    else launch_winok_variant<Cfg, false>(a, grid, st);
}

template <class Cfg>
void launch_winos2_pipe_stats(const ConvArgs &a, dim3 grid, hipStream_t st) {
    constexpr int BYTES = Cfg::LDS_BYTES + 256;
    static std::atomic<unsigned> attr_done{0};
    if (!allow_large_lds(reinterpret_cast<const void *>(&conv3d_winos2_pipe_kernel<Cfg, false, 3>), BYTES, attr_done)) return;
    conv3d_winos2_pipe_kernel<Cfg, false, 3><<<grid, 256, BYTES, st>>>(a);
}

template <class Cfg, bool RES, bool PLANE>
void launch_wino_variant(const ConvArgs &a, dim3 grid, hipStream_t st) {
    constexpr int BYTES = Cfg::LDS_BYTES + 256;   // + (scale | bias) of 32 channels
    static std::atomic<unsigned> attr_done{0};   // one bit per device: the attribute is per device
    if (!allow_large_lds(reinterpret_cast<const void *>(&conv3d_wino_kernel<Cfg, RES, PLANE>), BYTES, attr_done)) return;
    conv3d_wino_kernel<Cfg, RES, PLANE><<<grid, 256, BYTES, st>>>(a);
}

template <class Cfg, bool RES, bool PLANE, int XMODE = 0>
void launch_wino_dma_variant(const ConvArgs &a, dim3 grid, hipStream_t st) {
    constexpr int BYTES = Cfg::LDS_BYTES + (XMODE == 1 ? 384 : 256);   // + (scale | bias [| head weights]) of 32 channels
    static std::atomic<unsigned> attr_done{0};   // one bit per device: the attribute is per device
    if (!allow_large_lds(reinterpret_cast<const void *>(&conv3d_wino_dma_kernel<Cfg, RES, PLANE, XMODE>), BYTES, attr_done)) return;
    conv3d_wino_dma_kernel<Cfg, RES, PLANE, XMODE><<<grid, 256, BYTES, st>>>(a);
}

template <class Cfg>
void launch_wino_dma(const ConvArgs &a, dim3 grid, hipStream_t st) {
    if (a.res && a.plane) launch_wino_dma_variant<Cfg, true, true>(a, grid, st);
    else if (a.res) launch_wino_dma_variant<Cfg, true, false>(a, grid, st);
    else if (a.plane) launch_wino_dma_variant<Cfg, false, true>(a, grid, st);
    else launch_wino_dma_variant<Cfg, false, false>(a, grid, st);
}

template <class Cfg>
void launch_wino(const ConvArgs &a, dim3 grid, hipStream_t st) {
    if (a.res && a.plane) launch_wino_variant<Cfg, true, true>(a, grid, st);
    else if (a.res) launch_wino_variant<Cfg, true, false>(a, grid, st);
    else if (a.plane) launch_wino_variant<Cfg, false, true>(a, grid, st);
    else launch_wino_variant<Cfg, false, false>(a, grid, st);
}

template <class Cfg, int EPI>
void launch_deconv_variant(const ConvArgs &a, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};   // one bit per device: the attribute is per device
    if (!allow_large_lds(reinterpret_cast<const void *>(&deconv3d_mfma_kernel<Cfg, EPI>), Cfg::LDS_BYTES, attr_done)) return;
    deconv3d_mfma_kernel<Cfg, EPI><<<grid, 256, Cfg::LDS_BYTES, st>>>(a);
}

template <class Cfg>
void launch_deconv(const ConvArgs &a, dim3 grid, hipStream_t st) {
    // the pair exchange of the fast epilogue needs an even input width (both lanes of a pair in range)
    if (a.stats) {    // validated by conv3d_forward_impl: fast epilogue conditions, one channel group per workgroup, no addends
        if constexpr (Cfg::MI == 1) launch_deconv_variant<Cfg, 5>(a, grid, st);
    } else if (a.head_w) {   // validated by snvc_conv3d_forward_head: fast epilogue conditions hold, one channel group
        if constexpr (Cfg::MI == 1) {
            if (a.res) launch_deconv_variant<Cfg, 4>(a, grid, st);
            else launch_deconv_variant<Cfg, 3>(a, grid, st);
        }
    } else if (a.fast_epi && a.Win % 2 == 0) {
        if (a.res) launch_deconv_variant<Cfg, 2>(a, grid, st);
        else launch_deconv_variant<Cfg, 1>(a, grid, st);
    } else {
        launch_deconv_variant<Cfg, 0>(a, grid, st);
    }
}

}  // namespace
}  // namespace snvc

namespace snvc {
int conv3d_forward_impl(const snvc_conv3d_desc *d, const float *x, const float *packed_weight, const float *scale,
                        const float *bias, const float *residual, const float *depth_planes, float *y,
                        const float *head_w, float *y_head, void *stream, double *stats = nullptr);

namespace {
// stats[((n * tiles + t) * groups + cg)][32][2] -> partial[(n * C + c)][2]: one workgroup per (n, c), fixed-order tree
__global__ void __launch_bounds__(256)
conv_stats_fold_kernel(const double *__restrict__ stats, double *__restrict__ partial, int C, int groups, int tiles) {
    const int c = blockIdx.x, cg = c >> 5, ch = c & 31;
    const int64_t n = blockIdx.y;
    double s0 = 0.0, s1 = 0.0;
    for (int t = threadIdx.x; t < tiles; t += 256) {
        const double *p = stats + (((n * tiles + t) * groups + cg) * 32 + ch) * 2;
        s0 += p[0];
        s1 += p[1];
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s0;
    sh[1][threadIdx.x] = s1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { sh[0][threadIdx.x] += sh[0][threadIdx.x + w]; sh[1][threadIdx.x] += sh[1][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[(n * C + c) * 2] = sh[0][0];
        partial[(n * C + c) * 2 + 1] = sh[1][0];
    }
}
// the Winograd tiling of the layers the statistics epilogue is built for (the dispatcher's default forms)
inline int64_t stats_tiles(const snvc_conv3d_desc &d) {
    if (d.transposed)   // four parity classes of 2 x 4 x 32 INPUT tiles (DeconvCfg<1, 2, 4, 4>), one workgroup slot each
        return 4 * (int64_t)ceil_div(d.Din, 2) * ceil_div(d.Hin, 4) * ceil_div(d.Win, 32);
    return (int64_t)ceil_div(d.Dout, 4) * ceil_div(d.Hout, 4) * ceil_div(d.Wout, 32);
}
}  // namespace

namespace {
// The same fold for MANY slots (r6: the split kernels' statistics epilogue leaves one slot per wave, 46 k at cfg4's transposed layer; one
// workgroup per channel walking them 512 bytes apart took 106 us): a first round of 64 -> 1 with consecutive threads on consecutive
// doubles.  Row r' = sum of rows 64 r' .. 64 r' + 63 in ascending order (fixed order: deterministic), rows of `cols` doubles.
__global__ void __launch_bounds__(256)
conv_stats_fold_rows_kernel(const double *__restrict__ in, double *__restrict__ out, int64_t rows, int cols, int64_t in_ns, int64_t out_ns) {
    const int64_t n = blockIdx.z;
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    if (col >= cols) return;
    const double *p = in + n * in_ns + r0 * cols + col;
    const int64_t cnt = rows - r0 < 64 ? rows - r0 : 64;
    double s = 0.0;
#pragma unroll 8
    for (int64_t r = 0; r < cnt; ++r) s += p[r * cols];
    out[n * out_ns + (int64_t)blockIdx.y * cols + col] = s;
}
}  // namespace

// `scratch` (N * ceil(tiles / 64) * groups * 64 doubles, conv_stats_fold_scratch_doubles) is used when tiles > 4096.
int64_t conv_stats_fold_scratch_doubles(int64_t N, int groups, int64_t tiles) {
    return tiles > 4096 ? N * ((tiles + 63) / 64) * groups * 64 : 0;
}

void launch_conv_stats_fold(const double *stats, double *scratch, double *partial, int64_t N, int C, int groups, int64_t tiles, hipStream_t st) {
    const int cols = groups * 64;
    if (tiles > 4096 && scratch) {
        const int64_t out_rows = (tiles + 63) / 64;
        conv_stats_fold_rows_kernel<<<dim3((unsigned)((cols + 255) / 256), (unsigned)out_rows, (unsigned)N), 256, 0, st>>>(
            stats, scratch, tiles, cols, tiles * cols, out_rows * cols);
        stats = scratch;
        tiles = out_rows;
    }
    conv_stats_fold_kernel<<<dim3((unsigned)C, (unsigned)N), 256, 0, st>>>(stats, partial, C, groups, (int)tiles);
}
}

extern "C" {

int64_t snvc_conv3d_packed_weight_count(const snvc_conv3d_desc *d) {
    using namespace snvc;
    Plan p;
    if (!d || make_plan(*d, p) != SNVC_OK) return -1;
    const bool planar = d->ksize_d == 1;      // depth-1 layer: only the direct packing, no special forms
    const int64_t taps = d->transposed ? 27 : (int64_t)(planar ? 1 : d->ksize) * (planar && d->ksize_h ? d->ksize_h : d->ksize) * d->ksize;
    int64_t count = (int64_t)p.groups * p.nchunks * taps * (p.KC / 2) * 64 * p.MI;
    if (planar) return count + wino_packed_count(*d);
    count += wino_packed_count(*d);   // k3/s1 layers also carry the Winograd-transformed weights
    // 1x1x1 layers with <= 2 output channels also keep their raw [Cout][Cin] weights (streaming kernel)
    if (!d->transposed && d->ksize == 1 && (d->Cout <= 2 || d->Cin <= 2)) count += (int64_t)d->Cout * d->Cin;      // r6: or FROM <= 2 channels
    // 3x3x3 / stride-1 layers with ONE output channel too ([Cin][27], VALU kernel)
    if (!d->transposed && d->ksize == 3 && d->stride == 1 && d->dilation == 1 && d->Cout == 1) count += (int64_t)d->Cin * 27;
    // transposed layers to ONE channel as well ([Cin][1][27], VALU kernel of conv3d_small.hip)
    if (d->transposed && d->Cout == 1) count += (int64_t)d->Cin * 27;
    return count;
}

int snvc_conv3d_pack_weights(const snvc_conv3d_desc *d, const float *weight, float *packed, void *stream) {
    using namespace snvc;
    Plan p;
    if (!d) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_pack_weights: null desc");
    int rc = make_plan(*d, p);
    if (rc) return rc;
    if (!weight || !packed) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_pack_weights: null pointer");
    int64_t total = snvc_conv3d_packed_weight_count(d);
    const bool planar = d->ksize_d == 1;
    const bool k3c1 = !planar && ((!d->transposed && d->ksize == 3 && d->stride == 1 && d->dilation == 1 && d->Cout == 1) ||
                                  (d->transposed && d->Cout == 1));
    if (!planar && ((!d->transposed && d->ksize == 1 && (d->Cout <= 2 || d->Cin <= 2)) || k3c1)) {   // raw copy behind the MFMA packing
        const int64_t nraw = k3c1 ? (int64_t)d->Cin * 27 : (int64_t)d->Cout * d->Cin;
        total -= nraw;
        if (hipMemcpyAsync(packed + total, weight, sizeof(float) * nraw, hipMemcpyDeviceToDevice,
                           as_stream(stream)) != hipSuccess)
            return fail(SNVC_ERR_HIP, "snvc_conv3d_pack_weights: hipMemcpyAsync failed");
    }
    const int64_t wino = wino_packed_count(*d);
    if (wino) {
        total -= wino;
        const unsigned wb = (unsigned)ceil_div<int64_t>(wino, 256);
        const int nck = ceil_div(d->Cin, 2);
        if (planar)
            pack_wino_weights_kernel<<<wb, 256, 0, as_stream(stream)>>>(weight, packed + total, d->Cout, d->Cin, 2, nck, wino, 3);
        else if (d->ksize == 3 && d->stride == 2)
            pack_winos2_weights_kernel<<<wb, 256, 0, as_stream(stream)>>>(weight, packed + total, d->Cout, d->Cin, 2, nck, wino);
        else if (d->ksize == 3)
            pack_wino_weights_kernel<<<wb, 256, 0, as_stream(stream)>>>(weight, packed + total, d->Cout, d->Cin, 2, nck, wino);
        else if (d->ksize == 5)
            pack_winok_weights_kernel<5><<<wb, 256, 0, as_stream(stream)>>>(weight, packed + total, d->Cout, d->Cin, 2, nck, wino);
        else
            pack_winok_weights_kernel<7><<<wb, 256, 0, as_stream(stream)>>>(weight, packed + total, d->Cout, d->Cin, 2, nck, wino);
        int rcw = check_launch("snvc_conv3d_pack_weights(winograd)");
        if (rcw) return rcw;
    }
    const unsigned blocks = (unsigned)ceil_div<int64_t>(total, 256);
    if (d->transposed)
        pack_deconv_weights_kernel<<<blocks, 256, 0, as_stream(stream)>>>(weight, packed, d->Cout, d->Cin, p.MI, p.KC,
                                                                           p.nchunks, total);
    else
        pack_conv_weights_kernel<<<blocks, 256, 0, as_stream(stream)>>>(weight, packed, d->Cout, d->Cin,
                                                                         (planar ? 1 : d->ksize) * (planar && d->ksize_h ? d->ksize_h : d->ksize) * d->ksize, p.MI, p.KC,
                                                                         p.nchunks, total);
    return check_launch("snvc_conv3d_pack_weights");
}

int snvc_conv3d_forward(const snvc_conv3d_desc *d, const float *x, const float *packed_weight,
                        const float *scale, const float *bias, const float *residual, float *y,
                        void *stream) {
    return snvc_conv3d_forward_ex(d, x, packed_weight, scale, bias, residual, nullptr, y, stream);
}

int snvc_conv3d_forward_ex(const snvc_conv3d_desc *d, const float *x, const float *packed_weight,
                           const float *scale, const float *bias, const float *residual,
                           const float *depth_planes, float *y, void *stream) {
    return snvc::conv3d_forward_impl(d, x, packed_weight, scale, bias, residual, depth_planes, y, nullptr, nullptr, stream);
}

int snvc_conv3d_forward_head(const snvc_conv3d_desc *d, const float *x, const float *packed_weight,
                             const float *scale, const float *bias, const float *residual,
                             const float *head_weight, float *y_head, void *stream) {
    using namespace snvc;
    if (!d || !head_weight || !y_head) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_forward_head: null pointer");
    if (!d->transposed || d->Cout != 32 || d->Win % 2 != 0 || (d->flags & SNVC_EPI_SIGMOID) ||
        (int64_t)d->Dout * d->Hout * d->Wout >= ((int64_t)1 << 27) || (reinterpret_cast<uintptr_t>(y_head) & 7) ||
        (reinterpret_cast<uintptr_t>(residual) & 15) || (d->res_batch_stride % 4) != 0)
        return fail(SNVC_ERR_UNSUPPORTED,
                    "snvc_conv3d_forward_head: needs a transposed layer with 32 output channels, an even input width, "
                    "no Sigmoid, a 16-byte aligned residual and an 8-byte aligned head output");
    return conv3d_forward_impl(d, x, packed_weight, scale, bias, residual, nullptr, nullptr, head_weight, y_head, stream);
}

int snvc_conv3d_forward_side_head(const snvc_conv3d_desc *d, const float *x, const float *packed_weight,
                                  const float *scale, const float *bias, const float *residual, float *y,
                                  const float *head_weight, float *y_head, void *stream) {
    using namespace snvc;
    if (!d || !y || !head_weight || !y_head) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_forward_side_head: null pointer");
    return conv3d_forward_impl(d, x, packed_weight, scale, bias, residual, nullptr, y, head_weight, y_head, stream);
}

int64_t snvc_conv3d_stats_workspace_bytes(const snvc_conv3d_desc *d) {
    using namespace snvc;
    if (!d || d->N < 0 || d->Cout <= 0 || d->Dout <= 0 || d->Hout <= 0 || d->Wout <= 0) return -1;
    const int64_t groups = ceil_div(d->Cout, 32);
    return (d->N * stats_tiles(*d) * groups * 64 + d->N * (int64_t)d->Cout * 2) * (int64_t)sizeof(double);
}

int snvc_conv3d_forward_stats(const snvc_conv3d_desc *d, const float *x, const float *packed_weight, float *y, const float *gamma,
                              const float *beta, float *scale, float *shift, float *mean, float *var, void *workspace, float eps,
                              void *stream) {
    using namespace snvc;
    if (!d || !y || !scale || !shift || !workspace) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_forward_stats: null pointer");
    if (d->N <= 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_forward_stats: empty batch");
    if (d->flags || d->ksize_d == 1 || d->ksize != 3 || d->dilation != 1 || (d->stride != 1 && d->stride != 2) || d->Cout % 32 != 0 ||
        (d->transposed && d->stride != 2))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward_stats: built for 3x3x3 Conv3d layers (stride 1 or 2) and "
                                          "ConvTranspose3d(k3,s2,p1,op1) with whole 32-channel groups and no epilogue");
    double *stats = static_cast<double *>(workspace);
    int rc = conv3d_forward_impl(d, x, packed_weight, nullptr, nullptr, nullptr, nullptr, y, nullptr, nullptr, stream, stats);
    if (rc) return rc;
    const int groups = d->Cout / 32;
    const int64_t tiles = stats_tiles(*d);
    double *partial = stats + d->N * tiles * groups * 64;
    conv_stats_fold_kernel<<<dim3((unsigned)d->Cout, (unsigned)d->N), 256, 0, as_stream(stream)>>>(stats, partial, d->Cout, groups, (int)tiles);
    rc = check_launch("snvc_conv3d_forward_stats(fold)");
    if (rc) return rc;
    launch_norm_finalize(partial, gamma, beta, scale, shift, mean, var, d->N, d->Cout, (int64_t)d->Dout * d->Hout * d->Wout, 1, eps,
                         as_stream(stream));
    return check_launch("snvc_conv3d_forward_stats(finalize)");
}

}  // extern "C"

namespace snvc {

int conv3d_forward_impl(const snvc_conv3d_desc *d, const float *x, const float *packed_weight, const float *scale,
                        const float *bias, const float *residual, const float *depth_planes, float *y,
                        const float *head_w, float *y_head, void *stream, double *stats) {
    Plan p;
    if (!d) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_forward: null desc");
    int rc = make_plan(*d, p);
    if (rc) return rc;
    if (d->N == 0) return SNVC_OK;
    if (!x || !packed_weight || (!y && !head_w)) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_forward: null pointer");
    if ((scale == nullptr) != (bias == nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_forward: scale and bias must both be given or both be NULL");
    if ((d->flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST)) && !residual)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_forward: residual flag without residual pointer");
    if ((d->flags & SNVC_EPI_ADD_PRE) && (d->flags & SNVC_EPI_ADD_POST))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_forward: ADD_PRE and ADD_POST are exclusive");
    if (depth_planes && (d->transposed || d->stride != 1 || d->Dout < 2 || d->ksize_d == 1 || (d->ksize == 1 && d->Cout <= 2)))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward_ex: depth planes need a stride-1 Conv3d with Dout >= 2");
    if (d->transposed && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 7))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_conv3d_forward: transposed y / residual must be 8-byte aligned");
    const int64_t in_sz = (int64_t)d->Cin * d->Din * d->Hin * d->Win, out_sz = (int64_t)d->Cout * d->Dout * d->Hout * d->Wout;
    if (in_sz + (int64_t)8 * d->Din * d->Hin * d->Win >= ((int64_t)1 << 31))  // 32-bit in-sample offsets
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward: one sample must stay below 2^31 elements");

    ConvArgs a;
    a.x = x; a.wp = packed_weight; a.scale = scale; a.bias = bias;
    a.res = (d->flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST)) ? residual : nullptr;
    a.plane = depth_planes;
    a.head_w = head_w; a.y_head = y_head;
    a.stats = stats;
    a.dc_planar = (d->transposed && d->ksize_d == 1) ? 1 : 0;
    const bool side_head = head_w && y;   // snvc_conv3d_forward_side_head: y AND its one-channel projection
    const bool pooled = (d->flags & SNVC_EPI_AVGPOOL_D4) != 0;   // y is [N,Cout,Dout/4,Hout,Wout]
    if (pooled && (side_head || head_w || d->transposed || d->ksize_d == 1 || d->ksize != 3 || d->stride != 1 || d->dilation != 1 ||
                   depth_planes || d->Dout % 4 != 0 || d->Cout % 32 != 0 || (d->flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST | SNVC_EPI_SIGMOID)) ||
                   (d->algo & SNVC_ALGO_ARITH_MASK) == SNVC_ALGO_DIRECT))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward: SNVC_EPI_AVGPOOL_D4 is built for 3x3x3 / stride-1 Conv3d layers "
                                          "with Dout % 4 == 0, whole 32-channel groups and no residual");
    if (side_head && (d->transposed || d->ksize_d == 1 || d->ksize != 3 || d->stride != 1 || d->dilation != 1 || depth_planes))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward_side_head: built for 3x3x3 / stride-1 Conv3d layers");
    if (head_w && !side_head && p.kind != DC_M1)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward_head: layer is not a single-group transposed convolution");
    a.wp_wino = nullptr; a.nchunks_wino = 0;
    a.y = y;
    a.Cin = d->Cin; a.Din = d->Din; a.Hin = d->Hin; a.Win = d->Win;
    a.Cout = d->Cout; a.Dout = d->Dout; a.Hout = d->Hout; a.Wout = d->Wout;
    a.tiles_d = p.tiles_d; a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w;
    a.nchunks = p.nchunks; a.flags = d->flags;
    a.x_bs = d->x_batch_stride ? d->x_batch_stride : in_sz;
    a.y_bs = d->y_batch_stride ? d->y_batch_stride : (pooled ? out_sz / 4 : out_sz);
    a.r_bs = d->res_batch_stride ? d->res_batch_stride : out_sz;
    a.vec = (d->Win % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (a.x_bs % 4 == 0);
    const bool vec8 = (d->Win % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % 8 == 0) && (a.x_bs % 2 == 0);
    // fast epilogues: (wave-uniform channel base) + 32-bit lane byte offsets, whole 32-channel groups, no
    // Sigmoid; 16-byte stores when the output rows allow them
    const bool direct_only = (d->algo & SNVC_ALGO_ARITH_MASK) == SNVC_ALGO_DIRECT;
    const bool fast_common = (int64_t)d->Dout * d->Hout * d->Wout < ((int64_t)1 << 27) && d->Cout % 32 == 0 &&
                             !(d->flags & SNVC_EPI_SIGMOID) && !(d->algo & SNVC_ALGO_GENERIC_EPILOGUE);
    const bool epi16 = d->Wout % 4 == 0 && a.y_bs % 4 == 0 && a.r_bs % 4 == 0 &&
                       ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(a.res) |
                         reinterpret_cast<uintptr_t>(depth_planes)) & 15) == 0;
    a.fast_epi = fast_common && epi16;

    // 1x1x1 convolution to <= 2 channels: HBM-bound streaming kernel (raw weights ride at the end of
    // the packed buffer, see snvc_conv3d_pack_weights)
    const bool planar = d->ksize_d == 1;
    const int64_t S = (int64_t)d->Dout * d->Hout * d->Wout;
    if (!planar && !d->transposed && d->ksize == 1 && d->stride == 1 && d->Cout <= 2 && (S % 4) == 0 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(a.res)) & 15) == 0 &&
        a.x_bs % 4 == 0 && a.y_bs % 4 == 0 && a.r_bs % 4 == 0) {
        const float *wraw = packed_weight + snvc_conv3d_packed_weight_count(d) - (int64_t)d->Cout * d->Cin;
        int64_t blocks = ceil_div<int64_t>(S / 4, 256);
        if (blocks > 4096) blocks = 4096;
        dim3 g((unsigned)blocks, (unsigned)d->N);
        if (d->Cout == 1)
            pointwise_small_kernel<1><<<g, 256, 0, as_stream(stream)>>>(x, wraw, scale, bias, a.res, y, d->Cin, S, a.x_bs, a.y_bs, a.r_bs, d->flags);
        else
            pointwise_small_kernel<2><<<g, 256, 0, as_stream(stream)>>>(x, wraw, scale, bias, a.res, y, d->Cin, S, a.x_bs, a.y_bs, a.r_bs, d->flags);
        return check_launch("snvc_conv3d_forward(pointwise)");
    }
    // 1x1x1 from <= 2 channels (the classifier's data gradient): streamed as well
    if (!planar && !d->transposed && d->ksize == 1 && d->stride == 1 && d->Cin <= 2 && d->Cout > 2 && (S % 4) == 0 && !depth_planes &&
        !head_w && !stats && (d->algo & SNVC_ALGO_ARITH_MASK) == 0 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(a.res)) & 15) == 0 &&
        a.x_bs % 4 == 0 && a.y_bs % 4 == 0 && a.r_bs % 4 == 0) {
        const float *wraw = packed_weight + snvc_conv3d_packed_weight_count(d) - (int64_t)d->Cout * d->Cin;
        int64_t blocks = ceil_div<int64_t>(S / 4, 256);
        if (blocks > 4096) blocks = 4096;
        dim3 g((unsigned)blocks, (unsigned)d->N);
        if (d->Cin == 1)
            pointwise_expand_kernel<1><<<g, 256, 0, as_stream(stream)>>>(x, wraw, scale, bias, a.res, y, d->Cout, S, a.x_bs, a.y_bs, a.r_bs, d->flags);
        else
            pointwise_expand_kernel<2><<<g, 256, 0, as_stream(stream)>>>(x, wraw, scale, bias, a.res, y, d->Cout, S, a.x_bs, a.y_bs, a.r_bs, d->flags);
        return check_launch("snvc_conv3d_forward(pointwise expand)");
    }
    // 3x3x3 / stride 1 to ONE channel: VALU kernel (raw weights ride at the end of the packed buffer)
    if (!planar && !d->transposed && d->ksize == 3 && d->stride == 1 && d->dilation == 1 && d->Cout == 1 && !depth_planes) {
        const float *wraw = packed_weight + snvc_conv3d_packed_weight_count(d) - (int64_t)d->Cin * 27;
        const int th_ = ceil_div(d->Hout, K3C1_TH), tw_ = ceil_div(d->Wout, 32);
        const int64_t nt = (int64_t)ceil_div(d->Dout, K3C1_TD) * th_ * tw_;
        if (nt < ((int64_t)1 << 31) && d->N <= 65535) {
            conv3d_k3_cout1_kernel<<<dim3((unsigned)nt, (unsigned)d->N), 256, 0, as_stream(stream)>>>(
                x, wraw, scale, bias, a.res, y, d->Cin, d->Dout, d->Hout, d->Wout, th_, tw_, a.x_bs, a.y_bs, a.r_bs, d->flags);
            return check_launch("snvc_conv3d_forward(k3 to one channel)");
        }
    }
    // transposed layer to ONE channel (the folded hourglass tail + classifier): VALU kernel, raw weights as above
    if (!planar && !head_w && deconv3d_cout1_qualifies(*d, x, y, a.res, a.x_bs, a.y_bs, a.r_bs)) {
        const float *wraw = packed_weight + snvc_conv3d_packed_weight_count(d) - (int64_t)d->Cin * 27;
        deconv3d_cout1_launch(*d, x, wraw, scale, bias, a.res, y, a.x_bs, a.y_bs, a.r_bs, as_stream(stream));
        return check_launch("snvc_conv3d_forward(transposed to one channel)");
    }
    // k3 / stride 2: polyphase + F(4,2) along W (LDS-DMA staged: 16-byte INPUT rows; output rows may be 8-byte ones)
    {
        const bool out8 = fast_common && (d->Wout % 2 == 0) && a.y_bs % 2 == 0 && a.r_bs % 2 == 0 &&
                          ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(a.res)) & 7) == 0;
        if (!planar && !d->transposed && d->ksize == 3 && d->stride == 2 && d->dilation == 1 && a.vec && (a.fast_epi || out8) &&
            !depth_planes && !direct_only) {
            a.wp_wino = packed_weight + (int64_t)p.groups * p.nchunks * 27 * (p.KC / 2) * 64 * p.MI;
            a.nchunks_wino = ceil_div(d->Cin, 2);
            a.groups = ceil_div(d->Cout, 32);
            a.tiles_d = ceil_div(d->Dout, 4); a.tiles_h = ceil_div(d->Hout, 4); a.tiles_w = ceil_div(d->Wout, 32);
            const int64_t nj = (int64_t)a.tiles_d * a.tiles_h * a.tiles_w * a.groups * d->N;
            if (nj < ((int64_t)1 << 31)) {
                a.njobs = (int)nj;
                // default: the slice-pipelined refill; SNVC_ALGO_WINO_TILE_STD selects the per-chunk refill form
                const bool per_chunk = (d->algo & SNVC_ALGO_WINO_TILE_MASK) == SNVC_ALGO_WINO_TILE_STD;
                if (stats) {        // the default form with 16-byte output rows carries the statistics epilogue
                    if (per_chunk || !a.fast_epi || a.res)
                        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward_stats: this stride-2 layer does not take the default kernel form");
                    launch_winos2_pipe_stats<WinoS2PipeCfg<4>>(a, dim3((unsigned)nj, 1, 1), as_stream(stream));
                    return check_launch("snvc_conv3d_forward_stats(winograd stride 2)");
                }
                if (per_chunk && a.fast_epi) launch_winok<CfgWinoS2>(a, dim3((unsigned)nj, 1, 1), as_stream(stream));
                else if (per_chunk) launch_winok<CfgWinoS2v8>(a, dim3((unsigned)nj, 1, 1), as_stream(stream));
                else if (a.fast_epi) launch_winos2_pipe<WinoS2PipeCfg<4>>(a, dim3((unsigned)nj, 1, 1), as_stream(stream));
                else launch_winos2_pipe<WinoS2PipeCfg<2>>(a, dim3((unsigned)nj, 1, 1), as_stream(stream));
                return check_launch("snvc_conv3d_forward(winograd stride 2)");
            }
        }
    }
    // k5 / k7, stride 1, no dilation: Winograd F(4,KS) along W (LDS-DMA staged: 16-byte rows only)
    if (!planar && !d->transposed && (d->ksize == 5 || d->ksize == 7) && d->stride == 1 &&
        (d->dilation == 1 || (d->dilation == 2 && d->ksize == 5)) && a.vec && a.fast_epi && !depth_planes) {
        if (!direct_only) {
            const int64_t taps = (int64_t)d->ksize * d->ksize * d->ksize;
            a.wp_wino = packed_weight + (int64_t)p.groups * p.nchunks * taps * (p.KC / 2) * 64 * p.MI;
            a.nchunks_wino = ceil_div(d->Cin, 2);
            a.groups = ceil_div(d->Cout, 32);
            a.tiles_d = ceil_div(d->Dout, 4); a.tiles_h = ceil_div(d->Hout, 4); a.tiles_w = ceil_div(d->Wout, 32);
            const int64_t nj = (int64_t)a.tiles_d * a.tiles_h * a.tiles_w * a.groups * d->N;
            if (nj < ((int64_t)1 << 31)) {
                a.njobs = (int)nj;
                if (d->ksize == 5 && d->dilation == 2) launch_winok<CfgWinoK5D2>(a, dim3((unsigned)nj, 1, 1), as_stream(stream));
                else if (d->ksize == 5) launch_winok<CfgWinoK5>(a, dim3((unsigned)nj, 1, 1), as_stream(stream));
                else launch_winok<CfgWinoK7>(a, dim3((unsigned)nj, 1, 1), as_stream(stream));
                return check_launch("snvc_conv3d_forward(winograd k5/k7)");
            }
        }
    }
    // depth-1 k3 / stride 1 (the 2D neck's larger levels) on 16-byte rows: the same Winograd kernel on 1 x 16 x 32 tiles with
    // 3 taps per chunk instead of 9 (half the MFMAs of the direct depth-1 form)
    if (planar && planar_wino_layer(*d) && a.vec && epi16 && fast_common && !direct_only && !depth_planes && !head_w && !pooled && !stats) {
        const int th_ = ceil_div(d->Hout, CfgWinoP::TH), tw_ = ceil_div(d->Wout, CfgWinoP::TW), groups_ = ceil_div(d->Cout, 32);
        const int64_t nj = (int64_t)th_ * tw_ * groups_ * d->N;
        // a job is a serial chain of Cin / 2 chunks: below two jobs per CU the direct form's 4x smaller tiles (4x the
        // workgroups) finish sooner (2 crops of the released shape: heads 0.46 vs 0.44 ms/crop; 8 crops: 0.164 vs 0.196)
        const bool forced = (d->algo & SNVC_ALGO_WINO_TILE_MASK) == SNVC_ALGO_WINO_TILE_BIG;     // tests reach the form on small inputs
        if ((forced || nj >= 2 * device_cu_count()) && nj < ((int64_t)1 << 31)) {
            a.wp_wino = packed_weight + (int64_t)p.groups * p.nchunks * 9 * (p.KC / 2) * 64 * p.MI;
            a.nchunks_wino = ceil_div(d->Cin, CfgWinoP::KC);
            a.groups = groups_;
            a.tiles_d = 1; a.tiles_h = th_; a.tiles_w = tw_;
            a.njobs = (int)nj;
            launch_wino_dma<CfgWinoP>(a, dim3((unsigned)nj, 1, 1), as_stream(stream));
            return check_launch("snvc_conv3d_forward(depth-1 winograd)");
        }
    }
    // k3 / stride 1: Winograd F(4,3) along W when the rows allow 8-byte pair stores and 16-byte staging
    {
        const int64_t wino = (!planar && d->ksize == 3 && d->stride == 1 && !d->transposed) ? wino_packed_count(*d) : 0;
        const bool pair_ok = (d->Wout % 2 == 0) && a.y_bs % 2 == 0 && a.r_bs % 2 == 0 &&
                             ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(a.res) |
                               reinterpret_cast<uintptr_t>(depth_planes)) & 7) == 0;
        const bool wide = a.vec && epi16;   // 16-byte staging and stores; else 8-byte ones (pair_ok)
        if (wino && pair_ok && fast_common && (wide || vec8) && !direct_only) {
            a.wp_wino = packed_weight + (int64_t)p.groups * p.nchunks * 27 * (p.KC / 2) * 64 * p.MI;
            a.nchunks_wino = ceil_div(d->Cin, CfgWino::KC);
            a.groups = ceil_div(d->Cout, 32);
            // tile choice.  Default: the 4x4x32 tile, LDS-DMA staged (137 VGPRs, 51 KB LDS: three workgroups per
            // CU), or its register-staged 8-byte-row form when the rows are not 16-byte aligned.  Measured on cfg2:
            // conv1 2.70 ms / conv2 1.40 ms / hg conv2 0.73 ms, against 2.82 / 1.45 / 0.89 for the 4x4x64 LDS-DMA
            // tile (BIG) and 2.93 / 1.52 / 0.91 for the 2x4x64 register-staged one (STD).  desc.algo's
            // SNVC_ALGO_WINO_TILE_* bits select the other forms (the parity tests run all of them).
            const int tsel = d->algo & SNVC_ALGO_WINO_TILE_MASK;
            const bool big = tsel == SNVC_ALGO_WINO_TILE_BIG && wide, stdt = tsel == SNVC_ALGO_WINO_TILE_STD,
                       nreg = tsel == SNVC_ALGO_WINO_TILE_NARROW_REG;
            const bool narrow = !big && !stdt;
            const int TDc = (big || narrow) ? 4 : 2, THc = 4, TWc = narrow ? 32 : 64;
            a.tiles_d = ceil_div(d->Dout, TDc);
            a.tiles_h = ceil_div(d->Hout, THc);
            a.tiles_w = ceil_div(d->Wout, TWc);
            const int64_t nj = (int64_t)a.tiles_d * a.tiles_h * a.tiles_w * a.groups * d->N;
            if (nj < ((int64_t)1 << 31)) {
                a.njobs = (int)nj;
                const dim3 g((unsigned)nj, 1, 1);
                if (side_head) {   // built for the default kernel form without addends (the global model's conv2)
                    if (!(narrow && wide && !nreg) || a.res || a.plane || d->Cout != 32 || (reinterpret_cast<uintptr_t>(y_head) & 15))
                        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward_side_head: needs the default Winograd form, 32 output "
                                                          "channels, no residual / depth planes and a 16-byte aligned y_head");
                    launch_wino_dma_variant<CfgWinoN3, false, false, 1>(a, g, as_stream(stream));
                    return check_launch("snvc_conv3d_forward_side_head");
                }
                if (stats) {       // the default kernel form without addends carries the statistics epilogue
                    if (!(narrow && wide && !nreg) || a.res || a.plane)
                        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward_stats: this layer does not take the default Winograd form");
                    launch_wino_dma_variant<CfgWinoN3, false, false, 3>(a, g, as_stream(stream));
                    return check_launch("snvc_conv3d_forward_stats(winograd)");
                }
                if (pooled) {      // built for the default kernel form without addends (the local trunk's conv4)
                    if (!(narrow && wide && !nreg) || a.res || a.plane)
                        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward: SNVC_EPI_AVGPOOL_D4 needs the default Winograd "
                                                          "form and no residual / depth planes");
                    launch_wino_dma_variant<CfgWinoN3, false, false, 2>(a, g, as_stream(stream));
                    return check_launch("snvc_conv3d_forward(pooled)");
                }
                if (big) launch_wino_dma<CfgWinoBig>(a, g, as_stream(stream));
                else if (narrow && wide && !nreg) launch_wino_dma<CfgWinoN3>(a, g, as_stream(stream));
                else if (narrow && wide) launch_wino<CfgWinoN>(a, g, as_stream(stream));
                else if (narrow) launch_wino<CfgWinoN8>(a, g, as_stream(stream));
                else if (wide) launch_wino<CfgWino>(a, g, as_stream(stream));
                else launch_wino<CfgWino8>(a, g, as_stream(stream));
                return check_launch("snvc_conv3d_forward(winograd)");
            }
        }
    }
    if (stats && !(p.kind == DC_M1 && a.vec && a.fast_epi && d->Win % 2 == 0 && !a.res && !head_w))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward_stats: the layer's rows do not allow the kernel forms that carry the statistics epilogue");
    if (side_head)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward_side_head: built for 3x3x3 / stride-1 layers on the Winograd path");
    if (pooled)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward: SNVC_EPI_AVGPOOL_D4 needs 16-byte aligned rows (the Winograd path)");
    const int64_t ntiles = (int64_t)p.tiles_d * p.tiles_h * p.tiles_w;
    const int64_t gx = d->transposed ? ntiles * (a.dc_planar ? 2 : 4) : ntiles;
    if (gx >= ((int64_t)1 << 31)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward: too many tiles");
    dim3 grid((unsigned)gx, (unsigned)p.groups, (unsigned)d->N);
    hipStream_t st = as_stream(stream);
    switch (p.kind) {
        case K1_M1: launch_conv<CfgK1M1>(a, grid, st); break;
        case K1_M2: launch_conv<CfgK1M2>(a, grid, st); break;
        case K3_M1: launch_conv<CfgK3M1>(a, grid, st); break;
        case K3_M2: launch_conv<CfgK3M2>(a, grid, st); break;
        case K3S2_M1: launch_conv<CfgK3S2M1, true>(a, grid, st); break;
        case K3S2_M2: launch_conv<CfgK3S2M2, true>(a, grid, st); break;
        case K5_M1: launch_conv<CfgK5M1, true>(a, grid, st); break;
        case K5_M2: launch_conv<CfgK5M2>(a, grid, st); break;
        case K5D2_M1: launch_conv<CfgK5D2M1>(a, grid, st); break;
        case K5D2_M2: launch_conv<CfgK5D2M2>(a, grid, st); break;
        case K7_M1: launch_conv<CfgK7M1, true>(a, grid, st); break;
        case K7_M2: launch_conv<CfgK7M2>(a, grid, st); break;
        case P1_M1S: launch_conv<CfgP1M1s>(a, grid, st); break;
        case P1S2_M1S: launch_conv<CfgP1S2M1s>(a, grid, st); break;
        case P3_M1S: launch_conv<CfgP3M1s, true>(a, grid, st); break;
        case P3S2_M1S: launch_conv<CfgP3S2M1s, true>(a, grid, st); break;
        case P7_M1S: launch_conv<CfgP7M1s>(a, grid, st); break;
        case DC_M1:
            if (!a.vec && vec8) { a.vec = 1; launch_deconv<CfgDCM1v8>(a, grid, st); }
            else launch_deconv<CfgDCM1>(a, grid, st);
            break;
        case DC_M2:
            if (!a.vec && vec8) { a.vec = 1; launch_deconv<CfgDCM2v8>(a, grid, st); }
            else launch_deconv<CfgDCM2>(a, grid, st);
            break;
        case DCP_M1:
            if (!a.vec && vec8) { a.vec = 1; launch_deconv<CfgDCPv8>(a, grid, st); }
            else launch_deconv<CfgDCP>(a, grid, st);
            break;
        default: return fail(SNVC_ERR_UNSUPPORTED, "snvc_conv3d_forward: no kernel");
    }
    return check_launch("snvc_conv3d_forward");
}

}  // namespace snvc
