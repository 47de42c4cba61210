// fp16-storage 3D convolution family for gfx950 (BASELINE.json configs[4]: "High-res local model ... 64ch,
// fp16 with MFMA").  The layers are the local trunk's
// (snvc/models/vernier.py:249-264: 1x1x1, 7x7x7, 5x5x5, dilated 5x5x5, 3x3x3) and the 16x hourglass's
// (snvc/models/submodule.py:223-268: 3x3x3 stride 1 / stride 2, ConvTranspose3d(k3,s2,p1,op1)).
// The reference has no fp16 path (its kernels dispatch float/double only and the model never calls .half()),
// so this mode is an extension whose parity target is this library's own fp32 path (tests/test_gpu_f16.py).
//
// Storage: activations and weights are IEEE half in HBM, products accumulate in fp32 on
// v_mfma_f32_32x32x16_f16, the per-channel affine / residual / ReLU run in fp32 on the accumulators and the
// result is rounded to half once, on the way out.
//
// Layout "C8": [N][C/8][D][H][W][8] -- eight consecutive channels of one voxel are one 16-byte piece.  That
// is the MFMA's own operand shape (a lane feeds 8 k-values = 8 channels of ONE voxel), so:
//   * staging is a plain copy: the LDS image of a chunk is [channel group][IN_D][IN_H][IN_W] pieces, filled by
//     LDS-DMA (global_load_lds_dwordx4, one piece per lane, padding pieces read a zero constant), with
//     per-VOXEL granularity -- no alignment classes along W as in the fp32 NCDHW kernels;
//   * a B fragment is ONE ds_read_b128 (32 consecutive voxels x 2 k-groups), conflict-free;
//   * the epilogue stores 16-byte pieces, 512 contiguous bytes per half-wave: the rows of the weight matrix
//     are permuted when packed so that accumulator register r of a lane is channel 16*(lane>>5) + r, i.e. a
//     lane ends up with two whole 8-channel pieces of its voxel.
// K = 16 per MFMA is either two channel groups of one tap (MODE 0: small kernels, two groups resident) or
// two TAPS of one channel group (MODE 1: 5^3 / 7^3 / stride-2 layers, whose LDS image of a single group is
// already 37-92 KB): lanes 32-63 then read the same image one tap further on, which is only a different
// (precomputed) lane base address.
// A fragments (weights) are NOT staged in LDS: a chunk of a 7^3 layer needs 343 KB of them.  Each wave streams
// them from L2 in consumption order through a small register ring (they are packed exactly in that order), so
// the chunk loop has no barrier besides the image hand-over and the LDS holds images only.
#include <type_traits>

#include "common.hpp"
#include "elementwise_internal.hpp"

namespace snvc {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ const unsigned g_zero16h[4] __attribute__((aligned(16))) = {0u, 0u, 0u, 0u};

struct F16Args {
    const _Float16 *x;
    const _Float16 *wp;
    const float *scale, *bias;
    const _Float16 *res;
    _Float16 *y;
    float *y_f32;        // EPI 1: [N][1][Dout][Hout][Wout] fp32 plane of output channel 0; EPI 2: [N][Cout][Dout][Hout][Wout] fp32
    // split mode (PL = 2, "f16x3"): every tensor is a pair of C8 half planes, value = hi + lo; x / res / y are the hi planes
    const _Float16 *x_lo, *res_lo;
    _Float16 *y_lo;
    const float *head;   // split EPI 0, Cout == 32: y_head[n][voxel] = sum_c head[c] * (the value written for channel c, unscaled: * head_mul)
    float *y_head;       //   fp32 plane [N][Dout][Hout][Wout] (the classifier's projection of the layer's own result, r3 side head)
    float head_mul;      //   2^-out_exp (EPI 2: applied to the fp32 result, which the epilogue forms in the residual's units)
    float res_mul;       // split residual: its stored units relative to the result's, 2^(e_y - e_res) (exact)
    int *overflow;       // split output: set to 1 if a value had to be clamped to half's range (the caller's exponent was too large)
    // EPI 4 (r5, "tail projection"): the layer's result is not stored; its contraction with a one-channel transposed layer's
    // weights W'[c][27 taps] is -- t_out[n][tap][class][pd][ph][pw] fp32 (see snvc_f16x3_deconv3d_tail_forward)
    const _Float16 *tail_w;   // [block m][8-channel group j][hi | lo][lane][8]: A fragments (32 tap rows x 16 channels), values * 2^w_exp
    float *t_out;
    float tail_mul;           // 2^-(e_y + w_exp): exact
    int64_t t_bs;             // floats between samples of t_out
    const float *res_f32;     // EPI 2 (r6): a float32 NCDHW tensor of the result's shape added to the stored result (ADD_POST; the training
                              //   step's data gradients take a skip connection's gradient this way), batch stride yf_bs; or NULL
    double *stats;            // EPI 2 (r6): per-(sample, slot, 32-channel group) fp64 (sum, sum of squares) of the stored float32 result, the layout
                              //   conv_stats_fold_kernel folds: [((n * T + slot) * groups + cg)][32][2], slot = (z-class * gridDim.x + blockIdx.x) * 4 + wave
    const float *x_mul;       // conv2d_x3q_kernel: device scalar the input pair was multiplied by (a power of two from its own maximum), or NULL
    int CGin;            // input channel groups (Cin / 8, rounded up)
    int Cout;
    int Din, Hin, Win;
    int Dout, Hout, Wout;      // dims of the OUTPUT TENSOR
    int nd, nh, nw;            // output positions this launch computes per dim (== Dout.. except for parity classes)
    int osd, osh, osw, offd, offh, offw;   // output coordinate = position * os + off
    int pad_d, pad_h, pad_w;   // image coordinate = position * STRIDE + tap * DIL - pad
    int isd, ish, iod, ioh;    // input coordinate (D, H) = image coordinate * is + io  (sub-grid classes; W is never scaled)
    int tiles_d, tiles_h, tiles_w;
    int nchunks, flags;
    int N, cls_mode;           // blockIdx.z = class * N + n.  cls_mode 0: one class; 1: the 8 parity classes of a transposed
                               // layer (own weights each); 2: the 4 (depth, height) sub-grids of a dilation-2 layer
    int64_t cls_wstride;       // halves between the packed weights of consecutive classes (cls_mode 1)
    int64_t x_bs, y_bs, r_bs, yf_bs;   // batch strides in elements
};

// A-fragment register ring depth (k-steps ahead).  Measured on cfg5 (k7 / k5 / k3 ms): 2: 9.70 / 1.92 / 0.91;
// 4: 9.38 / 2.00 / 0.91; 6: 8.99 / 1.82 / 0.85 (242-246 VGPRs, no spills); 8: 8.89 / 1.83 / 0.86 with spills.

template <int KD_, int KH_, int KW_, int STRIDE_, int DIL_, int MI_, int TD_, int TH_, int KCG_, int MODE_, bool DB_, int OCC_,
          int DILW_ = DIL_, int PL_ = 1, bool DYN_ = false>
struct F16Cfg {
    // DYN (split transposed layers): the 2x2x2 box of a parity class is walked over its REAL taps only -- (1+pd)(1+ph)(1+pw) of
    // them, 27 over the eight classes instead of 64 -- in pairs (two taps per MFMA), their offsets held in scalar registers.
    static constexpr bool DYN = DYN_;
    // PL = 2: split mode.  Activations and weights are (hi, lo) pairs of halves, value = hi + lo (22 significant bits), and a
    // product is evaluated as hi*hi + lo_w*hi_x + hi_w*lo_x on three v_mfma_f32_32x32x16_f16 with fp32 accumulation: the fp32
    // layers' contraction at fp32 accuracy (the dropped lo*lo term is 2^-22 of the product) on the 16x faster half pipe.
    // PL = 3: split mode with the two planes of the image taken one after the other (a pass over the hi plane: lo_w*hi_x +
    // hi_w*hi_x; a pass over the lo plane: hi_w*lo_x) -- half the LDS for the layers whose image is large (stride 2).
    static constexpr int PL = PL_;
    static constexpr bool SPLIT = PL_ >= 2, SERIAL = PL_ == 3;
    static constexpr int RES_PL = PL_ == 2 ? 2 : 1;         // planes resident in LDS
    static constexpr int PASSES = SERIAL ? 2 : 1;           // image passes per channel chunk
    static constexpr int PF = DYN_ ? 2 : (SPLIT ? 3 : 6);   // A-fragment ring depth (k-steps ahead)
    // DIL applies to D and H, DILW to W (they differ only for the sub-grid form of a dilated layer, see F16K5D2)
    static constexpr int KD = KD_, KH = KH_, KW = KW_, STRIDE = STRIDE_, DIL = DIL_, DILW = DILW_, MI = MI_, TD = TD_, TH = TH_;
    static constexpr int KCG = KCG_, MODE = MODE_, OCC = OCC_;
    static constexpr bool DB = DB_;
    static constexpr int IN_D = (TD - 1) * STRIDE + (KD - 1) * DIL + 1;
    static constexpr int IN_H = (TH - 1) * STRIDE + (KH - 1) * DIL + 1;
    static constexpr int IN_W = 31 * STRIDE + (KW - 1) * DILW + 1;
    static constexpr int VOX = IN_D * IN_H * IN_W;          // pieces per channel-group image
    static constexpr int GB = VOX * 16;                     // bytes per channel-group image
    static constexpr int ITEMS = RES_PL * KCG * VOX;        // image = [plane][channel group][voxel] pieces
    static constexpr int PLANE_BYTES = KCG * GB;
    static constexpr int NIT = (ITEMS + 255) / 256;
    static constexpr int IMG_BYTES = NIT * 256 * 16;        // whole DMA rounds
    static constexpr int LDS_BYTES = IMG_BYTES * (DB ? 2 : 1);
    static constexpr int NB = TD * TH / 4;
    static constexpr bool UNROLL_D = KD <= 3;
    static constexpr int SEGS = UNROLL_D ? 1 : KD;          // runtime-looped kernel depth slices
    static constexpr int TSEG = UNROLL_D ? KD * KH * KW : KH * KW;
    static constexpr int NPS = (TSEG + 1) / 2;              // MODE 1: tap pairs per segment
    static constexpr int NS = MODE == 0 ? TSEG * (KCG / 2) : NPS * KCG;   // k-steps per segment
    static constexpr int STEPS = SEGS * NS;                 // k-steps per chunk
    static constexpr int SEG_BYTES = DIL * IN_H * IN_W * 16;
    static_assert(TD * TH % 4 == 0, "rows split over 4 waves");
    static_assert(MODE == 1 || KCG % 2 == 0, "MODE 0 pairs channel groups");
    static_assert(NIT <= 32, "validity mask is one register");
    static_assert((TD * TH / 4) * 4 == TD * TH, "whole rows per wave");
    // voxel offset of tap t of a segment (kd = 0 for looped segments)
    static constexpr int tapoff(int t) {
        const int kd = UNROLL_D ? t / (KH * KW) : 0, kh = (t / KW) % KH, kw = t % KW;
        return ((kd * DIL) * IN_H + kh * DIL) * IN_W + kw * DILW;
    }
};

// MFMA row i of a 32-channel block holds output channel 16*((i>>2)&1) + 4*(i>>3) + (i&3): accumulator register r of
// lane l (row (r&3) + 8*(r>>2) + 4*(l>>5)) is then channel 16*(l>>5) + r.
__host__ __device__ constexpr int f16_row_channel(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

__device__ __forceinline__ int xcd_remap16(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, k = b >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// EPI 0: C8 half output (+affine, residual, ReLU); 1: fp32 plane of channel 0 (+Sigmoid); 2: fp32 NCDHW output (split mode);
// 3: EPI 0 + the side head (split mode, one 32-channel block); 4: tail projection (split transposed layer whose ONLY consumer is a
// transposed layer to one channel: the per-voxel contraction with that layer's 27 taps is done here, on the matrix pipe)
template <class Cfg, int EPI>
__global__ void __launch_bounds__(256, Cfg::OCC)
conv3d_f16_kernel(const F16Args a_) {
    F16Args a = a_;
    // cls_mode 1 (transposed layer): blockIdx.z = (pd, ph) * N + n and the WIDTH parity is the low bit of the tile index -- the two
    // classes that write the even and the odd voxels of the same output lines (and read the same residual lines) are neighbours
    // in launch order on one XCD, so their half-line stores meet in that L2 and leave it as whole lines (r4; with all eight
    // classes on blockIdx.z the fp32-output layer of the cfg2 hourglass wrote and read every line twice: 0.223 ms)
    int cls = blockIdx.z / a.N;
    int tile_raw = blockIdx.x, tiles_launch = a.tiles_d * a.tiles_h * a.tiles_w;
    if (a.cls_mode == 1) {
        tiles_launch *= 2;
        tile_raw = xcd_remap16(blockIdx.x, tiles_launch);
        cls = (cls << 1) | (tile_raw & 1);
        tile_raw >>= 1;
        a.offd = (cls >> 2) & 1; a.offh = (cls >> 1) & 1; a.offw = cls & 1;
        a.wp += cls * a.cls_wstride;
    } else if (a.cls_mode == 2) {
        const int pd = (cls >> 1) & 1, ph = cls & 1;
        a.offd = a.iod = pd; a.offh = a.ioh = ph;
        a.nd = (a.Dout - pd + 1) / 2; a.nh = (a.Hout - ph + 1) / 2;
    }
    constexpr int S = Cfg::STRIDE, MI = Cfg::MI, TD = Cfg::TD, TH = Cfg::TH, NB = Cfg::NB, KCG = Cfg::KCG;
    constexpr int IN_H = Cfg::IN_H, IN_W = Cfg::IN_W, VOX = Cfg::VOX, GB = Cfg::GB, NIT = Cfg::NIT, ITEMS = Cfg::ITEMS;
    constexpr int PF = Cfg::PF, PL = Cfg::PL;
    constexpr bool SPLIT = Cfg::SPLIT, SERIAL = Cfg::SERIAL;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int ntiles = a.tiles_d * a.tiles_h * a.tiles_w;
    const int t = a.cls_mode == 1 ? tile_raw : xcd_remap16(blockIdx.x, ntiles);
    const int tw = t % a.tiles_w, th = (t / a.tiles_w) % a.tiles_h, td = t / (a.tiles_w * a.tiles_h);
    const int cb = blockIdx.y;              // block of 32*MI output channels
    const int64_t n = blockIdx.z % a.N;
    const int od0 = td * TD, oh0 = th * TH, ow0 = tw * 32;
    if (od0 >= a.nd || oh0 >= a.nh) return;      // a smaller class of an odd extent: whole tile outside (block-uniform)
    const int id0 = od0 * S - a.pad_d, ih0 = oh0 * S - a.pad_h, iw0 = ow0 * S - a.pad_w;

    f32x16 acc[NB][MI];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][m][r] = 0.0f;

    // ---- staging geometry: piece i of the image = (group, dd, hh, ww); its source offset (in pieces, relative to
    // the chunk's first channel group) and validity do not depend on the chunk
    const int in_hw = a.Hin * a.Win, in_dhw = in_hw * a.Din;
    unsigned off[NIT];
    unsigned vmask = 0, gmask = 0, pmask = 0;           // per piece: valid, its channel group (KCG <= 2), its plane
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int pg = i / VOX, r = i - pg * VOX;
        const int plane = pg / KCG, g = pg - plane * KCG;
        gmask |= (unsigned)(g & 1) << it;
        pmask |= (unsigned)(plane & 1) << it;
        const int dd = r / (IN_H * IN_W), r2 = r - dd * (IN_H * IN_W);
        const int hh = r2 / IN_W, ww = r2 - hh * IN_W;
        const int gd = (id0 + dd) * a.isd + a.iod, gh = (ih0 + hh) * a.ish + a.ioh, gw = iw0 + ww;
        const bool ok = i < ITEMS && (unsigned)gd < (unsigned)a.Din && (unsigned)gh < (unsigned)a.Hin &&
                        (unsigned)gw < (unsigned)a.Win;
        off[it] = ok ? (unsigned)(g * in_dhw + gd * in_hw + gh * a.Win + gw) : 0u;
        vmask |= (ok ? 1u : 0u) << it;
    }
    static_assert(KCG <= 2, "gmask holds one bit per piece");
    const _Float16 *xn = a.x + n * a.x_bs, *xn_lo = SPLIT ? a.x_lo + n * a.x_bs : nullptr;
    const int wbase = tid & ~63;
    auto issue = [&](int pass, int buf) {           // pass = chunk (* 2 + plane when the planes are taken serially)
        const int chunk = SERIAL ? pass >> 1 : pass;
        const bool lo_pass = SERIAL && (pass & 1);
        const int64_t coff = (int64_t)chunk * KCG * in_dhw * 8;
        const int cg_left = a.CGin - chunk * KCG;
        char *const ibuf = lds + buf * Cfg::IMG_BYTES;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * 256 + tid;
            const bool ok = ((vmask >> it) & 1u) && (KCG == 1 || (int)((gmask >> it) & 1u) < cg_left);
            const _Float16 *xc = (((PL == 2 && ((pmask >> it) & 1u)) || lo_pass) ? xn_lo : xn) + coff;
            const void *src = ok ? static_cast<const void *>(xc + (size_t)off[it] * 8) : static_cast<const void *>(g_zero16h);
            if (ITEMS % 256 == 0 || i < ITEMS)
                __builtin_amdgcn_global_load_lds(static_cast<const float *>(src),
                                                 reinterpret_cast<float *>(ibuf + (it * 256 + wbase) * 16), 16, 0, 0);
        }
    };

    // ---- B-fragment addressing (bytes inside an image).  Lane (l & 31) = output column; lanes 32..63 feed the
    // second k-group: the next channel group (MODE 0) or the next tap (MODE 1) -- a constant added to the lane base.
    const int lanebase = (lane & 31) * S * 16;
    int rowoff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row = wave * NB + nb;
        rowoff[nb] = ((row / TH) * S * IN_H + (row % TH) * S) * IN_W * 16;
    }
    constexpr int D_SAME = Cfg::DILW;
    constexpr int D_ROW = Cfg::DIL * IN_W - (Cfg::KW - 1) * Cfg::DILW;
    constexpr int D_SLICE = Cfg::DIL * IN_H * IN_W - (Cfg::KH - 1) * Cfg::DIL * IN_W - (Cfg::KW - 1) * Cfg::DILW;
    const int b_grp = lanebase + half * GB;
    const int b_same = lanebase + half * D_SAME * 16;
    const int b_row = lanebase + half * D_ROW * 16;
    const int b_slice = lanebase + half * D_SLICE * 16;
    const int b_none = lanebase;

    // ---- A fragments: [cout block][chunk][segment][k-step][m][lane] pieces, consumed in exactly that order
    // (split mode: [..][k-step][m][hi | lo][lane])
    constexpr int MA = MI * (SPLIT ? 2 : 1);
    // DYN: this class's real taps, in (kd, kh, kw) order with kw fastest; pair p = taps (2p, 2p + 1)
    int dyn_np = 0, dyn_toff[4] = {0, 0, 0, 0}, dyn_base[4] = {0, 0, 0, 0};
    if constexpr (Cfg::DYN) {
        const int nw = 1 + (cls & 1), nh = 1 + ((cls >> 1) & 1), ndp = 1 + ((cls >> 2) & 1);
        const int T = nw * nh * ndp;
        dyn_np = T > 1 ? T / 2 : 1;
        auto vox = [&](int t) { return (((t / (nw * nh)) * IN_H) + (t / nw) % nh) * IN_W + t % nw; };
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ta = 2 * p, tb = 2 * p + 1;
            dyn_toff[p] = vox(ta) * 16;
            dyn_base[p] = lanebase + half * ((tb < T ? vox(tb) - vox(ta) : 0) * 16);
        }
    }
    const int64_t steps_total = Cfg::DYN ? (int64_t)a.nchunks * 2 * dyn_np : (int64_t)a.nchunks * Cfg::PASSES * Cfg::STEPS;
    const h8 *wq = reinterpret_cast<const h8 *>(a.wp) + ((int64_t)cb * steps_total * MA) * 64 + lane;
    h8 q[PF][MA];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
#pragma unroll
        for (int m = 0; m < MA; ++m) q[i][m] = wq[m * 64];
        wq += MA * 64;     // the packed buffer carries PF steps of zero padding behind the last block
    }

    auto compute = [&](const char *img, bool lo_pass) {
        if constexpr (Cfg::DYN) {
            static_assert(!Cfg::DYN || (PL == 2 && KCG == 2 && Cfg::MODE == 1 && PF == 2), "DYN: resident split planes, two groups, tap pairs");
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (p >= dyn_np) break;                  // block-uniform
#pragma unroll
                for (int g = 0; g < 2; ++g) {            // k-step (p, g): ring slot g, refilled with the step two ahead
                    h8 af[MA];
#pragma unroll
                    for (int m = 0; m < MA; ++m) { af[m] = q[g][m]; q[g][m] = wq[m * 64]; }
                    wq += MA * 64;
                    h8 bf[NB], bl[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const char *src = img + g * GB + dyn_toff[p] + dyn_base[p] + rowoff[nb];
                        bf[nb] = *reinterpret_cast<const h8 *>(src);
                        bl[nb] = *reinterpret_cast<const h8 *>(src + Cfg::PLANE_BYTES);
                    }
#pragma unroll
                    for (int m = 0; m < MI; ++m) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2 * m + 1], bf[nb], acc[nb][m], 0, 0, 0);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2 * m], bl[nb], acc[nb][m], 0, 0, 0);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2 * m], bf[nb], acc[nb][m], 0, 0, 0);
                    }
                }
            }
            return;
        }
        // B fragments are double-buffered in registers: the LDS reads of k-step s + 1 are issued BEFORE the MFMAs of step s (r4;
        // the compiler's own schedule sank every ds_read to just in front of its MFMA -- an exposed LDS round trip per fragment,
        // hidden only by the other waves of the SIMD: 0.76 of the pipe on the bare loop at 3 waves per SIMD, 0.64 at 2)
        constexpr int NBF = NB * (PL == 2 ? 2 : 1);
        auto step_addr = [&](int s, int &base, int &toff) {      // s is a constant after unrolling
            if constexpr (Cfg::MODE == 0) {
                const int tp = s / (KCG / 2), j = s % (KCG / 2);
                base = b_grp;
                toff = 2 * j * GB + Cfg::tapoff(tp) * 16;
            } else {
                const int g = s / Cfg::NPS, p = s % Cfg::NPS;
                const int ta = 2 * p, tb = 2 * p + 1;
                toff = g * GB + Cfg::tapoff(ta) * 16;
                if (tb >= Cfg::TSEG) base = b_none;              // odd tap count: the pair's second half has zero weights
                else {
                    const int delta = Cfg::tapoff(tb) - Cfg::tapoff(ta);
                    base = delta == D_SAME ? b_same : (delta == D_ROW ? b_row : b_slice);
                }
            }
        };
        h8 bfr[2][NBF];
        auto load_b = [&](int buf, const char *simg, int s) {
            int base, toff;
            step_addr(s, base, toff);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                bfr[buf][nb] = *reinterpret_cast<const h8 *>(simg + base + toff + rowoff[nb]);
                if constexpr (PL == 2)
                    bfr[buf][NB + nb] = *reinterpret_cast<const h8 *>(simg + Cfg::PLANE_BYTES + base + toff + rowoff[nb]);
            }
        };
        load_b(0, img, 0);
#pragma unroll 1
        for (int seg = 0; seg < Cfg::SEGS; ++seg) {
            const char *simg = img + seg * Cfg::SEG_BYTES;
#pragma unroll
            for (int s = 0; s < Cfg::NS; ++s) {
                const int cur = s & 1, nxt = cur ^ 1;
                h8 af[MA];
#pragma unroll
                for (int m = 0; m < MA; ++m) af[m] = q[0][m];
#pragma unroll
                for (int i = 0; i + 1 < PF; ++i)
#pragma unroll
                    for (int m = 0; m < MA; ++m) q[i][m] = q[i + 1][m];
                // (r5, measured and dropped: not fetching w_lo for the steps of a lo pass -- they multiply w_hi alone.  The branch
                // around two of the four loads costs the counted vmcnt waits of the ring: hg conv1 0.345 -> 0.79 ms.)
#pragma unroll
                for (int m = 0; m < MA; ++m) q[PF - 1][m] = wq[m * 64];
                wq += MA * 64;
                // the next step's fragments (behind a depth slice's last step: the first step of the next slice)
                if (s + 1 < Cfg::NS) load_b(nxt, simg, s + 1);
                else if (Cfg::SEGS > 1 && seg + 1 < Cfg::SEGS) load_b(nxt, simg + Cfg::SEG_BYTES, 0);
                __builtin_amdgcn_sched_barrier(0);          // the loads stay up here: nothing of them sinks into the MFMA block
                if constexpr (PL == 2) {
                    // the three product terms row by row: consecutive MFMAs write different accumulators (no back-to-back
                    // dependence), the two correction terms before the leading one
#pragma unroll
                    for (int m = 0; m < MI; ++m) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2 * m + 1], bfr[cur][nb], acc[nb][m], 0, 0, 0);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2 * m], bfr[cur][NB + nb], acc[nb][m], 0, 0, 0);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2 * m], bfr[cur][nb], acc[nb][m], 0, 0, 0);
                    }
                } else if constexpr (SERIAL) {
#pragma unroll
                    for (int m = 0; m < MI; ++m) {
                        if (!lo_pass) {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[nb][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2 * m + 1], bfr[cur][nb], acc[nb][m], 0, 0, 0);
                        }
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2 * m], bfr[cur][nb], acc[nb][m], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < MI; ++m)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m], bfr[cur][nb], acc[nb][m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (Cfg::SEGS > 1 && Cfg::NS % 2 == 1) {      // an odd number of steps: the prefetched slice start sits in buffer 1
#pragma unroll
                for (int i = 0; i < NBF; ++i) bfr[0][i] = bfr[1][i];
            }
        }
    };

    const int npass = a.nchunks * Cfg::PASSES;
    if constexpr (Cfg::DB) {
        issue(0, 0);
        __syncthreads();
        for (int ps = 0; ps < npass; ++ps) {
            if (ps + 1 < npass) issue(ps + 1, (ps + 1) & 1);
            compute(lds + (ps & 1) * Cfg::IMG_BYTES, SERIAL && (ps & 1));
            __syncthreads();     // drains the DMA of the next pass and retires every read of this pass's buffer
        }
    } else {
        for (int ps = 0; ps < npass; ++ps) {
            issue(ps, 0);
            __syncthreads();
            compute(lds, SERIAL && (ps & 1));
            __syncthreads();
        }
    }

    // ---- epilogue: lane = voxel (lane & 31) of row nb; registers r = channels 16*half + r of block m
    const int out_hw = a.Hout * a.Wout;
    const int64_t out_dhw = (int64_t)out_hw * a.Dout;
    const int pw_ = ow0 + (lane & 31);
    const bool relu = (a.flags & SNVC_EPI_RELU) != 0, add_pre = (a.flags & SNVC_EPI_ADD_PRE) != 0,
               add_post = (a.flags & SNVC_EPI_ADD_POST) != 0;
    int64_t sp[NB];
    bool okv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row = wave * NB + nb;
        const int pd = od0 + row / TH, ph = oh0 + row % TH;
        okv[nb] = pd < a.nd && ph < a.nh && pw_ < a.nw;
        sp[nb] = okv[nb] ? ((int64_t)(pd * a.osd + a.offd) * out_hw + (ph * a.osh + a.offh) * a.Wout + (pw_ * a.osw + a.offw)) : 0;
    }
    if constexpr (EPI == 1) {
        // one output channel (row 0 of block 0 = register 0 of the lanes with half == 0), Sigmoid, fp32 plane
        const float sc = a.scale ? a.scale[0] : 1.0f, bi = a.scale ? a.bias[0] : 0.0f;
        float *yp = a.y_f32 + n * a.yf_bs;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float v = acc[nb][0][0] * sc + bi;
            if (relu) v = v > 0.0f ? v : 0.0f;
            if (a.flags & SNVC_EPI_SIGMOID) v = 1.0f / (1.0f + expf(-v));
            if (okv[nb] && half == 0) yp[sp[nb]] = v;
        }
    } else if constexpr (EPI == 4) {
        // The layer's result v = act(affine(acc) + residual) (64 channels per voxel, never stored) feeds ONLY a transposed layer to one
        // channel (the global stack's folded tail: classifier(bn(deconv(v)) + ...) -- reference submodule.py:127-146,166 with
        // vernier.py:366-371's composition).  That layer is out[o] = sum over (voxel i, tap k) with o = 2 i - 1 + k of T[k][i],
        // T[k][i] = sum_c W'[c][k] v[c][i]: a 27 x C contraction PER VOXEL, independent of every other voxel -- done here as three
        // half-precision MFMAs per 8-channel group on the (hi, lo) pair of v (the same split arithmetic as the layer itself; the B operand of
        // v_mfma_f32_32x32x16_f16 is exactly what a lane holds: 8 channels of one voxel).  Written: 27 fp32 planes per class instead of
        // C channels of fp32 (0.42 of the bytes at C = 64), contiguous along W inside a class (the NCDHW form stored 4 bytes at an 8-byte stride).
        static_assert(EPI != 4 || SPLIT, "tail projection: split mode");
        const _Float16 *rn = a.res ? a.res + n * a.r_bs : nullptr, *rn_lo = a.res ? a.res_lo + n * a.r_bs : nullptr;
        h8 tw[MI][2][2];
        {
            const h8 *twp = reinterpret_cast<const h8 *>(a.tail_w) + lane;
#pragma unroll
            for (int m = 0; m < MI; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) tw[m][j][s2] = twp[((m * 2 + j) * 2 + s2) * 64];
        }
        constexpr float kHalfMax = 65504.0f;
        float vmax = 0.0f;
        const int64_t cls_vox = (int64_t)a.nd * a.nh * a.nw;
        float *tn = a.t_out + n * a.t_bs + (int64_t)cls * cls_vox;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int row = wave * NB + nb;
            const int pd = od0 + row / TH, ph = oh0 + row % TH;
            f32x16 tacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) tacc[r] = 0.0f;
#pragma unroll
            for (int m = 0; m < MI; ++m) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int cj = m * 32 + 16 * half + 8 * j;
                    f32x4 sc[2], bi[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        sc[k] = a.scale ? *reinterpret_cast<const f32x4 *>(a.scale + cj + 4 * k) : f32x4(1.0f);
                        bi[k] = a.scale ? *reinterpret_cast<const f32x4 *>(a.bias + cj + 4 * k) : f32x4(0.0f);
                    }
                    h8 rv = h8((_Float16)0.0f), rl = h8((_Float16)0.0f);
                    if (rn) {
                        const int64_t gj = (int64_t)(cj >> 3) * out_dhw;
                        rv = *reinterpret_cast<const h8 *>(rn + (gj + sp[nb]) * 8);
                        rl = *reinterpret_cast<const h8 *>(rn_lo + (gj + sp[nb]) * 8);
                    }
                    h8 o, ol;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float v = __builtin_fmaf(acc[nb][m][8 * j + e], sc[e >> 2][e & 3], bi[e >> 2][e & 3]);
                        const float rr = ((float)rv[e] + (float)rl[e]) * a.res_mul;
                        if (add_pre) v += rr;
                        if (relu) v = __builtin_fmaxf(v, 0.0f);
                        if (add_post) v += rr;
                        v = __builtin_amdgcn_fmed3f(v, -kHalfMax, kHalfMax);
                        vmax = __builtin_fmaxf(vmax, okv[nb] ? __builtin_fabsf(v) : 0.0f);
                        o[e] = (_Float16)v;
                        ol[e] = (_Float16)(v - (float)o[e]);
                    }
                    tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tw[m][j][1], o, tacc, 0, 0, 0);
                    tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tw[m][j][0], ol, tacc, 0, 0, 0);
                    tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tw[m][j][0], o, tacc, 0, 0, 0);
                }
            }
            if (okv[nb]) {
                const int64_t spc = ((int64_t)pd * a.nh + ph) * a.nw + pw_;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tap = (r & 3) + 8 * (r >> 2) + 4 * half;       // accumulator row of this register
                    if (tap < 27) tn[(int64_t)tap * 8 * cls_vox + spc] = tacc[r] * a.tail_mul;
                }
            }
        }
        if (vmax >= kHalfMax && a.overflow) atomicOr(a.overflow, 1);
    } else {
        // EPI 0: C8 half output (split mode: a (hi, lo) pair); EPI 2: fp32 NCDHW output (a split-mode layer handing its result
        // to the fp32 kernels).  Residual: C8 (split mode: a pair).
        const _Float16 *rn = a.res ? a.res + n * a.r_bs : nullptr;
        const _Float16 *rn_lo = (SPLIT && a.res) ? a.res_lo + n * a.r_bs : nullptr;
        constexpr bool C8OUT = EPI == 0 || EPI == 3, side = EPI == 3 && SPLIT && MI == 1;
        _Float16 *yn = C8OUT ? a.y + n * a.y_bs : nullptr;
        _Float16 *yn_lo = (C8OUT && SPLIT) ? a.y_lo + n * a.y_bs : nullptr;
        float *yf = EPI == 2 ? a.y_f32 + n * a.yf_bs : nullptr;
        const float *__restrict__ rf = (EPI == 2 && a.res_f32) ? a.res_f32 + n * a.yf_bs : nullptr;
        constexpr float kHalfMax = 65504.0f;
        // Epilogue arithmetic is counted in VALU instructions (every one costs matrix-pipe time of the co-resident waves): the
        // affine is one FMA, ReLU and the clamp to half's range are ONE v_med3 (lower bound 0 or -65504), the overflow flag is a
        // running max of |v| compared once at the end, and the residual's conversions exist only in the with-residual instance.
        const float lo_bound = relu ? 0.0f : -kHalfMax;
        float vmax = 0.0f;
        const bool want_stats = EPI == 2 && a.stats != nullptr;
        // EPI 2 (r6): the input pair held x * x_mul[0] (a device-side power of two): taken out of the channel scale here, exactly
        const float x_inv = (EPI == 2 && a.x_mul) ? 1.0f / a.x_mul[0] : 1.0f;
        auto run = [&](auto has_res_tag) {
            constexpr bool HAS_RES = decltype(has_res_tag)::value;
#pragma unroll
            for (int m = 0; m < MI; ++m) {
                if ((cb * MI + m) * 32 >= a.Cout) break;                // Cout = 32 * odd: the last block is half empty
                const int c0 = (cb * MI + m) * 32 + 16 * half;          // first of this lane's 16 channels
                float hsum[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) hsum[nb] = 0.0f;
                float st_s[2][8], st_q[2][8];          // EPI 2 with a.stats: this lane's sums over its voxels, per channel
                if constexpr (EPI == 2) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) st_s[j][e] = st_q[j][e] = 0.0f;
                }
                // one 8-channel group (a C8 piece) at a time: its affine / head weights are live only while its pieces are formed
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int cj = c0 + 8 * j;
                    f32x4 sc[2], bi[2], hw4[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        sc[k] = (a.scale ? *reinterpret_cast<const f32x4 *>(a.scale + cj + 4 * k) : f32x4(1.0f)) * x_inv;
                        bi[k] = a.scale ? *reinterpret_cast<const f32x4 *>(a.bias + cj + 4 * k) : f32x4(0.0f);
                        if constexpr (side) hw4[k] = *reinterpret_cast<const f32x4 *>(a.head + cj + 4 * k);
                    }
                    const int64_t gj = (int64_t)(cj >> 3) * out_dhw;        // this group's plane of pieces
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        h8 rv = h8((_Float16)0.0f), rl = h8((_Float16)0.0f);
                        if constexpr (HAS_RES) {
                            rv = *reinterpret_cast<const h8 *>(rn + (gj + sp[nb]) * 8);
                            if (rn_lo) rl = *reinterpret_cast<const h8 *>(rn_lo + (gj + sp[nb]) * 8);
                        }
                        h8 o, ol;
                        float rfv[8];
                        if constexpr (EPI == 2) {
                            // the float32 residual's eight values in flight together, ahead of the stores (which the compiler must
                            // otherwise keep in order with them: one exposed HBM latency per channel, measured 2x on the whole kernel)
                            if (rf && okv[nb]) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) rfv[e] = rf[(int64_t)(cj + e) * out_dhw + sp[nb]];
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float v = __builtin_fmaf(acc[nb][m][8 * j + e], sc[e >> 2][e & 3], bi[e >> 2][e & 3]);
                            if constexpr (HAS_RES) {
                                const float rr = ((float)rv[e] + (float)rl[e]) * a.res_mul;
                                if (add_pre) v += rr;
                                if (relu) v = __builtin_fmaxf(v, 0.0f);
                                if (add_post) v += rr;
                            }
                            if constexpr (EPI == 2) {
                                if constexpr (!HAS_RES) { if (relu) v = __builtin_fmaxf(v, 0.0f); }
                                if (okv[nb]) {
                                    const int64_t at = (int64_t)(cj + e) * out_dhw + sp[nb];
                                    yf[at] = rf ? __builtin_fmaf(v, a.head_mul, rfv[e]) : v * a.head_mul;     // 2^-e_y: exact
                                    if (want_stats) {
                                        const float sv = v * a.head_mul;
                                        st_s[j][e] += sv;
                                        st_q[j][e] = __builtin_fmaf(sv, sv, st_q[j][e]);
                                    }
                                }
                            } else {
                                if constexpr (SPLIT) {
                                    // ReLU (when no residual follows it) and the clamp that keeps the pair finite, in one v_med3;
                                    // the flag looks at what is STORED (after the activation) and only at voxels of the tensor
                                    v = __builtin_amdgcn_fmed3f(v, HAS_RES ? -kHalfMax : lo_bound, kHalfMax);
                                    vmax = __builtin_fmaxf(vmax, okv[nb] ? __builtin_fabsf(v) : 0.0f);
                                } else if constexpr (!HAS_RES) {
                                    if (relu) v = __builtin_fmaxf(v, 0.0f);
                                }
                                if constexpr (side) hsum[nb] += hw4[e >> 2][e & 3] * v;
                                o[e] = (_Float16)v;
                                if constexpr (SPLIT) ol[e] = (_Float16)(v - (float)o[e]);
                            }
                        }
                        if constexpr (C8OUT) {
                            if (okv[nb]) {
                                *reinterpret_cast<h8 *>(yn + (gj + sp[nb]) * 8) = o;
                                if constexpr (SPLIT) *reinterpret_cast<h8 *>(yn_lo + (gj + sp[nb]) * 8) = ol;
                            }
                        }
                    }
                }
                if constexpr (EPI == 2) {
                    if (want_stats) {
                        // the 32 lanes of a half-wave hold the same 16 channels at 32 columns: fold them, lane 0 of the half writes its slot
                        const int zc = blockIdx.z / a.N, groups = (a.Cout + 31) >> 5;
                        const int64_t slots = (int64_t)(gridDim.z / a.N) * gridDim.x * 4;
                        const int64_t slot = ((int64_t)zc * gridDim.x + blockIdx.x) * 4 + wave;
                        double *dst = a.stats + ((((int64_t)n * slots + slot) * groups + (cb * MI + m)) * 32 + 16 * half) * 2;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float s0 = st_s[j][e], s1 = st_q[j][e];
#pragma unroll
                                for (int o = 16; o >= 1; o >>= 1) {
                                    s0 += __shfl_xor(s0, o, 64);
                                    s1 += __shfl_xor(s1, o, 64);
                                }
                                if ((lane & 31) == 0) {
                                    dst[(8 * j + e) * 2] = (double)s0;
                                    dst[(8 * j + e) * 2 + 1] = (double)s1;
                                }
                            }
                    }
                }
                if constexpr (side) {     // the two half-waves hold channels 0..15 / 16..31 of the same 32 voxels
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float tot = hsum[nb] + __shfl_xor(hsum[nb], 32, 64);
                        if (okv[nb] && half == 0) a.y_head[n * out_dhw + sp[nb]] = tot * a.head_mul;
                    }
                }
            }
        };
        if (rn) run(std::true_type{});
        else run(std::false_type{});
        const bool clamped = vmax >= kHalfMax;      // a NaN never raises the flag (fmaxf drops it)
        if constexpr (C8OUT && SPLIT) {
            if (clamped && a.overflow) atomicOr(a.overflow, 1);
        }
    }
}

// ------------------------------------------------------------------------------------ 16x16x32 form of the 3x3x3 layer (r4)
// Split-mode 3x3x3 / stride-1 layer on v_mfma_f32_16x16x32_f16 instead of 32x32x16: under the chip's power limit the 16x16x32
// shape sustains 18-26 % more flops on dense random operands (a quarter of the accumulator updates per flop; tools/micro/
// mfma_power.hip).  Same tile as conv3d_f16_kernel's F16K3X (4 x 4 x 32 voxels x 32 output channels per workgroup).
//   K = 32 per MFMA = 4 k-blocks of one C8 piece: k-block kb = lane >> 4 is ONE OF FOUR TAPS of one channel group (tap 4 q + kb of the
//   27 in (kd, kh, kw) raster order: 7 quads = 28 slots, the 28th with zero weights), so a chunk is one channel group and its image
//   -- both planes, [plane][voxel] pieces, 39 KB -- is DOUBLE-BUFFERED at two workgroups per CU (2 x 78 KB);
//   B fragment (rows = k, 16 voxels): lane (voxel lane & 15, kb) reads ONE piece -- 16 voxels of the row's left or right half;
//   A fragment (16 output channels x 32 k): row i of co-half h is channel 8 * (i >> 2) + 4 * h + (i & 3) of the block, so that
//   lane (kb, voxel) ends up with accumulator rows 4 kb .. 4 kb + 3 of both halves = the 8 channels of C8 group kb: one piece.
//   Per k-step (a tap quad): 4 A fragments (2 halves x hi | lo), 16 B fragments (4 rows x 2 halves of the row x hi | lo, fetched
//   in two half-steps of 8), 48 MFMAs (16 cycles each).
// (First form, measured and replaced: two taps x two channel groups per MFMA -- an 80 KB image, single-buffered, whose refills cost
// 10 % of conv2: 0.838 ms against 0.818 for this one; profiles/r4/kernel_experiments_r4.txt items 16-18.)
// How the refill overlaps the MFMAs -- three facts about the toolchain and the hardware decide the shape of the loop:
//   * the backend drains every LDS-DMA in flight (s_waitcnt vmcnt(0)) in front of the first ds_read that follows one: it cannot tell
//     the two buffers apart (the DMA intrinsic carries no alias scope; __restrict__ does not reach it).  So the B-fragment reads are
//     inline-asm ds_read_b128 with hand-counted lgkmcnt waits: the 8 reads of half-step hs + 1 are issued in front of the MFMAs of
//     half-step hs, whose own 8 reads are then exactly what `lgkmcnt(8)` waits for; the wait names the registers as in/out operands
//     so that no MFMA moves in front of it;
//   * with an LDS-DMA in flight, every compiler-placed vmcnt wait is vmcnt(0) (the DMA is a FLAT-class access to two address spaces:
//     "pending flat");
//   * vmcnt retires in order anyway: waiting for a weight fragment issued after a DMA waits for that DMA.
//   So the DMA of the next chunk is issued LATE -- in front of k-step FILL_STEP of 7, after a forced wait for the weight fragments
//   still in flight -- and no register load issued after it is consumed before the barrier that ends the chunk (FILL_STEP = 7 - PF):
//   the refill has the last PF k-steps of MFMAs to land in.
typedef float f32x4q __attribute__((ext_vector_type(4)));

struct X3QCfg {
    static constexpr int TD = 4, TH = 4, NB = 4, IN_D = 6, IN_H = 6, IN_W = 34, VOX = IN_D * IN_H * IN_W, GB = VOX * 16;
    static constexpr int ITEMS = 2 * VOX, PLANE_BYTES = GB, NIT = (ITEMS + 255) / 256;
    static constexpr int IMG_BYTES = NIT * 256 * 16, LDS_BYTES = 2 * IMG_BYTES;
    static constexpr int NQ = 7, PF = 2, FILL_STEP = NQ - PF;      // a fragment fetched at k-step s is consumed at s + PF
};

template <int IMM>
__device__ __forceinline__ void lds_read_b128_to(h8 &v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_lgkm_for(h8 (&b)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
                 : "n"(N)
                 : "memory");
}

template <int EPI>      // 0: split C8 output; 3: + the side head (Cout == 32); 2 (r6): float32 NCDHW output (+ float32 residual, + statistics)
__global__ void __launch_bounds__(256, 2)
conv3d_x3q_kernel(const F16Args a) {
    using Cfg = X3QCfg;
    constexpr int NB = Cfg::NB, TH = Cfg::TH, IN_H = Cfg::IN_H, IN_W = Cfg::IN_W, VOX = Cfg::VOX, NIT = Cfg::NIT, NQ = Cfg::NQ;
    constexpr int ITEMS = Cfg::ITEMS, PF = Cfg::PF;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kb = lane >> 4, col = lane & 15;
    // the output-channel blocks of a tile are consecutive jobs of the same XCD: the second reads the tile's input from that XCD's L2
    // (as separate grid rows -- blockIdx.y -- the 64 -> 64 hourglass layer fetched its input twice: 299 MB against 159)
    const int ntiles = a.tiles_d * a.tiles_h * a.tiles_w, cblocks = a.Cout >> 5;
    const int job = xcd_remap16(blockIdx.x, ntiles * cblocks);
    const int t = job / cblocks, cb = job - t * cblocks;
    const int tw = t % a.tiles_w, th = (t / a.tiles_w) % a.tiles_h, td = t / (a.tiles_w * a.tiles_h);
    const int64_t n = blockIdx.z;
    const int od0 = td * Cfg::TD, oh0 = th * TH, ow0 = tw * 32;
    const int id0 = od0 - a.pad_d, ih0 = oh0 - a.pad_h, iw0 = ow0 - a.pad_w;

    f32x4q acc[NB][2][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[nb][ph][h] = f32x4q{0.0f, 0.0f, 0.0f, 0.0f};

    const int in_hw = a.Hin * a.Win, in_dhw = in_hw * a.Din;
    unsigned off[NIT];
    unsigned vmask = 0, pmask = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int plane = i / VOX, r = i - plane * VOX;
        pmask |= (unsigned)(plane & 1) << it;
        const int dd = r / (IN_H * IN_W), r2 = r - dd * (IN_H * IN_W);
        const int hh = r2 / IN_W, ww = r2 - hh * IN_W;
        const int gd = id0 + dd, gh = ih0 + hh, gw = iw0 + ww;
        const bool ok = i < ITEMS && (unsigned)gd < (unsigned)a.Din && (unsigned)gh < (unsigned)a.Hin && (unsigned)gw < (unsigned)a.Win;
        off[it] = ok ? (unsigned)(gd * in_hw + gh * a.Win + gw) : 0u;
        vmask |= (ok ? 1u : 0u) << it;
    }
    const _Float16 *xn = a.x + n * a.x_bs, *xn_lo = a.x_lo + n * a.x_bs;
    const int wbase = tid & ~63;
    auto issue = [&](int chunk, int buf) {     // every wave issues all NIT instructions (pieces past the image land in the buffer's padding)
        const int64_t coff = (int64_t)chunk * in_dhw * 8;
        char *const ibuf = lds + buf * Cfg::IMG_BYTES;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const _Float16 *xc = (((pmask >> it) & 1u) ? xn_lo : xn) + coff;
            const void *src = ((vmask >> it) & 1u) ? static_cast<const void *>(xc + (size_t)off[it] * 8) : static_cast<const void *>(g_zero16h);
            __builtin_amdgcn_global_load_lds(static_cast<const float *>(src), reinterpret_cast<float *>(ibuf + (it * 256 + wbase) * 16), 16, 0, 0);
        }
    };
    int qoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int tp = 4 * q + kb;
        tp = tp < 27 ? tp : 26;                     // the 28th slot: zero weights, tap 26's piece
        qoff[q] = (((tp / 9) * IN_H + (tp / 3) % 3) * IN_W + tp % 3 + col) * 16;
    }
    int rowoff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row = wave * NB + nb;
        rowoff[nb] = ((row / TH) * IN_H + (row % TH)) * IN_W * 16;
    }
    const int64_t steps_total = (int64_t)a.nchunks * NQ;
    const h8 *wq = reinterpret_cast<const h8 *>(a.wp) + ((int64_t)cb * steps_total * 4) * 64 + lane;
    h8 q_[PF][4];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) q_[i][m] = wq[m * 64];
        wq += 4 * 64;
    }

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    issue(0, 0);
    __syncthreads();
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const unsigned img = lds_base + (unsigned)((chunk & 1) * Cfg::IMG_BYTES);
        const bool more = chunk + 1 < a.nchunks;
        h8 bfr[2][8];      // [buffer][(row of the half-step) * 4 + (row half) * 2 + plane]
        auto load_b = [&](int buf, int hs) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const unsigned addr = img + (unsigned)(qoff[hs >> 1] + rowoff[2 * (hs & 1) + rr]);
                lds_read_b128_to<0>(bfr[buf][rr * 4 + 0], addr);
                lds_read_b128_to<Cfg::PLANE_BYTES>(bfr[buf][rr * 4 + 1], addr);
                lds_read_b128_to<256>(bfr[buf][rr * 4 + 2], addr);
                lds_read_b128_to<Cfg::PLANE_BYTES + 256>(bfr[buf][rr * 4 + 3], addr);
            }
        };
        load_b(0, 0);
        h8 af[4];
#pragma unroll
        for (int hs = 0; hs < 2 * NQ; ++hs) {
            const int cur = hs & 1, nx = cur ^ 1;
            if ((hs & 1) == 0) {
                if (hs == 2 * Cfg::FILL_STEP) {
                    // every weight fragment in flight lands first: none is waited for between here and the barrier
#pragma unroll
                    for (int i = 0; i < PF; ++i) asm volatile("" ::"v"(q_[i][0]), "v"(q_[i][1]), "v"(q_[i][2]), "v"(q_[i][3]));
                    if (more) issue(chunk + 1, (chunk + 1) & 1);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = q_[0][m];
#pragma unroll
                for (int i = 0; i + 1 < PF; ++i)
#pragma unroll
                    for (int m = 0; m < 4; ++m) q_[i][m] = q_[i + 1][m];
#pragma unroll
                for (int m = 0; m < 4; ++m) q_[PF - 1][m] = wq[m * 64];      // from k-step FILL_STEP on: fragments of the NEXT chunk
                wq += 4 * 64;
            }
            if (hs + 1 < 2 * NQ) {
                load_b(nx, hs + 1);
                wait_lgkm_for<8>(bfr[cur]);
            } else {
                wait_lgkm_for<0>(bfr[cur]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const h8 aw = term == 0 ? af[2 * h + 1] : af[2 * h];
                            const h8 bx = bfr[cur][rr * 4 + ph * 2 + (term == 1 ? 1 : 0)];
                            acc[2 * (hs & 1) + rr][ph][h] =
                                __builtin_amdgcn_mfma_f32_16x16x32_f16(aw, bx, acc[2 * (hs & 1) + rr][ph][h], 0, 0, 0);
                        }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();     // drains the DMA of the next chunk and retires every read of this chunk's buffer
    }

    const int out_hw = a.Hout * a.Wout;
    const int64_t out_dhw = (int64_t)out_hw * a.Dout;
    const bool relu = (a.flags & SNVC_EPI_RELU) != 0;
    const int c0 = cb * 32 + 8 * kb;
    float sc[8], bi[8], hw8[8];
    const float x_inv = (EPI == 2 && a.x_mul) ? 1.0f / a.x_mul[0] : 1.0f;      // see conv3d_f16_kernel's EPI 2
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = (a.scale ? a.scale[c0 + e] : 1.0f) * x_inv;
        bi[e] = a.scale ? a.bias[c0 + e] : 0.0f;
        hw8[e] = (EPI == 3) ? a.head[c0 + e] : 0.0f;
    }
    if constexpr (EPI == 2) {
        // the training step's layers (r6): lane (kb, col) stores its 8 channels of 16 consecutive voxels as float32 NCDHW (64-byte runs per
        // channel), adds the float32 residual (a skip connection's gradient), and leaves the statistics of what it stored -- per wave and
        // 32-channel group, the layout conv_stats_fold_kernel folds (slot = tile * 4 + wave)
        float *yf = a.y_f32 + n * a.yf_bs + (int64_t)c0 * out_dhw;
        const float *__restrict__ rf = a.res_f32 ? a.res_f32 + n * a.yf_bs + (int64_t)c0 * out_dhw : nullptr;
        const bool want_stats = a.stats != nullptr;
        float st_s[8], st_q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.0f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int row = wave * NB + nb;
            const int pd = od0 + row / TH, phh = oh0 + row % TH;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int pw = ow0 + 16 * ph + col;
                const bool ok = pd < a.nd && phh < a.nh && pw < a.nw;
                const int64_t sp = ok ? ((int64_t)pd * out_hw + phh * a.Wout + pw) : 0;
                float rfv[8];
                if (rf && ok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) rfv[e] = rf[(int64_t)e * out_dhw + sp];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = __builtin_fmaf(acc[nb][ph][e >> 2][e & 3], sc[e], bi[e]);
                    if (relu) v = __builtin_fmaxf(v, 0.0f);
                    const float sv = v * a.head_mul;
                    if (ok) {
                        yf[(int64_t)e * out_dhw + sp] = rf ? sv + rfv[e] : sv;
                        if (want_stats) {
                            st_s[e] += sv;
                            st_q[e] = __builtin_fmaf(sv, sv, st_q[e]);
                        }
                    }
                }
            }
        }
        if (want_stats) {
            const int groups = a.Cout >> 5;
            double *dst = a.stats + ((((int64_t)n * ntiles * 4 + (int64_t)t * 4 + wave) * groups + cb) * 32 + 8 * kb) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s0 = st_s[e], s1 = st_q[e];
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) {
                    s0 += __shfl_xor(s0, o, 64);
                    s1 += __shfl_xor(s1, o, 64);
                }
                if (col == 0) {
                    dst[e * 2] = (double)s0;
                    dst[e * 2 + 1] = (double)s1;
                }
            }
        }
        return;
    }
    _Float16 *yn = a.y + n * a.y_bs + (int64_t)(cb * 4 + kb) * out_dhw * 8;
    _Float16 *yn_lo = a.y_lo + n * a.y_bs + (int64_t)(cb * 4 + kb) * out_dhw * 8;
    constexpr float kHalfMax = 65504.0f;
    const float lo_bound = relu ? 0.0f : -kHalfMax;
    float vmax = 0.0f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row = wave * NB + nb;
        const int pd = od0 + row / TH, phh = oh0 + row % TH;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int pw = ow0 + 16 * ph + col;
            const bool ok = pd < a.nd && phh < a.nh && pw < a.nw;
            const int64_t sp = ok ? ((int64_t)pd * out_hw + phh * a.Wout + pw) : 0;
            h8 o, ol;
            float hsum = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = __builtin_fmaf(acc[nb][ph][e >> 2][e & 3], sc[e], bi[e]);
                v = __builtin_amdgcn_fmed3f(v, lo_bound, kHalfMax);
                vmax = __builtin_fmaxf(vmax, ok ? __builtin_fabsf(v) : 0.0f);      // what is stored (after the ReLU), inside the tensor
                if constexpr (EPI == 3) hsum += hw8[e] * v;
                o[e] = (_Float16)v;
                ol[e] = (_Float16)(v - (float)o[e]);
            }
            if (ok) {
                *reinterpret_cast<h8 *>(yn + sp * 8) = o;
                *reinterpret_cast<h8 *>(yn_lo + sp * 8) = ol;
            }
            if constexpr (EPI == 3) {
                float tot = hsum + __shfl_xor(hsum, 16, 64);
                tot += __shfl_xor(tot, 32, 64);
                if (ok && kb == 0) a.y_head[n * out_dhw + sp] = tot * a.head_mul;
            }
        }
    }
    if (vmax >= kHalfMax && a.overflow) atomicOr(a.overflow, 1);
}

// ------------------------------------------------------------------------- depth-1 (2D) layers in split mode (r5)
// The sheared first layer's prep -- G = a depth-1 3x7 convolution of the upsampled right feature, 32 -> 96 channels, and its small
// relatives (G', the left half's 3x3 depth-class planes) -- ran on the fp32 matrix pipe: 7.8 GFLOP at 93 TFLOP/s = 84 us of a
// 2.1 ms step whose other layers had all moved to the 16x faster half pipe.  Same scheme as conv3d_x3q_kernel (three
// v_mfma_f32_16x16x32_f16 per fp32 product on (hi, lo) pairs, K = 32 = four taps of one C8 piece, image double-buffered and refilled
// late under hand-counted waits), on a 2D tile: 16 rows x 32 columns x 32 output channels per workgroup, taps in (kh, kw) raster order
// (3x7: 21 taps in 6 quads; 3x3: 9 in 3).  The result leaves as plain fp32 [N][Cout][H][W] times `head_mul` (the expand pass that
// consumes G applies the layer's affine itself); the input's scale is read from the device (`x_mul`: the feature's own maximum).
template <int KH_, int KW_>
struct X2QCfg {
    static constexpr int KH = KH_, KW = KW_, TH = 16, NB = 4, IN_H = TH + KH - 1, IN_W = 32 + KW - 1, VOX = IN_H * IN_W, GB = VOX * 16;
    static constexpr int ITEMS = 2 * VOX, PLANE_BYTES = GB, NIT = (ITEMS + 255) / 256;
    static constexpr int IMG_BYTES = NIT * 256 * 16, LDS_BYTES = 2 * IMG_BYTES;
    static constexpr int TAPS = KH * KW, NQ = (TAPS + 3) / 4, PF = 2, FILL_STEP = NQ - PF;
    static_assert(NQ > PF, "the refill is issued PF k-steps before the chunk ends");
};

template <class Cfg>
__global__ void __launch_bounds__(256, 2)
conv2d_x3q_kernel(const F16Args a) {
    constexpr int NB = Cfg::NB, IN_W = Cfg::IN_W, VOX = Cfg::VOX, NIT = Cfg::NIT, NQ = Cfg::NQ;
    constexpr int ITEMS = Cfg::ITEMS, PF = Cfg::PF;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kb = lane >> 4, col = lane & 15;
    const int ntiles = a.tiles_h * a.tiles_w, cblocks = a.Cout >> 5;
    const int job = xcd_remap16(blockIdx.x, ntiles * cblocks);
    const int t = job / cblocks, cb = job - t * cblocks;
    const int tw = t % a.tiles_w, th = t / a.tiles_w;
    const int64_t n = blockIdx.z;
    const int oh0 = th * Cfg::TH, ow0 = tw * 32;
    const int ih0 = oh0 - a.pad_h, iw0 = ow0 - a.pad_w;

    f32x4q acc[NB][2][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[nb][ph][h] = f32x4q{0.0f, 0.0f, 0.0f, 0.0f};

    const int in_hw = a.Hin * a.Win;
    unsigned off[NIT];
    unsigned vmask = 0, pmask = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int plane = i / VOX, r = i - plane * VOX;
        pmask |= (unsigned)(plane & 1) << it;
        const int hh = r / IN_W, ww = r - hh * IN_W;
        const int gh = ih0 + hh, gw = iw0 + ww;
        const bool ok = i < ITEMS && (unsigned)gh < (unsigned)a.Hin && (unsigned)gw < (unsigned)a.Win;
        off[it] = ok ? (unsigned)(gh * a.Win + gw) : 0u;
        vmask |= (ok ? 1u : 0u) << it;
    }
    const _Float16 *xn = a.x + n * a.x_bs, *xn_lo = a.x_lo + n * a.x_bs;
    const int wbase = tid & ~63;
    auto issue = [&](int chunk, int buf) {
        const int64_t coff = (int64_t)chunk * in_hw * 8;
        char *const ibuf = lds + buf * Cfg::IMG_BYTES;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const _Float16 *xc = (((pmask >> it) & 1u) ? xn_lo : xn) + coff;
            const void *src = ((vmask >> it) & 1u) ? static_cast<const void *>(xc + (size_t)off[it] * 8) : static_cast<const void *>(g_zero16h);
            __builtin_amdgcn_global_load_lds(static_cast<const float *>(src), reinterpret_cast<float *>(ibuf + (it * 256 + wbase) * 16), 16, 0, 0);
        }
    };
    int qoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int tp = 4 * q + kb;
        tp = tp < Cfg::TAPS ? tp : Cfg::TAPS - 1;       // a slot beyond the last tap: zero weights, the last tap's piece
        qoff[q] = ((tp / Cfg::KW) * IN_W + tp % Cfg::KW + col) * 16;
    }
    int rowoff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) rowoff[nb] = (wave * NB + nb) * IN_W * 16;
    const int64_t steps_total = (int64_t)a.nchunks * NQ;
    const h8 *wq = reinterpret_cast<const h8 *>(a.wp) + ((int64_t)cb * steps_total * 4) * 64 + lane;
    h8 q_[PF][4];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) q_[i][m] = wq[m * 64];
        wq += 4 * 64;
    }

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    issue(0, 0);
    __syncthreads();
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const unsigned img = lds_base + (unsigned)((chunk & 1) * Cfg::IMG_BYTES);
        const bool more = chunk + 1 < a.nchunks;
        h8 bfr[2][8];      // [buffer][(row of the half-step) * 4 + (row half) * 2 + plane]
        auto load_b = [&](int buf, int hs) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const unsigned addr = img + (unsigned)(qoff[hs >> 1] + rowoff[2 * (hs & 1) + rr]);
                lds_read_b128_to<0>(bfr[buf][rr * 4 + 0], addr);
                lds_read_b128_to<Cfg::PLANE_BYTES>(bfr[buf][rr * 4 + 1], addr);
                lds_read_b128_to<256>(bfr[buf][rr * 4 + 2], addr);
                lds_read_b128_to<Cfg::PLANE_BYTES + 256>(bfr[buf][rr * 4 + 3], addr);
            }
        };
        load_b(0, 0);
        h8 af[4];
#pragma unroll
        for (int hs = 0; hs < 2 * NQ; ++hs) {
            const int cur = hs & 1, nx = cur ^ 1;
            if ((hs & 1) == 0) {
                if (hs == 2 * Cfg::FILL_STEP) {
                    // every weight fragment in flight lands first: none is waited for between here and the barrier
#pragma unroll
                    for (int i = 0; i < PF; ++i) asm volatile("" ::"v"(q_[i][0]), "v"(q_[i][1]), "v"(q_[i][2]), "v"(q_[i][3]));
                    if (more) issue(chunk + 1, (chunk + 1) & 1);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = q_[0][m];
#pragma unroll
                for (int i = 0; i + 1 < PF; ++i)
#pragma unroll
                    for (int m = 0; m < 4; ++m) q_[i][m] = q_[i + 1][m];
#pragma unroll
                for (int m = 0; m < 4; ++m) q_[PF - 1][m] = wq[m * 64];      // from k-step FILL_STEP on: fragments of the NEXT chunk
                wq += 4 * 64;
            }
            if (hs + 1 < 2 * NQ) {
                load_b(nx, hs + 1);
                wait_lgkm_for<8>(bfr[cur]);
            } else {
                wait_lgkm_for<0>(bfr[cur]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const h8 aw = term == 0 ? af[2 * h + 1] : af[2 * h];
                            const h8 bx = bfr[cur][rr * 4 + ph * 2 + (term == 1 ? 1 : 0)];
                            acc[2 * (hs & 1) + rr][ph][h] =
                                __builtin_amdgcn_mfma_f32_16x16x32_f16(aw, bx, acc[2 * (hs & 1) + rr][ph][h], 0, 0, 0);
                        }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();     // drains the DMA of the next chunk and retires every read of this chunk's buffer
    }

    // epilogue: lane (kb, col) holds channels cb * 32 + 8 kb + e of columns col / col + 16 of its four rows; fp32 [N][Cout][H][W]
    const int64_t out_hw = (int64_t)a.Hout * a.Wout;
    const bool relu = (a.flags & SNVC_EPI_RELU) != 0;
    const int c0 = cb * 32 + 8 * kb;
    const float xm = a.x_mul ? a.head_mul / a.x_mul[0] : a.head_mul;
    float sc[8], bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = (a.scale ? a.scale[c0 + e] : 1.0f) * xm;
        bi[e] = a.scale ? a.bias[c0 + e] : 0.0f;
    }
    float *yn = a.y_f32 + n * a.yf_bs + (int64_t)c0 * out_hw;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int phh = oh0 + wave * NB + nb;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int pw = ow0 + 16 * ph + col;
            if (phh < a.nh && pw < a.nw) {
                float *yp = yn + (int64_t)phh * a.Wout + pw;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = __builtin_fmaf(acc[nb][ph][e >> 2][e & 3], sc[e], bi[e]);
                    if (relu) v = __builtin_fmaxf(v, 0.0f);
                    yp[e * out_hw] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------- 16x16x32 form of the stride-2 3x3x3 split layer (r5)
// The hourglass's stride-2 layers (snvc/models/submodule.py:88-94,170-181) were the layers furthest from any roof: the 2x4x32 output tile
// of conv3d_f16_kernel<F16K3S2X> needs a 5 x 9 x 65-piece image PER PLANE (46.8 KB), so the planes were taken one after the other,
// single-buffered at two workgroups per CU: every pass a fill with nothing under it but the other workgroup, the weights of a
// chunk streamed twice (hi pass, lo pass), 14 k-steps of 32x32x16 MFMAs per fill.  hg conv1 at cfg2: 1.16 GB moved at 3.4 TB/s.
// This form: ONE PERSISTENT workgroup of 8 waves per CU with THREE image slots (144 KB of the CU's 160 KB LDS),
// v_mfma_f32_16x16x32_f16 (K = 32 = four taps of one C8 piece, as conv3d_x3q_kernel), 64 output channels per workgroup: wave =
// (row pair of the 2x4 output rows) x (32-channel half), so a wave holds only ITS half's weights -- a step's w_hi and w_lo, 2 x 56
// VGPRs: MFMA operands cannot sit in AGPRs, and vmcnt retires in order: a weight fetched behind an LDS-DMA would wait for it, so a
// step's weights are all in registers before its refills are requested.  A step (one channel group of one tile, both planes):
//   the NEXT step's hi plane is requested into the spare slot;
//   phase L: acc += w_hi * x_lo   from the lo slot, 7 tap quads x 8 MFMAs per wave;
//   barrier (LDS reads only); the next step's lo plane is requested into the lo slot just released;
//   phase H: acc += w_lo * x_hi + w_hi * x_hi   from the hi slot, 7 quads x 16 MFMAs; as each quad retires its weight registers the
//            next step's fragments are requested into them (first read behind the barrier below);
//   barrier (drains everything); the hi slot becomes the spare one.
// Steps run on across tile boundaries (the next tile's first planes are requested under this tile's last step, the epilogue's
// stores drain under the next tile's MFMAs): with one workgroup per CU nothing else would hide a workgroup's first fill, last
// wait and stores -- one workgroup per tile measured 397 us on hg conv1, of which 257 remained with MFMAs AND refills off.
// Measured (tools/time_hg.py, back to back): hg conv1 395-407 -> 326 us, hg conv3 93 -> 85 us.  What still bounds it is the
// refill itself: 1.08 GB through 94 KB-per-CU bursts reach 3.3 TB/s (a deeper queue needs a fourth slot the LDS does not have).
// The image rows are stored POLYPHASE in blocks of 32 columns -- 16 even columns, then the 16 odd ones -- so that the 16 output columns of
// a fragment read consecutive pieces for every tap (column 2c + kw: kw = 0 / 2 -> even piece c / c + 1, kw = 1 -> odd piece c):
// conflict-free ds_read_b128 where the natural order reads at a 32-byte stride, while 32 consecutive LDS positions are still 32
// consecutive columns, i.e. a DMA round fetches whole 128-byte lines.  Same values as the 32x32x16 form up to fp32 summation order.
struct X3S2Cfg {
    static constexpr int TD = 2, TH = 4, NB = 2, IN_D = 5, IN_H = 9, IN_W = 65, VOX = IN_D * IN_H * IN_W;
    static constexpr int THREADS = 512, NIT = (VOX + THREADS - 1) / THREADS, SLOT_BYTES = NIT * THREADS * 16, LDS_BYTES = 3 * SLOT_BYTES;
    static constexpr int NQ = 7;
};

template <int N>
__device__ __forceinline__ void wait_lgkm_for4(h8 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}

__global__ void __launch_bounds__(512, 1)
conv3d_x3s2q_kernel(const F16Args a, const int total_jobs) {
    using Cfg = X3S2Cfg;
    constexpr int NB = Cfg::NB, TH = Cfg::TH, IN_H = Cfg::IN_H, IN_W = Cfg::IN_W, VOX = Cfg::VOX, NIT = Cfg::NIT, NQ = Cfg::NQ;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // 8 waves: wave = (row pair 0..3) + 4 * (32-channel half of the 64 output channels).  Two waves per SIMD: each holds the weights of ITS
    // 32 channels only (2 x 56 VGPRs for a step's w_hi and w_lo -- MFMA operands cannot sit in AGPRs, and all 64 channels' 224 do not
    // fit beside the B fragments), every B fragment is read by the two waves that share its rows
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, chalf = tid >> 8, kb = lane >> 4, col = lane & 15;
    // PERSISTENT: one workgroup per CU walks the jobs b, b + G, b + 2 G, ... (job = (sample, tile, 64-channel block); G a multiple of 8,
    // so a workgroup stays inside its XCD's contiguous range of tiles).  With one workgroup per CU nothing else hides a workgroup's
    // start-up (first fill: a whole HBM round trip), its last wait and its stores -- measured with the MFMAs AND the refills
    // switched off, one workgroup per tile still took 257 of 397 us on hg conv1.  Here the chunk pipeline simply runs on across
    // tile boundaries: the next tile's first planes are requested under the last phase H of this one, the epilogue's stores drain
    // under the next tile's MFMAs.
    const int ntiles = a.tiles_d * a.tiles_h * a.tiles_w, cblocks = a.Cout >> 6;
    const int in_hw = a.Hin * a.Win, in_dhw = in_hw * a.Din;
    const int wbase = tid & ~63;

    struct Job { int od0, oh0, ow0, cb; int64_t n; };
    auto decode = [&](int b) {
        const int j = xcd_remap16(b, total_jobs);
        const int cb = j % cblocks, r = j / cblocks, t = r % ntiles;
        Job jb;
        jb.n = r / ntiles; jb.cb = cb;
        const int tw = t % a.tiles_w, th = (t / a.tiles_w) % a.tiles_h, td = t / (a.tiles_w * a.tiles_h);
        jb.od0 = td * Cfg::TD; jb.oh0 = th * TH; jb.ow0 = tw * 32;
        return jb;
    };
    // staging geometry of a tile: piece i of a slot = (dd, hh, position in the polyphase row)
    unsigned off[NIT];
    unsigned vmask = 0;
    auto geometry = [&](const Job &jb) {
        const int id0 = 2 * jb.od0 - a.pad_d, ih0 = 2 * jb.oh0 - a.pad_h, iw0 = 2 * jb.ow0 - a.pad_w;
        vmask = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * 512 + tid;
            const int dd = i / (IN_H * IN_W), r2 = i - dd * (IN_H * IN_W);
            const int hh = r2 / IN_W, pp = r2 - hh * IN_W;
            // row layout: [16 even columns][16 odd columns] twice, then column 64 -- 32 consecutive positions are 32 consecutive
            // columns (permuted): a DMA round still fetches whole 128-byte lines
            const int blk = pp >> 5, rr_ = pp & 31;
            const int ww = pp == 64 ? 64 : (rr_ < 16 ? 2 * (16 * blk + rr_) : 2 * (16 * blk + rr_ - 16) + 1);
            const int gd = id0 + dd, gh = ih0 + hh, gw = iw0 + ww;
            const bool ok = i < VOX && (unsigned)gd < (unsigned)a.Din && (unsigned)gh < (unsigned)a.Hin && (unsigned)gw < (unsigned)a.Win;
            off[it] = ok ? (unsigned)(gd * in_hw + gh * a.Win + gw) : 0u;
            vmask |= (ok ? 1u : 0u) << it;
        }
    };
    auto issue = [&](int64_t n, int chunk, bool lo_plane, int slot) {      // every wave issues all NIT rounds (pieces past the image land in the slot's padding)
        const _Float16 *xc = (lo_plane ? a.x_lo : a.x) + n * a.x_bs + (int64_t)chunk * in_dhw * 8;
        char *const ibuf = lds + slot * Cfg::SLOT_BYTES;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const void *src = ((vmask >> it) & 1u) ? static_cast<const void *>(xc + (size_t)off[it] * 8) : static_cast<const void *>(g_zero16h);
            __builtin_amdgcn_global_load_lds(static_cast<const float *>(src), reinterpret_cast<float *>(ibuf + (it * 512 + wbase) * 16), 16, 0, 0);
        }
    };
    // this lane's byte offset of tap 4 q + kb inside a slot (output column `col` of the tile's left half; + 256 B: the right half)
    int qoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int tp = 4 * q + kb;
        tp = tp < 27 ? tp : 26;                     // the 28th slot: zero weights, tap 26's piece
        const int kd = tp / 9, kh = (tp / 3) % 3, kw = tp % 3;
        // column 2 c + kw of output column c = col (+ 16 for the right half: + 32 positions): kw = 0 -> even piece c, kw = 1 -> odd piece c,
        // kw = 2 -> even piece c + 1 (for c = 15 the first piece of the next block: position 32)
        const int pos = kw == 0 ? col : (kw == 1 ? 16 + col : (col == 15 ? 32 : col + 1));
        qoff[q] = ((kd * IN_H + kh) * IN_W + pos) * 16;
    }
    int rowoff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row = wave * NB + nb;
        rowoff[nb] = ((2 * (row / TH)) * IN_H + 2 * (row % TH)) * IN_W * 16;
    }
    // weights: [cb][chunk][quad][co half (4)][hi | lo][lane] pieces (pack_q16s_weights_kernel with NH = 4)
    // (the pointer stays wave-uniform and the lane is added in the index: the loads take the scalar-base form, one VGPR of offset
    // for all 56 fragments of a step instead of a 64-bit address pair each)
    const h8 *const wall = reinterpret_cast<const h8 *>(a.wp);
    const int64_t wchunk = (int64_t)NQ * 8 * 64;     // pieces per (cb, chunk)

    int b = blockIdx.x;
    Job cur = decode(b);
    geometry(cur);
    h8 WH[NQ][2], WL[NQ][2];
    {
        const h8 *wc = wall + (int64_t)cur.cb * a.nchunks * wchunk;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                WH[q][h] = wc[(q * 8 + (2 * chalf + h) * 2) * 64 + lane];
                WL[q][h] = wc[(q * 8 + (2 * chalf + h) * 2 + 1) * 64 + lane];
            }
    }
    f32x4q acc[NB][2][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[nb][ph][h] = f32x4q{0.0f, 0.0f, 0.0f, 0.0f};

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    int s_hi = 0, s_sp = 2;                          // slots: hi plane, (1 = lo plane, fixed), spare
    issue(cur.n, 0, false, 0);
    issue(cur.n, 0, true, 1);
    __syncthreads();
    const int gstride = gridDim.x;
    constexpr float kHalfMax = 65504.0f;
    float vmax = 0.0f;
    int chunk = 0;
    while (true) {
        const bool last_chunk = chunk + 1 == a.nchunks;
        const bool next_job = last_chunk && b + gstride < total_jobs;
        const bool more = !last_chunk || next_job;
        // the next step's hi plane is requested FIRST (the spare slot is free): with the lo plane following behind phase L, a request
        // is outstanding for all but the few hundred cycles around the two barriers (requested only behind phase L, every CU asked
        // for its 94 KB at the same moment and then waited: hg conv1 moved 3.2 TB/s)
        Job nxt = cur;
        if (next_job) {                              // block-uniform: the tile behind this one (all of this tile's planes are requested already)
            nxt = decode(b + gstride);
            geometry(nxt);
        }
        const int nchunk = last_chunk ? 0 : chunk + 1;
        if (more) issue(nxt.n, nchunk, false, s_sp);
        h8 bfr[2][4];      // [buffer][row * 2 + row half]
        auto load_b = [&](int buf, unsigned img, int q) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const unsigned addr = img + (unsigned)(qoff[q] + rowoff[rr]);
                lds_read_b128_to<0>(bfr[buf][rr * 2 + 0], addr);
                lds_read_b128_to<512>(bfr[buf][rr * 2 + 1], addr);
            }
        };
        // ---- phase L: w_hi * x_lo
        {
            const unsigned img = lds_base + (unsigned)Cfg::SLOT_BYTES;
            load_b(0, img, 0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int cb_ = q & 1, nx = cb_ ^ 1;
                if (q + 1 < NQ) {
                    load_b(nx, img, q + 1);
                    wait_lgkm_for4<4>(bfr[cb_]);
                } else {
                    wait_lgkm_for4<0>(bfr[cb_]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            acc[rr][ph][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WH[q][h], bfr[cb_][rr * 2 + ph], acc[rr][ph][h], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the lo slot is released by every wave (a bare s_barrier: no vector-memory wait -- the hi plane's refill stays in flight; this
        // chunk's weights landed before the barrier that ended the previous step)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (more) issue(nxt.n, nchunk, true, 1);
        // ---- phase H: (w_lo + w_hi) * x_hi
        {
            const h8 *wn = wall + ((int64_t)nxt.cb * a.nchunks + nchunk) * wchunk;
            const unsigned img = lds_base + (unsigned)(s_hi * Cfg::SLOT_BYTES);
            load_b(0, img, 0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int cb_ = q & 1, nx = cb_ ^ 1;
                if (q + 1 < NQ) {
                    load_b(nx, img, q + 1);
                    wait_lgkm_for4<4>(bfr[cb_]);
                } else {
                    wait_lgkm_for4<0>(bfr[cb_]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int term = 0; term < 2; ++term)
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                                acc[rr][ph][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(term == 0 ? WL[q][h] : WH[q][h], bfr[cb_][rr * 2 + ph],
                                                                                        acc[rr][ph][h], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // the next step's weights of this quad, into the registers the quad has just retired (first read behind the barrier below)
                if (more) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        WH[q][h] = wn[(q * 8 + (2 * chalf + h) * 2) * 64 + lane];
                        WL[q][h] = wn[(q * 8 + (2 * chalf + h) * 2 + 1) * 64 + lane];
                    }
                }
            }
        }
        __syncthreads();                             // drains the refill and retires every read of the hi slot
        { const int o = s_hi; s_hi = s_sp; s_sp = o; }
        if (!last_chunk) { ++chunk; continue; }

        // ---- epilogue of the finished tile: lane (kb, col) holds, per 32-channel block bb, the 8 channels of C8 group 4 bb + kb for
        // voxel col (+16) of its rows.  Its stores drain under the next tile's first phase.
        {
            const int out_hw = a.Hout * a.Wout;
            const int64_t out_dhw = (int64_t)out_hw * a.Dout;
            const bool relu = (a.flags & SNVC_EPI_RELU) != 0;
            const float lo_bound = relu ? 0.0f : -kHalfMax;
            {
                const int bb = chalf;
                const int c0 = cur.cb * 64 + bb * 32 + 8 * kb;
                float sc[8], bi[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sc[e] = a.scale ? a.scale[c0 + e] : 1.0f;
                    bi[e] = a.scale ? a.bias[c0 + e] : 0.0f;
                }
                const int64_t gplane = (int64_t)(cur.cb * 8 + bb * 4 + kb) * out_dhw * 8;
                _Float16 *yn = a.y + cur.n * a.y_bs + gplane, *yn_lo = a.y_lo + cur.n * a.y_bs + gplane;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int row = wave * NB + nb;
                    const int pd = cur.od0 + row / TH, phh = cur.oh0 + row % TH;
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
                        const int pw = cur.ow0 + 16 * ph + col;
                        const bool ok = pd < a.nd && phh < a.nh && pw < a.nw;
                        const int64_t sp = ok ? ((int64_t)pd * out_hw + phh * a.Wout + pw) : 0;
                        h8 o, ol;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float v = __builtin_fmaf(acc[nb][ph][e >> 2][e & 3], sc[e], bi[e]);
                            v = __builtin_amdgcn_fmed3f(v, lo_bound, kHalfMax);
                            vmax = __builtin_fmaxf(vmax, ok ? __builtin_fabsf(v) : 0.0f);
                            o[e] = (_Float16)v;
                            ol[e] = (_Float16)(v - (float)o[e]);
                            acc[nb][ph][e >> 2][e & 3] = 0.0f;
                        }
                        if (ok) {
                            *reinterpret_cast<h8 *>(yn + sp * 8) = o;
                            *reinterpret_cast<h8 *>(yn_lo + sp * 8) = ol;
                        }
                    }
                }
            }
        }
        if (!next_job) break;                        // block-uniform: every wave of the workgroup leaves here
        b += gstride;
        cur = nxt;
        chunk = 0;
    }
    if (vmax >= kHalfMax && a.overflow) atomicOr(a.overflow, 1);
}

// 16x16x32 form of the 5^3 / dilated 5^3 / 7^3 layers (split mode with the planes serial, one channel group per chunk, as F16K5X /
// F16K5D2X / F16K7X; and the fp16-storage family): K = 32 = FOUR TAPS of one C8 piece -- k-block kb = lane >> 4 is tap 4 q + kb of
// the list of ALL K^3 taps in (kd, kh, kw) raster order, cut into equal runtime-looped segments (a segment is what gets unrolled):
// 5^3: one segment of 32 quads for 125 taps (2 % padding; slice by slice -- 5 x 7 quads for 5 x 25 taps, the first form -- the
// plain 5^3 layer gained nothing over the 13 tap pairs of the 32x32x16 form), 7^3: two segments of 43 quads for 343 taps (0.3 %
// instead of 6 %).  A slot beyond the last tap has zero weights and re-reads the last tap's piece.  A lane's byte offset per quad
// comes from a table in LDS behind the image, fetched one quad ahead (a register table of 32 or 43 does not fit; computed per quad
// -- two divisions -- it ate the gain in the forms with 16 MFMAs per half-step).  Measured against the slice-by-slice form: 7^3
// -2..3 %, dilated 5^3 -6..10 %, 5^3 -18..22 % against its 32x32x16 form.  Same tile, image, staging, sub-grid classes and epilogue
// contract as the 32x32x16 forms; residual (before / after the activation) supported; C8 output only.
// flat segments of the 16x16x32 forms (Q16SCfg::FSEG): 5^3 and dilated 5^3 one (125 taps = 32 quads), 7^3 two (343 taps = 2 x 43 quads)
constexpr int q16s_flat_segments(int ks) { return ks == 7 ? 2 : 1; }
constexpr int q16s_steps(int ks) {      // k-steps per pass
    const int f = q16s_flat_segments(ks), k3 = ks * ks * ks;
    return f * (((k3 + f - 1) / f + 3) / 4);
}

template <int KS_, int DILW_, int PLQ_ = 3, int NHB_ = 1>
struct Q16SCfg {
    // PLQ = 3: split mode, planes serial (a hi pass: lo_w * hi_x + hi_w * hi_x; a lo pass: hi_w * lo_x); PLQ = 1: the fp16-STORAGE family
    // (one plane, one MFMA per product, C8 half output).  NHB: 32-channel blocks per workgroup (2: every B fragment feeds two
    // blocks -- the fp16-storage form needs that: with one block its LDS reads alone saturate the LDS at the MFMA rate)
    static constexpr int KS = KS_, DILW = DILW_, PLQ = PLQ_, NHB = NHB_, TD = 4, TH = 4, NB = 4;
    static constexpr bool SPLIT = PLQ_ == 3;
    static constexpr int NH = 2 * NHB, MA = NH * (SPLIT ? 2 : 1), PASSES = SPLIT ? 2 : 1;
    static constexpr int IN_D = TD + KS - 1, IN_H = TH + KS - 1, IN_W = 32 + (KS - 1) * DILW, VOX = IN_D * IN_H * IN_W;
    static constexpr int ITEMS = VOX, NIT = (ITEMS + 255) / 256, IMG_BYTES = NIT * 256 * 16;
    static constexpr int K3 = KS * KS * KS, KSEG = q16s_flat_segments(KS);      // segments of the flat tap list
    static constexpr int NQ = ((K3 + KSEG - 1) / KSEG + 3) / 4, NT = 4 * NQ, PF = 2;      // quads / tap slots per segment
    static constexpr int STEPS = KSEG * NQ;                 // k-steps per pass
    static_assert(STEPS == q16s_steps(KS) && KSEG * NT >= K3 && (KSEG - 1) * NT < K3, "the segments cover the taps");
    static constexpr int TAB_BYTES = KSEG * NT * 4;         // byte offset of every tap slot's piece inside the image
    static constexpr int LDS_BYTES = IMG_BYTES + TAB_BYTES;
    static_assert(NIT <= 32, "validity mask is one register");
    static_assert(PLQ_ == 3 || PLQ_ == 1, "planes serial (split) or one plane (fp16 storage)");
};

template <class Cfg>
__global__ void __launch_bounds__(256, 2)
conv3d_q16s_kernel(const F16Args a_) {
    F16Args a = a_;
    const int cls = blockIdx.z / a.N;
    if (a.cls_mode == 2) {
        const int pd = (cls >> 1) & 1, ph = cls & 1;
        a.offd = a.iod = pd; a.offh = a.ioh = ph;
        a.nd = (a.Dout - pd + 1) / 2; a.nh = (a.Hout - ph + 1) / 2;
    }
    constexpr int NB = Cfg::NB, TH = Cfg::TH, IN_H = Cfg::IN_H, IN_W = Cfg::IN_W, NIT = Cfg::NIT, NQ = Cfg::NQ;
    constexpr int ITEMS = Cfg::ITEMS, PF = Cfg::PF, KS = Cfg::KS, NH = Cfg::NH, MA = Cfg::MA;
    constexpr bool SPLIT = Cfg::SPLIT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kb = lane >> 4, col = lane & 15;
    const int ntiles = a.tiles_d * a.tiles_h * a.tiles_w;
    const int t = xcd_remap16(blockIdx.x, ntiles);
    const int tw = t % a.tiles_w, th = (t / a.tiles_w) % a.tiles_h, td = t / (a.tiles_w * a.tiles_h);
    const int cb = blockIdx.y;
    const int64_t n = blockIdx.z - cls * a.N;
    const int od0 = td * Cfg::TD, oh0 = th * TH, ow0 = tw * 32;
    if (od0 >= a.nd || oh0 >= a.nh) return;      // a smaller class of an odd extent: whole tile outside (block-uniform)
    const int id0 = od0 - a.pad_d, ih0 = oh0 - a.pad_h, iw0 = ow0 - a.pad_w;

    f32x4q acc[NB][2][NH];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int h = 0; h < NH; ++h) acc[nb][ph][h] = f32x4q{0.0f, 0.0f, 0.0f, 0.0f};

    const int in_hw = a.Hin * a.Win, in_dhw = in_hw * a.Din;
    unsigned off[NIT];
    unsigned vmask = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int dd = i / (IN_H * IN_W), r2 = i - dd * (IN_H * IN_W);
        const int hh = r2 / IN_W, ww = r2 - hh * IN_W;
        const int gd = (id0 + dd) * a.isd + a.iod, gh = (ih0 + hh) * a.ish + a.ioh, gw = iw0 + ww;
        const bool ok = i < ITEMS && (unsigned)gd < (unsigned)a.Din && (unsigned)gh < (unsigned)a.Hin && (unsigned)gw < (unsigned)a.Win;
        off[it] = ok ? (unsigned)(gd * in_hw + gh * a.Win + gw) : 0u;
        vmask |= (ok ? 1u : 0u) << it;
    }
    const _Float16 *xn = a.x + n * a.x_bs, *xn_lo = SPLIT ? a.x_lo + n * a.x_bs : nullptr;
    const int wbase = tid & ~63;
    auto issue = [&](int pass) {                    // split: pass = chunk * 2 + plane; fp16 storage: pass = chunk
        const int chunk = SPLIT ? pass >> 1 : pass;
        const _Float16 *xc = ((SPLIT && (pass & 1)) ? xn_lo : xn) + (int64_t)chunk * in_dhw * 8;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * 256 + tid;
            const void *src = ((vmask >> it) & 1u) ? static_cast<const void *>(xc + (size_t)off[it] * 8) : static_cast<const void *>(g_zero16h);
            if (ITEMS % 256 == 0 || i < ITEMS)
                __builtin_amdgcn_global_load_lds(static_cast<const float *>(src), reinterpret_cast<float *>(lds + (it * 256 + wbase) * 16), 16, 0, 0);
        }
    };
    // this lane's byte offset inside the image for every quad of taps: the LDS table (written here, visible behind the first pass's
    // barrier), two quads of it in registers at a time
    int qo2[2] = {0, 0};
    const int *const qlane = reinterpret_cast<const int *>(lds + Cfg::IMG_BYTES) + kb;
    {
        int *const qtab = reinterpret_cast<int *>(lds + Cfg::IMG_BYTES);
        for (int i = tid; i < Cfg::KSEG * Cfg::NT; i += 256) {
            const int tp = i < Cfg::K3 ? i : Cfg::K3 - 1;          // a slot beyond the last tap: zero weights, the last tap's piece
            const int kd = tp / (KS * KS), r = tp - kd * (KS * KS), kh = r / KS, kw = r - kh * KS;
            qtab[i] = ((kd * IN_H + kh) * IN_W + kw * Cfg::DILW) * 16;
        }
    }
    const int col16 = col * 16;
    int rowoff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row = wave * NB + nb;
        rowoff[nb] = ((row / TH) * IN_H + (row % TH)) * IN_W * 16;
    }
    // A fragments: [cout block][chunk][pass][segment][quad][co half][hi | lo (split)][lane] pieces, in consumption order
    const int npass = a.nchunks * Cfg::PASSES;
    const int64_t steps_total = (int64_t)npass * Cfg::STEPS;
    const h8 *wq = reinterpret_cast<const h8 *>(a.wp) + ((int64_t)cb * steps_total * MA) * 64 + lane;
    h8 q_[PF][MA];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
#pragma unroll
        for (int m = 0; m < MA; ++m) q_[i][m] = wq[m * 64];
        wq += MA * 64;
    }

    for (int ps = 0; ps < npass; ++ps) {
        issue(ps);
        __syncthreads();
        const bool lo_pass = SPLIT && (ps & 1);
        // half-step hs = 2 s + j: rows 2 j, 2 j + 1 of k-step s = (segment, quad); its 4 B fragments are fetched one half-step ahead
        h8 bfr[2][4];
        auto load_b = [&](int buf, int qo, int j) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
                    bfr[buf][rr * 2 + ph] = *reinterpret_cast<const h8 *>(lds + qo + rowoff[2 * j + rr] + ph * 256);
        };
        qo2[0] = qlane[0] + col16;
        load_b(0, qo2[0], 0);
        h8 af[MA];
#pragma unroll 1
        for (int seg = 0; seg < Cfg::KSEG; ++seg) {
            const int *const qseg = qlane + seg * Cfg::NT;
#pragma unroll
            for (int hs = 0; hs < 2 * NQ; ++hs) {
                const int cur = hs & 1, nxt = cur ^ 1, j = hs & 1;
                if (j == 0) {
#pragma unroll
                    for (int m = 0; m < MA; ++m) af[m] = q_[0][m];
#pragma unroll
                    for (int i = 0; i + 1 < PF; ++i)
#pragma unroll
                        for (int m = 0; m < MA; ++m) q_[i][m] = q_[i + 1][m];
#pragma unroll
                    for (int m = 0; m < MA; ++m) q_[PF - 1][m] = wq[m * 64];
                    wq += MA * 64;
                    // the next quad's offset (behind the segment's last quad: the next segment's first)
                    const int qn = (hs >> 1) + 1;      // a constant after unrolling
                    if (qn < NQ) qo2[qn & 1] = qseg[4 * qn] + col16;
                    else if (seg + 1 < Cfg::KSEG) qo2[qn & 1] = qseg[Cfg::NT] + col16;
                }
                if (hs + 1 < 2 * NQ) load_b(nxt, qo2[((hs + 1) >> 1) & 1], (hs + 1) & 1);
                else if (seg + 1 < Cfg::KSEG) load_b(nxt, qo2[NQ & 1], 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (SPLIT) {
                    if (!lo_pass) {
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                                for (int h = 0; h < NH; ++h)
                                    acc[2 * j + rr][ph][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2 * h + 1], bfr[cur][rr * 2 + ph], acc[2 * j + rr][ph][h], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                        for (int h = 0; h < NH; ++h)
                            acc[2 * j + rr][ph][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[SPLIT ? 2 * h : h], bfr[cur][rr * 2 + ph], acc[2 * j + rr][ph][h], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // 2 NQ half-steps per segment: an even count, so the buffer roles repeat from segment to segment
            if constexpr (NQ % 2 == 1) qo2[0] = qo2[1];      // the next segment's first quad looks in slot 0
        }
        __syncthreads();
    }

    // ---- epilogue: lane (kb, col) holds, per 32-channel block, the 8 channels of C8 group 4 * block + kb for voxel col (+16) of its rows
    const int out_hw = a.Hout * a.Wout;
    const int64_t out_dhw = (int64_t)out_hw * a.Dout;
    const bool relu = (a.flags & SNVC_EPI_RELU) != 0, add_pre = (a.flags & SNVC_EPI_ADD_PRE) != 0, add_post = (a.flags & SNVC_EPI_ADD_POST) != 0;
    constexpr float kHalfMax = 65504.0f;
    float vmax = 0.0f;
#pragma unroll
    for (int blk = 0; blk < Cfg::NHB; ++blk) {
        const int cblk = cb * Cfg::NHB + blk;
        if (cblk * 32 >= a.Cout) break;
        const int c0 = cblk * 32 + 8 * kb;
        float sc[8], bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = a.scale ? a.scale[c0 + e] : 1.0f;
            bi[e] = a.scale ? a.bias[c0 + e] : 0.0f;
        }
        const int64_t gplane = (int64_t)(cblk * 4 + kb) * out_dhw * 8;
        _Float16 *yn = a.y + n * a.y_bs + gplane, *yn_lo = SPLIT ? a.y_lo + n * a.y_bs + gplane : nullptr;
        const _Float16 *rn = a.res ? a.res + n * a.r_bs + gplane : nullptr, *rn_lo = (SPLIT && a.res) ? a.res_lo + n * a.r_bs + gplane : nullptr;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int row = wave * NB + nb;
            const int pd = od0 + row / TH, phh = oh0 + row % TH;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int pw = ow0 + 16 * ph + col;
                const bool ok = pd < a.nd && phh < a.nh && pw < a.nw;
                const int64_t sp = ok ? ((int64_t)(pd * a.osd + a.offd) * out_hw + (phh * a.osh + a.offh) * a.Wout + (pw * a.osw + a.offw)) : 0;
                h8 rv = h8((_Float16)0.0f), rl = h8((_Float16)0.0f);
                if (rn) {
                    rv = *reinterpret_cast<const h8 *>(rn + sp * 8);
                    if constexpr (SPLIT) rl = *reinterpret_cast<const h8 *>(rn_lo + sp * 8);
                }
                h8 o, ol;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = __builtin_fmaf(acc[nb][ph][2 * blk + (e >> 2)][e & 3], sc[e], bi[e]);
                    const float rr = SPLIT ? ((float)rv[e] + (float)rl[e]) * a.res_mul : (float)rv[e];
                    if (rn && add_pre) v += rr;
                    if (relu) v = v > 0.0f ? v : 0.0f;
                    if (rn && add_post) v += rr;
                    if constexpr (SPLIT) {
                        v = __builtin_amdgcn_fmed3f(v, -kHalfMax, kHalfMax);
                        vmax = __builtin_fmaxf(vmax, ok ? __builtin_fabsf(v) : 0.0f);
                    }
                    o[e] = (_Float16)v;
                    if constexpr (SPLIT) ol[e] = (_Float16)(v - (float)o[e]);
                }
                if (ok) {
                    *reinterpret_cast<h8 *>(yn + sp * 8) = o;
                    if constexpr (SPLIT) *reinterpret_cast<h8 *>(yn_lo + sp * 8) = ol;
                }
            }
        }
    }
    if constexpr (SPLIT) {
        if (vmax >= kHalfMax && a.overflow) atomicOr(a.overflow, 1);
    }
}

// packed weights of conv3d_q16s_kernel: [cb][chunk][pass][slice kd][quad][co half (NH)][hi | lo (NPL)][lane][8]
__global__ void pack_q16s_weights_kernel(const float *__restrict__ w, _Float16 *__restrict__ out, int Cout, int Cin, int K3, int NSEG, int NT,
                                         int NQ, int nchunks, int NH, int NPL, int PASSES, float wmul, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int64_t r = i;
    const int e = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int pl = (int)(r % NPL); r /= NPL;
    const int h = (int)(r % NH); r /= NH;
    const int q = (int)(r % NQ); r /= NQ;
    const int kd = (int)(r % NSEG); r /= NSEG;                 // segment: a depth slice of NT = K * K taps, or all K^3 taps (3^3)
    r /= PASSES;                                               // both passes of a chunk see the same weights
    const int chunk = (int)(r % nchunks); r /= nchunks;
    const int cb = (int)r;
    const int row = lane & 15, kb = lane >> 4;
    const int co = (cb * (NH / 2) + (h >> 1)) * 32 + 8 * (row >> 2) + 4 * (h & 1) + (row & 3);
    const int ci = chunk * 8 + e;
    const int tp = 4 * q + kb;
    float v = 0.0f;
    // segment kd holds taps kd * NT .. of the (kd, kh, kw) raster: a depth slice (NT = K * K) or a run of the flat list (NT = 4 NQ slots)
    if (tp < NT && kd * NT + tp < K3 && co < Cout && ci < Cin) v = w[((int64_t)co * Cin + ci) * K3 + kd * NT + tp] * wmul;
    const _Float16 hi = (_Float16)v;
    out[i] = pl == 0 ? hi : (_Float16)(v - (float)hi);
}

// ------------------------------------------------------------------------------------ weight packing
struct PackArgs {
    const float *w;
    _Float16 *out;
    int Cout, Cin, K;          // K = original cubic kernel size
    int transposed, pd, ph, pw;  // transposed: parity class of this packing ([Cin][Cout][3][3][3] source)
    int KD, KH, KW, KCG, MODE, MI, SEGS, TSEG, NPS, NS, unroll_d;
    int nchunks, cblocks;
    int PL;                    // >= 2: split mode, [..][m][hi | lo][lane][8]; values are w * wmul (a power of two) split as hi + lo
    int PASSES;                // 2: planes taken serially, the k-steps of a chunk are stored once per pass
    int dyn;                   // 1: transposed class packed over its real taps only (F16Cfg::DYN): NS = 2 * pairs of this class
    float wmul;
    int64_t total;             // elements (halves)
};

__global__ void pack_f16_weights_kernel(const PackArgs p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.total) return;
    int64_t r = i;
    const int e = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int npl = p.PL >= 2 ? 2 : 1;
    const int pl = (int)(r % npl); r /= npl;
    const int m = (int)(r % p.MI); r /= p.MI;
    const int s = (int)(r % p.NS); r /= p.NS;
    const int seg = (int)(r % p.SEGS); r /= p.SEGS;
    r /= p.PASSES;                                             // both passes of a chunk see the same weights
    const int chunk = (int)(r % p.nchunks); r /= p.nchunks;
    const int cb = (int)r;
    const int half = lane >> 5;
    const int co = (cb * p.MI + m) * 32 + f16_row_channel(lane & 31);
    int tap, g;
    if (p.dyn) { g = s % p.KCG; tap = 2 * (s / p.KCG) + half; }
    else if (p.MODE == 0) { tap = s / (p.KCG / 2); g = 2 * (s % (p.KCG / 2)) + half; }
    else { g = s / p.NPS; tap = 2 * (s % p.NPS) + half; }
    const int ci = (chunk * p.KCG + g) * 8 + e;
    float v = 0.0f;
    if (p.dyn) {
        const int nw = 1 + p.pw, nh = 1 + p.ph, ndp = 1 + p.pd;
        if (tap < nw * nh * ndp && co < p.Cout && ci < p.Cin) {
            const int b[3] = {tap / (nw * nh), (tap / nw) % nh, tap % nw}, par[3] = {p.pd, p.ph, p.pw};
            int k3[3];
            for (int d = 0; d < 3; ++d) k3[d] = par[d] == 0 ? 1 : (b[d] == 0 ? 2 : 0);      // parity 1: box tap 0 = kernel index 2, tap 1 = index 0
            v = p.w[((((int64_t)ci * p.Cout + co) * 3 + k3[0]) * 3 + k3[1]) * 3 + k3[2]];
        }
    } else if (tap < p.TSEG && co < p.Cout && ci < p.Cin) {
        const int kd = p.unroll_d ? tap / (p.KH * p.KW) : seg, kh = (tap / p.KW) % p.KH, kw = tap % p.KW;
        if (!p.transposed) {
            v = p.w[((((int64_t)co * p.Cin + ci) * p.K + kd) * p.K + kh) * p.K + kw];
        } else {
            // parity class: per dim, parity 0 -> only box tap 0 exists (kernel index 1); parity 1 -> box tap 0 is
            // kernel index 2 (input offset 0), box tap 1 is kernel index 0 (input offset +1)
            const int par[3] = {p.pd, p.ph, p.pw}, b[3] = {kd, kh, kw};
            int k3[3];
            bool ok = true;
            for (int d = 0; d < 3; ++d) {
                if (par[d] == 0) { ok = ok && b[d] == 0; k3[d] = 1; }
                else k3[d] = b[d] == 0 ? 2 : 0;
            }
            if (ok) v = p.w[((((int64_t)ci * p.Cout + co) * 3 + k3[0]) * 3 + k3[1]) * 3 + k3[2]];
        }
    }
    if (p.PL >= 2) {
        v *= p.wmul;
        const _Float16 hi = (_Float16)v;
        p.out[i] = pl ? (_Float16)(v - (float)hi) : hi;
    } else {
        p.out[i] = (_Float16)v;
    }
}

// ------------------------------------------------------------------------------------ configurations
//                        KD KH KW S  D  MI TD TH KCG MODE DB   OCC
using F16K1   = F16Cfg<1, 1, 1, 1, 1, 2, 4, 4, 2, 0, true, 2>;
using F16K3   = F16Cfg<3, 3, 3, 1, 1, 2, 4, 4, 2, 0, true, 2>;
using F16K3H  = F16Cfg<3, 3, 3, 1, 1, 1, 4, 4, 2, 0, true, 2>;     // one 32-channel block: the 1-channel head
using F16K3S2 = F16Cfg<3, 3, 3, 2, 1, 2, 2, 4, 1, 1, false, 2>;
using F16K5   = F16Cfg<5, 5, 5, 1, 1, 2, 4, 4, 1, 1, true, 2>;
// dilation 2 = four independent (depth, height)-parity sub-grids, each a convolution with dilation (1,1,2): the image
// of a tile is 8x8x40 pieces (41 KB) instead of 12x12x40 (92 KB: one workgroup per CU with its staging exposed)
using F16K5D2 = F16Cfg<5, 5, 5, 1, 1, 2, 4, 4, 1, 1, false, 2, 2>;
// (a double-buffered image at one workgroup per CU: 14.3 ms against 9.0 on cfg5's conv1 -- two co-resident workgroups matter)
using F16K7   = F16Cfg<7, 7, 7, 1, 1, 2, 4, 4, 1, 1, false, 2>;
using F16DC   = F16Cfg<2, 2, 2, 1, 1, 2, 4, 4, 2, 0, true, 2>;      // one parity class of ConvTranspose3d(k3,s2,p1,op1)
// r3: MI = 1 forms for layers with exactly 32 output channels (the released F = 32 model, vernier.py:249-264: every
// trunk layer but the hourglass's inner ones).  The 64-channel forms above would multiply a second, all-zero weight
// block (half of the MFMAs wasted: released-shape conv1 ran at 0.27 of the fp16 peak).  Same tiles and images; a wave
// carries 4 accumulators instead of 8.
using F16K1N   = F16Cfg<1, 1, 1, 1, 1, 1, 4, 4, 2, 0, true, 2>;
using F16K3S2N = F16Cfg<3, 3, 3, 2, 1, 1, 2, 4, 1, 1, false, 2>;
using F16K5N   = F16Cfg<5, 5, 5, 1, 1, 1, 4, 4, 1, 1, true, 2>;
using F16K5D2N = F16Cfg<5, 5, 5, 1, 1, 1, 4, 4, 1, 1, false, 2, 2>;
using F16K7N   = F16Cfg<7, 7, 7, 1, 1, 1, 4, 4, 1, 1, false, 2>;
using F16DCN   = F16Cfg<2, 2, 2, 1, 1, 1, 4, 4, 2, 0, true, 2>;

// r4 split mode ("f16x3": fp32-accurate layers on the half pipe, see F16Cfg::PL): one channel group per chunk, two taps per MFMA
// (MODE 1), hi and lo planes of the image side by side in LDS (2 x 19.6 KB), single-buffered at two workgroups per CU (three
// before the k-loop's B fragments were double-buffered in registers: 194-238 VGPRs now).  conv2 at cfg2 takes 0.88-0.90 ms in
// every form tried since (2 or 3 WG/CU, image single- or double-buffered, loads sunk or hoisted): the SAME launch on all-zero
// operands takes 0.76 ms -- the layer is bound by the chip's power limit (the clock under this kernel is 1.62 GHz of 2.4), not
// by a stall a schedule could remove; profiles/r4/kernel_experiments_r4.txt item 10.
//                         KD KH KW S  D  MI TD TH KCG MODE DB    OCC DILW PL
using F16K3X  = F16Cfg<3, 3, 3, 1, 1, 1, 4, 4, 1, 1, false, 2, 1, 2>;
using F16K3X2 = F16Cfg<3, 3, 3, 1, 1, 2, 4, 4, 1, 1, false, 2, 1, 2>;
// experiment (desc.algo & SNVC_ALGO_X3_SERIAL): planes taken serially, the 19.6 KB image double-buffered
using F16K3XS  = F16Cfg<3, 3, 3, 1, 1, 1, 4, 4, 1, 1, true, 3, 1, 3>;
// (measured and not kept, conv2 at cfg2: resident planes double-buffered at 2 WG/CU 1.03 ms, a 4x8x32 tile 0.94 ms, the serial
// form above 0.91-1.07 ms, against 0.89-0.92 ms for the single-buffered 4x4x32 tile at 3 WG/CU)
// small layers (SNVC_ALGO_X3_SMALL, chosen by the caller when a launch would not fill the chip): 2x4x32 tiles, one 32-channel
// block per workgroup -- four times the workgroups of F16K3X2 (the hourglass's 48x24x78 level: 216 -> 864)
using F16K3XT  = F16Cfg<3, 3, 3, 1, 1, 1, 2, 4, 1, 1, false, 4, 1, 2>;
using F16K3X2S = F16Cfg<3, 3, 3, 1, 1, 2, 4, 4, 1, 1, true, 2, 1, 3>;
// stride 2: the image of a 2x4x32 tile is 5 x 9 x 65 pieces (46.8 KB per plane): the planes are taken serially (PL = 3)
using F16K3S2X = F16Cfg<3, 3, 3, 2, 1, 2, 2, 4, 1, 1, false, 2, 1, 3>;
// one parity class of ConvTranspose3d(k3,s2,p1,op1): 2x2x2 box taps, both planes resident, double-buffered
// (first forms, all 8 box taps with zero weights where a class has none: KCG = 1 double-buffered 0.255 ms, KCG = 2 0.284 ms on hg conv5)
using F16DCX  = F16Cfg<2, 2, 2, 1, 1, 2, 4, 4, 2, 1, false, 2, 1, 2, true>;
// r5, SNVC_ALGO_X3_SMALL on the stride-2 and transposed split layers: the hourglass's quarter-resolution level (cfg2: 48x24x78) gives
// a 2x4x32 / 4x4x32 tiling 432 / 216 tiles -- fewer workgroups than the chip has slots, each one a serial chain of 16 / 4 single-
// buffered fills whose latency nothing hides.  Half-height tiles: twice the workgroups, 64 accumulator registers instead of 128
// (three workgroups per CU), images of 28 KB (stride 2, per plane) / 32 KB (transposed, both planes).
using F16K3S2XT = F16Cfg<3, 3, 3, 2, 1, 2, 1, 4, 1, 1, false, 3, 1, 3>;
using F16DCXT   = F16Cfg<2, 2, 2, 1, 1, 2, 2, 4, 2, 1, false, 3, 1, 2, true>;

// split forms of the local trunk's other layer kinds (snvc/models/vernier.py:249-278): 1x1x1 (two channel groups per MFMA, both
// planes resident), 5^3 / dilated 5^3 (sub-grid classes) / 7^3 (planes serial: their single-plane images are 37 / 41 / 61 KB),
// the 32-output-channel transposed layer (hourglass_downsample_16's conv12), and the one-channel occupancy head (EPI 1 on F16K3X)
//                         KD KH KW S  D  MI TD TH KCG MODE DB    OCC DILW PL
using F16K1X   = F16Cfg<1, 1, 1, 1, 1, 1, 4, 4, 2, 0, true, 3, 1, 2>;
using F16K5X   = F16Cfg<5, 5, 5, 1, 1, 1, 4, 4, 1, 1, false, 3, 1, 3>;
using F16K5D2X = F16Cfg<5, 5, 5, 1, 1, 1, 4, 4, 1, 1, false, 2, 2, 3>;
using F16K7X   = F16Cfg<7, 7, 7, 1, 1, 1, 4, 4, 1, 1, false, 2, 1, 3>;
using F16DCXN  = F16Cfg<2, 2, 2, 1, 1, 1, 4, 4, 2, 1, false, 3, 1, 2, true>;

enum F16Kind { FK1, FK3, FK3H, FK3S2, FK5, FK5D2, FK7, FDC, FK1N, FK3N, FK3S2N, FK5N, FK5D2N, FK7N, FDCN, FK3X, FK3X2, FK3S2X, FDCX, FK3XS, FK3X2S, FK3XT, FK1X, FK5X, FK5D2X, FK7X, FDCXN, FK3XH, FK3XQ, FK5XQ, FK5D2XQ, FK7XQ, FK5D2Q, FK7Q, FK5D2QN, FK7QN, FK5Q, FK5QN, FK3S2XT, FDCXT, FK3S2XQ, FNONE };

struct F16Plan {
    int kind, MI, KCG, MODE, TD, TH, STEPS, SEGS, TSEG, NPS, NS, KD, KH, KW, unroll_d, PL, PF, PASSES, dyn;
    int nchunks, cblocks;
    int64_t block_halves;      // packed halves of one class (without padding)
};

template <class Cfg>
F16Plan plan_from(int kind) {
    F16Plan p{};
    p.kind = kind; p.MI = Cfg::MI; p.KCG = Cfg::KCG; p.MODE = Cfg::MODE; p.TD = Cfg::TD; p.TH = Cfg::TH;
    p.STEPS = Cfg::STEPS; p.SEGS = Cfg::SEGS; p.TSEG = Cfg::TSEG; p.NPS = Cfg::NPS; p.NS = Cfg::NS;
    p.KD = Cfg::KD; p.KH = Cfg::KH; p.KW = Cfg::KW; p.unroll_d = Cfg::UNROLL_D ? 1 : 0;
    p.PL = Cfg::PL; p.PF = Cfg::PF; p.PASSES = Cfg::PASSES; p.dyn = Cfg::DYN ? 1 : 0;
    return p;
}

int make_f16_plan(const snvc_conv3d_desc &d, F16Plan &p, bool split = false) {
    if (d.N < 0 || d.Cin <= 0 || d.Cout <= 0 || d.Din <= 0 || d.Hin <= 0 || d.Win <= 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d: sizes must be positive");
    if (d.Cin % 8 != 0) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d: Cin must be a multiple of 8 (C8 layout)");
    if (split) {
        if (d.transposed) {
            if (d.ksize != 3 || d.stride != 2 || d.pad != 1 || d.dilation != 1 || d.Cout % 32 != 0)
                return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d: transposed layers: k3,s2,p1,op1 with Cout % 32 == 0");
            if (d.Dout != 2 * d.Din || d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win)
                return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d: transposed output must be 2x the input");
            p = d.Cout % 64 == 0 ? ((d.algo & SNVC_ALGO_X3_SMALL) ? plan_from<F16DCXT>(FDCXT) : plan_from<F16DCX>(FDCX)) : plan_from<F16DCXN>(FDCXN);
        } else {
            if (d.pad != d.dilation * (d.ksize - 1) / 2)
                return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d: pad must equal dilation*(ksize-1)/2");
            const int eff = d.dilation * (d.ksize - 1) + 1;
            if (d.Dout != (d.Din + 2 * d.pad - eff) / d.stride + 1 || d.Hout != (d.Hin + 2 * d.pad - eff) / d.stride + 1 ||
                d.Wout != (d.Win + 2 * d.pad - eff) / d.stride + 1)
                return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d: output size does not match the convolution arithmetic");
            const int key = d.ksize * 100 + d.stride * 10 + d.dilation;
            if (d.Cout == 1) {
                if (key != 311) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d: one-channel output is built for k3/s1 only");
                p = plan_from<F16K3X>(FK3XH);
            } else if (key == 321) {
                if (d.Cout % 64 != 0) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d: stride-2 layers need Cout % 64 == 0");
                if (d.algo & SNVC_ALGO_X3_Q16) {      // the 16x16x32 form: both planes, three image slots, one workgroup per CU
                    p = plan_from<F16K3S2X>(FK3S2XQ);
                    p.KCG = 1; p.MI = 2; p.STEPS = X3S2Cfg::NQ; p.NS = X3S2Cfg::NQ; p.PF = 0; p.PASSES = 1;
                    p.nchunks = d.Cin / 8;
                    p.cblocks = d.Cout / 64;
                    p.block_halves = (int64_t)p.cblocks * p.nchunks * X3S2Cfg::NQ * 8 * 64 * 8;
                    if (p.cblocks > 65535 || d.N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d: too many channel blocks or samples");
                    return SNVC_OK;
                }
                p = (d.algo & SNVC_ALGO_X3_SMALL) ? plan_from<F16K3S2XT>(FK3S2XT) : plan_from<F16K3S2X>(FK3S2X);
            } else {
                if (d.Cout % 32 != 0) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d: Cout % 32 == 0");
                if ((d.algo & SNVC_ALGO_X3_Q16) && (key == 511 || key == 512 || key == 711)) {      // 16x16x32 form, planes serial
                    const int ks = d.ksize;
                    p = key == 511 ? plan_from<F16K5X>(FK5XQ) : (key == 512 ? plan_from<F16K5D2X>(FK5D2XQ) : plan_from<F16K7X>(FK7XQ));
                    p.KCG = 1; p.MI = 1; p.PF = 2; p.STEPS = q16s_steps(ks);
                    p.nchunks = d.Cin / 8;
                    p.cblocks = d.Cout / 32;
                    p.block_halves = (int64_t)p.cblocks * p.nchunks * 2 * p.STEPS * 4 * 64 * 8;
                    if (p.cblocks > 65535 || d.N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d: too many channel blocks or samples");
                    return SNVC_OK;
                }
                switch (key) {
                    case 111: p = plan_from<F16K1X>(FK1X); break;
                    case 511: p = plan_from<F16K5X>(FK5X); break;
                    case 512: p = plan_from<F16K5D2X>(FK5D2X); break;
                    case 711: p = plan_from<F16K7X>(FK7X); break;
                    case 311:
                        if (d.algo & SNVC_ALGO_X3_Q16) {      // the 16x16x32 form (four taps x one channel group per MFMA): its own packing
                            p = plan_from<F16K3X>(FK3XQ);
                            p.KCG = 1; p.MI = 1; p.STEPS = X3QCfg::NQ; p.NS = X3QCfg::NQ; p.PF = X3QCfg::PF;
                            p.nchunks = d.Cin / 8;
                            p.cblocks = d.Cout / 32;
                            p.block_halves = (int64_t)p.cblocks * p.nchunks * X3QCfg::NQ * 4 * 64 * 8;
                            if (p.cblocks > 65535 || d.N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d: too many channel blocks or samples");
                            return SNVC_OK;
                        }
                        if (d.algo & SNVC_ALGO_X3_SMALL) p = plan_from<F16K3XT>(FK3XT);
                        else if (d.algo & SNVC_ALGO_X3_NARROW) p = plan_from<F16K3X>(FK3X);
                        else if (d.algo & SNVC_ALGO_X3_SERIAL) p = d.Cout == 32 ? plan_from<F16K3XS>(FK3XS) : plan_from<F16K3X2S>(FK3X2S);
                        else p = d.Cout == 32 ? plan_from<F16K3X>(FK3X) : plan_from<F16K3X2>(FK3X2);
                        break;
                    default:
                        return fail(SNVC_ERR_UNSUPPORTED,
                                    "snvc_f16x3_conv3d: (ksize,stride,dilation) not in {(1,1,1),(3,1,1),(3,2,1),(5,1,1),(5,1,2),(7,1,1)}");
                }
            }
        }
        p.nchunks = ceil_div(d.Cin / 8, p.KCG);
        p.cblocks = ceil_div(d.Cout, 32 * p.MI);
        p.block_halves = (int64_t)p.cblocks * p.nchunks * p.PASSES * p.STEPS * p.MI * 2 * 64 * 8;
        if (p.cblocks > 65535 || d.N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d: too many channel blocks or samples");
        return SNVC_OK;
    }
    if (d.Cout != 1 && d.Cout % 32 != 0)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d: Cout must be a multiple of 32, or 1 (fp32 plane output)");
    if (d.transposed) {
        if (d.ksize != 3 || d.stride != 2 || d.pad != 1 || d.dilation != 1)
            return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d: transposed conv supports k3,s2,p1,op1 only");
        if (d.Dout != 2 * d.Din || d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win)
            return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d: transposed output must be 2x the input");
        if (d.Cout == 1) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d: transposed conv to one channel");
        p = d.Cout == 32 ? plan_from<F16DCN>(FDCN) : plan_from<F16DC>(FDC);
    } else {
        if (d.pad != d.dilation * (d.ksize - 1) / 2)
            return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d: pad must equal dilation*(ksize-1)/2");
        const int eff = d.dilation * (d.ksize - 1) + 1;
        if (d.Dout != (d.Din + 2 * d.pad - eff) / d.stride + 1 || d.Hout != (d.Hin + 2 * d.pad - eff) / d.stride + 1 ||
            d.Wout != (d.Win + 2 * d.pad - eff) / d.stride + 1)
            return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d: output size does not match the convolution arithmetic");
        const int key = d.ksize * 100 + d.stride * 10 + d.dilation;
        if (d.Cout == 1) {
            if (key != 311) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d: one-channel output is built for k3/s1 only");
            p = plan_from<F16K3H>(FK3H);
        } else {
            const bool narrow = d.Cout == 32;      // one 32-channel block: the MI = 1 forms
            if ((d.algo & SNVC_ALGO_X3_Q16) && (key == 711 || key == 512 || key == 511) && d.Cout % 32 == 0) {     // 16x16x32 form, two blocks per
                const int ks = d.ksize;                                                             // workgroup (one for Cout = 32 * odd)
                const bool two = d.Cout % 64 == 0;
                p = key == 711 ? plan_from<F16K7>(two ? FK7Q : FK7QN) : key == 512 ? plan_from<F16K5D2>(two ? FK5D2Q : FK5D2QN)
                                                                                   : plan_from<F16K5>(two ? FK5Q : FK5QN);
                p.KCG = 1; p.MI = two ? 2 : 1; p.PF = 2; p.STEPS = q16s_steps(ks);
                p.nchunks = d.Cin / 8;
                p.cblocks = d.Cout / (two ? 64 : 32);
                p.block_halves = (int64_t)p.cblocks * p.nchunks * p.STEPS * (two ? 4 : 2) * 64 * 8;
                if (p.cblocks > 65535 || d.N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d: too many channel blocks or samples");
                return SNVC_OK;
            }
            switch (key) {
                case 111: p = narrow ? plan_from<F16K1N>(FK1N) : plan_from<F16K1>(FK1); break;
                case 311: p = narrow ? plan_from<F16K3H>(FK3N) : plan_from<F16K3>(FK3); break;
                case 321: p = narrow ? plan_from<F16K3S2N>(FK3S2N) : plan_from<F16K3S2>(FK3S2); break;
                case 511: p = narrow ? plan_from<F16K5N>(FK5N) : plan_from<F16K5>(FK5); break;
                case 512: p = narrow ? plan_from<F16K5D2N>(FK5D2N) : plan_from<F16K5D2>(FK5D2); break;
                case 711: p = narrow ? plan_from<F16K7N>(FK7N) : plan_from<F16K7>(FK7); break;
                default:
                    return fail(SNVC_ERR_UNSUPPORTED,
                                "snvc_f16_conv3d: (ksize,stride,dilation) not in {(1,1,1),(3,1,1),(3,2,1),(5,1,1),(5,1,2),(7,1,1)}");
            }
        }
    }
    p.nchunks = ceil_div(d.Cin / 8, p.KCG);
    p.cblocks = ceil_div(d.Cout, 32 * p.MI);
    p.block_halves = (int64_t)p.cblocks * p.nchunks * p.STEPS * p.MI * 64 * 8;
    if (p.cblocks > 65535 || d.N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d: too many channel blocks or samples");
    return SNVC_OK;
}

inline int64_t f16_class_stride(const F16Plan &p) {
    if (p.kind == FK3S2XQ) return p.block_halves + (int64_t)X3S2Cfg::NQ * 8 * 64 * 8;      // the last chunk's phase H never reads ahead, but stays in bounds if it did
    if (p.kind == FK3XQ || p.kind == FK5XQ || p.kind == FK5D2XQ || p.kind == FK7XQ || p.kind == FK5D2Q || p.kind == FK7Q || p.kind == FK5Q)
        return p.block_halves + (int64_t)p.PF * 4 * 64 * 8;
    if (p.kind == FK5D2QN || p.kind == FK7QN || p.kind == FK5QN) return p.block_halves + (int64_t)p.PF * 2 * 64 * 8;
    return p.block_halves + (int64_t)p.PF * p.MI * (p.PL >= 2 ? 2 : 1) * 64 * 8;
}

// A fragments of the tail projection (conv3d_f16_kernel EPI 4): [m][j][hi | lo][lane][8]; row (lane & 31) = tap, k = 8 * (lane >> 5) + e
// is channel 32 m + 16 (lane >> 5) + 8 j + e -- the channel that accumulator register 8 j + e of lane (voxel, lane >> 5) holds
__global__ void pack_tail_weights_kernel(const float *__restrict__ w, _Float16 *__restrict__ out, int Cin, float wmul, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int r = i;
    const int e = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int pl = r % 2; r /= 2;
    const int j = r % 2; r /= 2;
    const int m = r;
    const int tap = lane & 31, c = 32 * m + 16 * (lane >> 5) + 8 * j + e;
    const float v = (tap < 27 && c < Cin) ? w[c * 27 + tap] * wmul : 0.0f;
    const _Float16 hi = (_Float16)v;
    out[i] = pl == 0 ? hi : (_Float16)(v - (float)hi);
}

// out[o] = bias + res[o] + sum over (i, k) with o = 2 i - 1 + k (per dimension) of T[k][i]: the scatter half of a transposed layer
// (k3, s2, p1, op1) to one channel whose per-voxel contraction T was formed by the producing layer (EPI 4).  T is class-major:
// [27][8 = (i_d & 1, i_h & 1, i_w & 1)][nd][nh][nw], i = 2 p + r.  A thread owns 4 consecutive outputs of one output row
// (ow = 4 j .. 4 j + 3): every T element is read exactly once, 4-byte loads consecutive across the lanes, one 16-byte store.
__global__ void __launch_bounds__(256)
deconv_tail_gather_kernel(const float *__restrict__ t, const float *__restrict__ bias, const float *__restrict__ res, float *__restrict__ y,
                          int nd, int nh, int nw, int64_t t_bs, int64_t y_bs, int64_t r_bs) {
    const int OD = 4 * nd, OH = 4 * nh;
    const int64_t total = (int64_t)OD * OH * nw;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % nw);
    const int64_t r0 = i / nw;
    const int oh = (int)(r0 % OH), od = (int)(r0 / OH);
    const int64_t n = blockIdx.y;
    const int64_t cls_vox = (int64_t)nd * nh * nw, tap_stride = 8 * cls_vox;
    const float *tn = t + n * t_bs;
    const float b = bias ? bias[0] : 0.0f;
    float o0 = b, o1 = b, o2 = b, o3 = b;
    // per dimension: even output 2 a: (k = 1, i = a); odd output 2 a + 1: (k = 2, i = a) and (k = 0, i = a + 1)
    const int ad = od >> 1, ah = oh >> 1;
    const int ndd = (od & 1) ? 2 : 1, nhh = (oh & 1) ? 2 : 1;
    const bool last_w = j + 1 >= nw;
    for (int a_ = 0; a_ < ndd; ++a_) {
        const int kd = (od & 1) ? (a_ ? 0 : 2) : 1, id = ad + a_;
        if (id >= 2 * nd) continue;
        for (int b_ = 0; b_ < nhh; ++b_) {
            const int kh = (oh & 1) ? (b_ ? 0 : 2) : 1, ih = ah + b_;
            if (ih >= 2 * nh) continue;
            const int cdh = ((id & 1) << 2) | ((ih & 1) << 1);
            const int64_t base = ((int64_t)(id >> 1) * nh + (ih >> 1)) * nw + j;
            const float *t0 = tn + (int64_t)((kd * 3 + kh) * 3) * tap_stride + base;      // kw = 0
            const float *c0 = t0 + (int64_t)cdh * cls_vox, *c1 = t0 + (int64_t)(cdh | 1) * cls_vox;      // i_w even / odd at p_w = j
            // ow = 4j: (kw 1, i_w = 2j); 4j+1: (kw 2, 2j) + (kw 0, 2j+1); 4j+2: (kw 1, 2j+1); 4j+3: (kw 2, 2j+1) + (kw 0, 2j+2)
            o0 += c0[tap_stride];
            o1 += c0[2 * tap_stride] + c1[0];
            o2 += c1[tap_stride];
            o3 += c1[2 * tap_stride] + (last_w ? 0.0f : c0[1]);
        }
    }
    const int64_t sp = ((int64_t)od * OH + oh) * (4 * nw) + 4 * j;
    if (res) {
        const f32x4 rv = *reinterpret_cast<const f32x4 *>(res + n * r_bs + sp);
        o0 += rv[0]; o1 += rv[1]; o2 += rv[2]; o3 += rv[3];
    }
    *reinterpret_cast<f32x4 *>(y + n * y_bs + sp) = f32x4{o0, o1, o2, o3};
}

// The power of two that puts max|x| into [2^13, 2^14) (1 for an all-zero or non-finite tensor) -- the scale of a split pair whose range is
// only known from the data -- in ONE launch (r5; the torch expression it replaces was ten small launches): every workgroup folds its
// share into scratch[0] (float bits of a non-negative maximum order like unsigned integers), the last one to arrive (scratch[1]) turns
// the maximum into the scale and leaves both words zero for the next call.  NaNs do not take part (fmaxf drops them), +-inf gives 1.
__global__ void __launch_bounds__(256)
split_scale_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ scratch, float *__restrict__ out) {
    float m = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    const int64_t n4 = n & ~(int64_t)3;
    auto amax4 = [](const f32x4 v) {
        return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
    };
    // few workgroups, four independent 16-byte loads in flight per thread: the launch's time is the chain of same-address atomics
    // at its end (r5: 234 workgroups 9.1 us for 3.8 MB; <= 48 workgroups: see profiles/r5)
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(x + i), v1 = *reinterpret_cast<const f32x4 *>(x + i + stride);
        const f32x4 v2 = *reinterpret_cast<const f32x4 *>(x + i + 2 * stride), v3 = *reinterpret_cast<const f32x4 *>(x + i + 3 * stride);
        m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fmaxf(amax4(v0), amax4(v1)), __builtin_fmaxf(amax4(v2), amax4(v3))));
    }
    for (; i < n4; i += stride) m = __builtin_fmaxf(m, amax4(*reinterpret_cast<const f32x4 *>(x + i)));
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4)) m = __builtin_fmaxf(m, __builtin_fabsf(x[n4 + threadIdx.x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = __builtin_fmaxf(__builtin_fmaxf(part[0], part[1]), __builtin_fmaxf(part[2], part[3]));
        atomicMax(scratch, __builtin_bit_cast(unsigned, m));
        __threadfence();
        if (atomicAdd(scratch + 1, 1u) == gridDim.x - 1) {
            const float amax = __builtin_bit_cast(float, atomicExch(scratch, 0u));
            atomicExch(scratch + 1, 0u);
            float e = floorf(log2f(16384.0f / amax));          // amax = 0 -> +inf, amax = inf -> -inf: "not finite" below
            e = (e == e && __builtin_fabsf(e) < 3.0e38f) ? __builtin_fminf(__builtin_fmaxf(e, -24.0f), 40.0f) : 0.0f;
            out[0] = exp2f(e);
        }
    }
}

template <class Cfg, int EPI>
void launch_f16(const F16Args &a, dim3 grid, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    if (!allow_large_lds(reinterpret_cast<const void *>(&conv3d_f16_kernel<Cfg, EPI>), Cfg::LDS_BYTES, attr_done)) return;
    conv3d_f16_kernel<Cfg, EPI><<<grid, 256, Cfg::LDS_BYTES, st>>>(a);
}

}  // namespace
}  // namespace snvc

extern "C" {

int64_t snvc_f16_conv3d_packed_weight_bytes(const snvc_conv3d_desc *d) {
    using namespace snvc;
    F16Plan p;
    if (!d || make_f16_plan(*d, p) != SNVC_OK) return -1;
    return 2 * f16_class_stride(p) * (d->transposed ? 8 : 1);
}

static int f16_pack_common(const snvc_conv3d_desc *d, const float *weight, void *packed, bool split, float wmul, void *stream,
                           const char *who) {
    using namespace snvc;
    F16Plan p;
    if (!d) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d_pack_weights: null desc");
    int rc = make_f16_plan(*d, p, split);
    if (rc) return rc;
    if (!weight || !packed) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d_pack_weights: null pointer");
    const int classes = d->transposed ? 8 : 1;
    const int64_t bytes = 2 * f16_class_stride(p) * classes;
    if (hipMemsetAsync(packed, 0, (size_t)bytes, as_stream(stream)) != hipSuccess)   // the ring's read-ahead padding
        return fail(SNVC_ERR_HIP, "snvc_f16_conv3d_pack_weights: hipMemsetAsync failed");
    if (p.kind == FK5XQ || p.kind == FK5D2XQ || p.kind == FK7XQ || p.kind == FK5D2Q || p.kind == FK7Q || p.kind == FK5D2QN || p.kind == FK7QN ||
        p.kind == FK5Q || p.kind == FK5QN) {
        const int ks = d->ksize, k3 = ks * ks * ks;
        const int fseg = q16s_flat_segments(ks);      // 5^3: one segment of all 125 taps; 7^3: two of 172 slots
        const int nseg = fseg, nq = ((k3 + fseg - 1) / fseg + 3) / 4, nt = 4 * nq;
        const bool sp_ = p.PL >= 2;
        const int nh_ = sp_ ? 2 : 2 * p.MI;
        pack_q16s_weights_kernel<<<(unsigned)ceil_div<int64_t>(p.block_halves, 256), 256, 0, as_stream(stream)>>>(
            weight, reinterpret_cast<_Float16 *>(packed), d->Cout, d->Cin, k3, nseg, nt, nq, p.nchunks, nh_, sp_ ? 2 : 1, sp_ ? 2 : 1, wmul,
            p.block_halves);
        return check_launch(who);
    }
    if (p.kind == FK3S2XQ) {    // [cb (64 channels)][chunk][quad][co half (4)][hi | lo][lane][8]
        pack_q16s_weights_kernel<<<(unsigned)ceil_div<int64_t>(p.block_halves, 256), 256, 0, as_stream(stream)>>>(
            weight, reinterpret_cast<_Float16 *>(packed), d->Cout, d->Cin, 27, 1, 4 * X3S2Cfg::NQ, X3S2Cfg::NQ, p.nchunks, 4, 2, 1, wmul, p.block_halves);
        return check_launch(who);
    }
    if (p.kind == FK3XQ) {      // [cb][chunk][quad][co half][hi | lo][lane][8]: one segment of all 27 taps
        pack_q16s_weights_kernel<<<(unsigned)ceil_div<int64_t>(p.block_halves, 256), 256, 0, as_stream(stream)>>>(
            weight, reinterpret_cast<_Float16 *>(packed), d->Cout, d->Cin, 27, 1, 4 * X3QCfg::NQ, X3QCfg::NQ, p.nchunks, 2, 2, 1, wmul, p.block_halves);
        return check_launch(who);
    }
    for (int c = 0; c < classes; ++c) {
        PackArgs a;
        a.w = weight;
        a.out = reinterpret_cast<_Float16 *>(packed) + c * f16_class_stride(p);
        a.Cout = d->Cout; a.Cin = d->Cin; a.K = d->ksize;
        a.transposed = d->transposed; a.pd = (c >> 2) & 1; a.ph = (c >> 1) & 1; a.pw = c & 1;
        a.KD = p.KD; a.KH = p.KH; a.KW = p.KW; a.KCG = p.KCG; a.MODE = p.MODE; a.MI = p.MI; a.SEGS = p.SEGS;
        a.TSEG = p.TSEG; a.NPS = p.NPS; a.NS = p.NS; a.unroll_d = p.unroll_d;
        a.nchunks = p.nchunks; a.cblocks = p.cblocks; a.total = p.block_halves;
        a.PL = p.PL; a.PASSES = p.PASSES; a.wmul = wmul;
        a.dyn = p.dyn;
        if (p.dyn) {        // compact per class: 2 * pairs k-steps per chunk
            const int T = (1 + a.pd) * (1 + a.ph) * (1 + a.pw);
            a.NS = 2 * (T > 1 ? T / 2 : 1);
            a.total = (int64_t)p.cblocks * p.nchunks * a.NS * p.MI * 2 * 64 * 8;
        }
        pack_f16_weights_kernel<<<(unsigned)ceil_div<int64_t>(a.total, 256), 256, 0, as_stream(stream)>>>(a);
    }
    return check_launch(who);
}

int snvc_f16_conv3d_pack_weights(const snvc_conv3d_desc *d, const float *weight, void *packed, void *stream) {
    return f16_pack_common(d, weight, packed, false, 1.0f, stream, "snvc_f16_conv3d_pack_weights");
}

int64_t snvc_f16x3_conv3d_packed_weight_bytes(const snvc_conv3d_desc *d) {
    using namespace snvc;
    F16Plan p;
    if (!d || make_f16_plan(*d, p, true) != SNVC_OK) return -1;
    return 2 * f16_class_stride(p) * (d->transposed ? 8 : 1);
}

int snvc_f16x3_conv3d_pack_weights(const snvc_conv3d_desc *d, const float *weight, void *packed, float wmul, void *stream) {
    return f16_pack_common(d, weight, packed, true, wmul, stream, "snvc_f16x3_conv3d_pack_weights");
}

int snvc_f16_conv3d_forward(const snvc_conv3d_desc *d, const void *x, const void *packed_weight, const float *scale,
                            const float *bias, const void *residual, void *y, float *y_f32, void *stream) {
    using namespace snvc;
    F16Plan p;
    if (!d) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d_forward: null desc");
    int rc = make_f16_plan(*d, p);
    if (rc) return rc;
    if (d->N == 0) return SNVC_OK;
    const bool plane = d->Cout == 1;
    if (!x || !packed_weight || (plane ? !y_f32 : !y))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d_forward: null pointer");
    if ((scale == nullptr) != (bias == nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d_forward: scale and bias must both be given or both be NULL");
    const int resflags = d->flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST);
    if (resflags && (!residual || plane))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d_forward: residual flag without a C8 residual");
    if (resflags == (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d_forward: ADD_PRE and ADD_POST are exclusive");
    if (!plane && (d->flags & SNVC_EPI_SIGMOID))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d_forward: Sigmoid is built for the one-channel fp32 output only");
    const int64_t in_sp = (int64_t)d->Din * d->Hin * d->Win, out_sp = (int64_t)d->Dout * d->Hout * d->Wout;
    if ((int64_t)(d->Cin / 8 + 2) * in_sp >= ((int64_t)1 << 31) || out_sp >= ((int64_t)1 << 31))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d_forward: one sample must stay below 2^31 pieces");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
         reinterpret_cast<uintptr_t>(packed_weight)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d_forward: C8 tensors must be 16-byte aligned");
    if ((d->x_batch_stride | d->y_batch_stride | d->res_batch_stride) % 8)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16_conv3d_forward: batch strides must be multiples of 8 elements");

    F16Args a{};
    a.x = reinterpret_cast<const _Float16 *>(x);
    a.scale = scale; a.bias = bias;
    a.res = resflags ? reinterpret_cast<const _Float16 *>(residual) : nullptr;
    a.y = reinterpret_cast<_Float16 *>(y); a.y_f32 = y_f32;
    a.res_mul = 1.0f;
    a.CGin = d->Cin / 8;
    a.Cout = d->Cout;
    a.Din = d->Din; a.Hin = d->Hin; a.Win = d->Win;
    a.Dout = d->Dout; a.Hout = d->Hout; a.Wout = d->Wout;
    a.nchunks = p.nchunks; a.flags = d->flags;
    a.x_bs = d->x_batch_stride ? d->x_batch_stride : (int64_t)d->Cin * in_sp;
    a.y_bs = d->y_batch_stride ? d->y_batch_stride : (int64_t)d->Cout * out_sp;
    a.r_bs = d->res_batch_stride ? d->res_batch_stride : (int64_t)d->Cout * out_sp;
    a.yf_bs = out_sp;
    hipStream_t st = as_stream(stream);
    const bool subgrid = p.kind == FK5D2 || p.kind == FK5D2N || p.kind == FK5D2Q || p.kind == FK5D2QN;      // (depth, height) parity classes share ONE packed weight block
    const int classes = d->transposed ? 8 : (subgrid ? 4 : 1);
    a.wp = reinterpret_cast<const _Float16 *>(packed_weight);
    a.N = d->N; a.cls_mode = d->transposed ? 1 : (subgrid ? 2 : 0); a.cls_wstride = f16_class_stride(p);
    a.isd = a.ish = 1; a.iod = a.ioh = 0; a.offd = a.offh = a.offw = 0;
    if (d->transposed) {
        a.nd = d->Din; a.nh = d->Hin; a.nw = d->Win;
        a.osd = a.osh = a.osw = 2;
        a.pad_d = a.pad_h = a.pad_w = 0;
    } else if (subgrid) {
        a.nd = (d->Dout + 1) / 2; a.nh = (d->Hout + 1) / 2; a.nw = d->Wout;     // the largest class; the kernel trims the others
        a.osd = a.osh = 2; a.osw = 1;
        a.isd = a.ish = 2;
        a.pad_d = a.pad_h = d->pad / 2; a.pad_w = d->pad;
    } else {
        a.nd = d->Dout; a.nh = d->Hout; a.nw = d->Wout;
        a.osd = a.osh = a.osw = 1;
        a.pad_d = a.pad_h = a.pad_w = d->pad;
    }
    a.tiles_d = ceil_div(a.nd, p.TD); a.tiles_h = ceil_div(a.nh, p.TH); a.tiles_w = ceil_div(a.nw, 32);
    const int64_t ntiles = (int64_t)a.tiles_d * a.tiles_h * a.tiles_w;
    if (ntiles >= ((int64_t)1 << 30) || (int64_t)d->N * classes > 65535)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d_forward: too many tiles or samples");
    dim3 grid((unsigned)(d->transposed ? 2 * ntiles : ntiles), (unsigned)p.cblocks, (unsigned)(d->N * (d->transposed ? 4 : classes)));
    switch (p.kind) {
        case FK7QN: case FK5D2QN: case FK5QN: {
            using C7 = Q16SCfg<7, 1, 1, 1>; using C5D = Q16SCfg<5, 2, 1, 1>; using C5 = Q16SCfg<5, 1, 1, 1>;
            static std::atomic<unsigned> at7{0}, at5{0}, at5p{0};
            if (p.kind == FK7QN) {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_q16s_kernel<C7>), C7::LDS_BYTES, at7))
                    conv3d_q16s_kernel<C7><<<grid, 256, C7::LDS_BYTES, st>>>(a);
            } else if (p.kind == FK5QN) {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_q16s_kernel<C5>), C5::LDS_BYTES, at5p))
                    conv3d_q16s_kernel<C5><<<grid, 256, C5::LDS_BYTES, st>>>(a);
            } else {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_q16s_kernel<C5D>), C5D::LDS_BYTES, at5))
                    conv3d_q16s_kernel<C5D><<<grid, 256, C5D::LDS_BYTES, st>>>(a);
            }
            break;
        }
        case FK7Q: case FK5D2Q: case FK5Q: {
            if (plane) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d_forward: the 16x16x32 form writes C8 tensors");
            using C7 = Q16SCfg<7, 1, 1, 2>; using C5D = Q16SCfg<5, 2, 1, 2>; using C5 = Q16SCfg<5, 1, 1, 2>;
            static std::atomic<unsigned> at7{0}, at5{0}, at5p{0};
            if (p.kind == FK7Q) {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_q16s_kernel<C7>), C7::LDS_BYTES, at7))
                    conv3d_q16s_kernel<C7><<<grid, 256, C7::LDS_BYTES, st>>>(a);
            } else if (p.kind == FK5Q) {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_q16s_kernel<C5>), C5::LDS_BYTES, at5p))
                    conv3d_q16s_kernel<C5><<<grid, 256, C5::LDS_BYTES, st>>>(a);
            } else {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_q16s_kernel<C5D>), C5D::LDS_BYTES, at5))
                    conv3d_q16s_kernel<C5D><<<grid, 256, C5D::LDS_BYTES, st>>>(a);
            }
            break;
        }
        case FK1: launch_f16<F16K1, 0>(a, grid, st); break;
        case FK3: launch_f16<F16K3, 0>(a, grid, st); break;
        case FK3H: launch_f16<F16K3H, 1>(a, grid, st); break;
        case FK3S2: launch_f16<F16K3S2, 0>(a, grid, st); break;
        case FK5: launch_f16<F16K5, 0>(a, grid, st); break;
        case FK5D2: launch_f16<F16K5D2, 0>(a, grid, st); break;
        case FK7: launch_f16<F16K7, 0>(a, grid, st); break;
        case FDC: launch_f16<F16DC, 0>(a, grid, st); break;
        case FK1N: launch_f16<F16K1N, 0>(a, grid, st); break;
        case FK3N: launch_f16<F16K3H, 0>(a, grid, st); break;
        case FK3S2N: launch_f16<F16K3S2N, 0>(a, grid, st); break;
        case FK5N: launch_f16<F16K5N, 0>(a, grid, st); break;
        case FK5D2N: launch_f16<F16K5D2N, 0>(a, grid, st); break;
        case FK7N: launch_f16<F16K7N, 0>(a, grid, st); break;
        case FDCN: launch_f16<F16DCN, 0>(a, grid, st); break;
        default: return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16_conv3d_forward: no kernel");
    }
    return check_launch("snvc_f16_conv3d_forward");
}

static int f16x3_forward(const snvc_conv3d_desc *d, const void *x_hi, const void *x_lo, const void *packed_weight,
                         const float *scale, const float *bias, const void *res_hi, const void *res_lo, void *y_hi,
                         void *y_lo, float *y_f32, const float *head, float *y_head, float head_mul, float res_mul,
                         int *overflow, const void *tail_w, float *t_out, float tail_mul, void *stream, double *stats = nullptr,
                         int64_t *stats_slots = nullptr, const float *x_mul = nullptr) {
    using namespace snvc;
    F16Plan p;
    if (!d) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward: null desc");
    int rc = make_f16_plan(*d, p, true);
    if (rc) return rc;
    if (d->N == 0) return SNVC_OK;
    const bool tail = tail_w != nullptr;          // EPI 4: the result is contracted with a one-channel transposed layer's taps, not stored
    if (tail && (!t_out || !d->transposed || p.cblocks != 1 || d->Cout != 32 * p.MI || (reinterpret_cast<uintptr_t>(tail_w) & 15)))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_deconv3d_tail_forward: a transposed split layer whose 32 or 64 output channels sit in one block");
    const bool plane = d->Cout == 1;              // the occupancy head: y_f32 is then the fp32 plane [N][1][D][H][W] (Sigmoid honoured)
    const bool to_f32 = y_f32 != nullptr;
    if (plane && !to_f32) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward: a one-channel layer writes y_f32");
    if (!x_hi || !x_lo || !packed_weight || (!tail && !to_f32 && (!y_hi || !y_lo)))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward: null pointer");
    if ((scale == nullptr) != (bias == nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward: scale and bias must both be given or both be NULL");
    int resflags = d->flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST);
    // r6: a float32 result (y_f32) takes a float32 NCDHW residual of its own shape and batch stride: res_hi = that tensor, res_lo = NULL,
    // ADD_POST only (the data gradients of the training step add a skip connection's gradient this way)
    const float *res_f32 = nullptr;
    if (resflags == SNVC_EPI_ADD_POST && to_f32 && !plane && res_hi && !res_lo && !tail_w) {
        if (d->res_batch_stride && d->res_batch_stride != (d->y_batch_stride ? d->y_batch_stride : (int64_t)d->Cout * d->Dout * d->Hout * d->Wout))
            return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward: a float32 residual shares the result's batch stride");
        res_f32 = reinterpret_cast<const float *>(res_hi);
        res_hi = nullptr;
        resflags = 0;
    }
    if (resflags && (!res_hi || !res_lo || plane))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward: residual flag without a split residual");
    if (resflags == (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward: ADD_PRE and ADD_POST are exclusive");
    if (d->flags & ~(SNVC_EPI_RELU | SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST | (plane ? SNVC_EPI_SIGMOID : 0)))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward: RELU / ADD_PRE / ADD_POST only (SIGMOID with the one-channel output)");
    if ((head != nullptr) != (y_head != nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward: head and y_head go together");
    if (head && ((p.kind != FK3X && p.kind != FK3XS && p.kind != FK3XT && p.kind != FK3XQ) || d->Cout != 32 || to_f32))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward: the side head is built for 32-channel stride-1 layers with a split output");
    const int64_t in_sp = (int64_t)d->Din * d->Hin * d->Win, out_sp = (int64_t)d->Dout * d->Hout * d->Wout;
    if ((int64_t)(d->Cin / 8 + 2) * in_sp >= ((int64_t)1 << 31) || out_sp >= ((int64_t)1 << 31))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward: one sample must stay below 2^31 pieces");
    if ((reinterpret_cast<uintptr_t>(x_hi) | reinterpret_cast<uintptr_t>(x_lo) | reinterpret_cast<uintptr_t>(y_hi) |
         reinterpret_cast<uintptr_t>(y_lo) | reinterpret_cast<uintptr_t>(res_hi) | reinterpret_cast<uintptr_t>(res_lo) |
         reinterpret_cast<uintptr_t>(packed_weight) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(bias) |
         reinterpret_cast<uintptr_t>(head)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward: C8 tensors and the per-channel vectors must be 16-byte aligned");
    if ((d->x_batch_stride | (res_f32 ? 0 : d->res_batch_stride)) % 8 || (!to_f32 && d->y_batch_stride % 8))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward: batch strides must be multiples of 8 elements");
    F16Args a{};
    a.x = reinterpret_cast<const _Float16 *>(x_hi); a.x_lo = reinterpret_cast<const _Float16 *>(x_lo);
    a.scale = scale; a.bias = bias;
    a.res = resflags ? reinterpret_cast<const _Float16 *>(res_hi) : nullptr;
    a.res_lo = resflags ? reinterpret_cast<const _Float16 *>(res_lo) : nullptr;
    a.y = reinterpret_cast<_Float16 *>(y_hi); a.y_lo = reinterpret_cast<_Float16 *>(y_lo); a.y_f32 = y_f32;
    a.head = head; a.y_head = y_head; a.head_mul = head_mul; a.res_mul = res_mul; a.overflow = overflow;
    a.res_f32 = res_f32;
    a.stats = stats;
    a.x_mul = to_f32 ? x_mul : nullptr;
    a.tail_w = reinterpret_cast<const _Float16 *>(tail_w); a.t_out = t_out; a.tail_mul = tail_mul;
    a.t_bs = (int64_t)27 * 8 * d->Din * d->Hin * d->Win;
    a.CGin = d->Cin / 8; a.Cout = d->Cout;
    a.Din = d->Din; a.Hin = d->Hin; a.Win = d->Win;
    a.Dout = d->Dout; a.Hout = d->Hout; a.Wout = d->Wout;
    a.nchunks = p.nchunks; a.flags = res_f32 ? (d->flags & ~SNVC_EPI_ADD_POST) : d->flags;
    a.x_bs = d->x_batch_stride ? d->x_batch_stride : 2 * (int64_t)d->Cin * in_sp;
    a.y_bs = d->y_batch_stride ? d->y_batch_stride : 2 * (int64_t)d->Cout * out_sp;
    a.r_bs = d->res_batch_stride ? d->res_batch_stride : 2 * (int64_t)d->Cout * out_sp;
    a.yf_bs = plane ? out_sp : (d->y_batch_stride ? d->y_batch_stride : (int64_t)d->Cout * out_sp);
    a.wp = reinterpret_cast<const _Float16 *>(packed_weight);
    const bool subgrid = p.kind == FK5D2X || p.kind == FK5D2XQ;        // (depth, height) parity classes share ONE packed weight block
    const int classes = d->transposed ? 8 : (subgrid ? 4 : 1);
    a.N = d->N; a.cls_mode = d->transposed ? 1 : (subgrid ? 2 : 0); a.cls_wstride = f16_class_stride(p);
    a.isd = a.ish = 1; a.iod = a.ioh = 0; a.offd = a.offh = a.offw = 0;
    if (d->transposed) {
        a.nd = d->Din; a.nh = d->Hin; a.nw = d->Win;
        a.osd = a.osh = a.osw = 2;
        a.pad_d = a.pad_h = a.pad_w = 0;
    } else if (subgrid) {
        a.nd = (d->Dout + 1) / 2; a.nh = (d->Hout + 1) / 2; a.nw = d->Wout;     // the largest class; the kernel trims the others
        a.osd = a.osh = 2; a.osw = 1;
        a.isd = a.ish = 2;
        a.pad_d = a.pad_h = d->pad / 2; a.pad_w = d->pad;
    } else {
        a.nd = d->Dout; a.nh = d->Hout; a.nw = d->Wout;
        a.osd = a.osh = a.osw = 1;
        a.pad_d = a.pad_h = a.pad_w = d->pad;
    }
    a.tiles_d = ceil_div(a.nd, p.TD); a.tiles_h = ceil_div(a.nh, p.TH); a.tiles_w = ceil_div(a.nw, 32);
    const int64_t ntiles = (int64_t)a.tiles_d * a.tiles_h * a.tiles_w;
    if (ntiles >= ((int64_t)1 << 30) || (int64_t)d->N * classes > 65535)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward: too many tiles or samples");
    dim3 grid((unsigned)(d->transposed ? 2 * ntiles : ntiles), (unsigned)p.cblocks, (unsigned)(d->N * (d->transposed ? 4 : classes)));
    hipStream_t st = as_stream(stream);
    if (stats_slots) {          // the statistics epilogue (r6): only the forms that write float32 through the shared epilogue, 256 threads
        const bool form_ok = to_f32 && !plane && !tail && (p.kind == FK3XQ || p.kind == FK3X || p.kind == FK3X2 || p.kind == FK3XS || p.kind == FK3X2S || p.kind == FK3XT ||
                                                            p.kind == FK3S2X || p.kind == FK3S2XT || p.kind == FDCX || p.kind == FDCXT || p.kind == FDCXN);
        if (!form_ok) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward_stats: this layer's kernel form has no statistics epilogue");
        *stats_slots = p.kind == FK3XQ ? ntiles * 4 : (int64_t)(grid.z / d->N) * grid.x * 4;
        if (!stats) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward_stats: null workspace");
        // whole tiles outside a smaller parity class return early: their slots must read as zero (one class: every workgroup writes its slots)
        if (classes > 1) {
            const hipError_t e = hipMemsetAsync(stats, 0, (size_t)d->N * *stats_slots * ((d->Cout + 31) / 32) * 64 * sizeof(double), st);
            if (e != hipSuccess) return fail(SNVC_ERR_HIP, "snvc_f16x3_conv3d_forward_stats: hipMemsetAsync failed");
        }
    }
#define SNVC_X3_LAUNCH(CFG) do { if (to_f32) launch_f16<CFG, 2>(a, grid, st); else launch_f16<CFG, 0>(a, grid, st); } while (0)
    switch (p.kind) {
        case FK3X:
            if (to_f32) launch_f16<F16K3X, 2>(a, grid, st);
            else if (head) launch_f16<F16K3X, 3>(a, grid, st);
            else launch_f16<F16K3X, 0>(a, grid, st);
            break;
        case FK3X2: SNVC_X3_LAUNCH(F16K3X2); break;
        case FK3XS:
            if (to_f32) launch_f16<F16K3XS, 2>(a, grid, st);
            else if (head) launch_f16<F16K3XS, 3>(a, grid, st);
            else launch_f16<F16K3XS, 0>(a, grid, st);
            break;
        case FK3X2S: SNVC_X3_LAUNCH(F16K3X2S); break;
        case FK3XT:
            if (to_f32) launch_f16<F16K3XT, 2>(a, grid, st);
            else if (head) launch_f16<F16K3XT, 3>(a, grid, st);
            else launch_f16<F16K3XT, 0>(a, grid, st);
            break;
        case FK3S2X: SNVC_X3_LAUNCH(F16K3S2X); break;
        case FK3S2XT: SNVC_X3_LAUNCH(F16K3S2XT); break;
        case FDCX: if (tail) launch_f16<F16DCX, 4>(a, grid, st); else SNVC_X3_LAUNCH(F16DCX); break;
        case FDCXT: if (tail) launch_f16<F16DCXT, 4>(a, grid, st); else SNVC_X3_LAUNCH(F16DCXT); break;
        case FDCXN: if (tail) launch_f16<F16DCXN, 4>(a, grid, st); else SNVC_X3_LAUNCH(F16DCXN); break;
        case FK1X: SNVC_X3_LAUNCH(F16K1X); break;
        case FK5X: SNVC_X3_LAUNCH(F16K5X); break;
        case FK5D2X: SNVC_X3_LAUNCH(F16K5D2X); break;
        case FK7X: SNVC_X3_LAUNCH(F16K7X); break;
        case FK3XH: launch_f16<F16K3X, 1>(a, grid, st); break;
        case FK5XQ: case FK5D2XQ: case FK7XQ: {
            if (to_f32 || head) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward: the 16x16x32 form writes a split C8 tensor, no side head");
#define SNVC_Q16S(CFG) do { static std::atomic<unsigned> at_{0};                                                                    \
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_q16s_kernel<CFG>), CFG::LDS_BYTES, at_))                \
                    conv3d_q16s_kernel<CFG><<<grid, 256, CFG::LDS_BYTES, st>>>(a); } while (0)
            using C5 = Q16SCfg<5, 1>; using C5D = Q16SCfg<5, 2>; using C7 = Q16SCfg<7, 1>;
            if (p.kind == FK5XQ) SNVC_Q16S(C5); else if (p.kind == FK5D2XQ) SNVC_Q16S(C5D); else SNVC_Q16S(C7);
#undef SNVC_Q16S
            break;
        }
        case FK3S2XQ: {
            if (to_f32 || resflags || head) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward: the stride-2 16x16x32 form writes a split C8 tensor, no residual / head");
            // persistent: one workgroup per CU (144 KB of LDS each), a multiple of 8 so that a workgroup's jobs b, b + G, ... stay on its XCD
            const int64_t total = ntiles * p.cblocks * d->N;
            if (total >= ((int64_t)1 << 30)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward: too many tiles");
            int g = device_cu_count();
            if (g <= 0) g = 256;
            if (total < g) g = total >= 8 ? (int)(total & ~(int64_t)7) : (int)total;
            static std::atomic<unsigned> attr_s2{0};
            if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_x3s2q_kernel), X3S2Cfg::LDS_BYTES, attr_s2))
                conv3d_x3s2q_kernel<<<dim3((unsigned)g), X3S2Cfg::THREADS, X3S2Cfg::LDS_BYTES, st>>>(a, (int)total);
            break;
        }
        case FK3XQ: {
            if (resflags || (to_f32 && head)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward: the 16x16x32 form has no split residual, no side head beside a float32 result");
            if (ntiles * p.cblocks >= ((int64_t)1 << 30)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward: too many tiles");
            grid = dim3((unsigned)(ntiles * p.cblocks), 1, (unsigned)d->N);      // (tile, channel block) jobs, channel block fastest
            static std::atomic<unsigned> attr_40{0}, attr_43{0}, attr_42{0};
            if (to_f32) {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_x3q_kernel<2>), X3QCfg::LDS_BYTES, attr_42))
                    conv3d_x3q_kernel<2><<<grid, 256, X3QCfg::LDS_BYTES, st>>>(a);
            } else if (head) {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_x3q_kernel<3>), X3QCfg::LDS_BYTES, attr_43))
                    conv3d_x3q_kernel<3><<<grid, 256, X3QCfg::LDS_BYTES, st>>>(a);
            } else {
                if (allow_large_lds(reinterpret_cast<const void *>(&conv3d_x3q_kernel<0>), X3QCfg::LDS_BYTES, attr_40))
                    conv3d_x3q_kernel<0><<<grid, 256, X3QCfg::LDS_BYTES, st>>>(a);
            }
            break;
        }
        default: return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward: no kernel");
    }
#undef SNVC_X3_LAUNCH
    return check_launch("snvc_f16x3_conv3d_forward");
}

int snvc_f16x3_conv3d_forward(const snvc_conv3d_desc *d, const void *x_hi, const void *x_lo, const void *packed_weight,
                              const float *scale, const float *bias, const void *res_hi, const void *res_lo, void *y_hi,
                              void *y_lo, float *y_f32, const float *head, float *y_head, float head_mul, float res_mul,
                              int *overflow, void *stream) {
    return f16x3_forward(d, x_hi, x_lo, packed_weight, scale, bias, res_hi, res_lo, y_hi, y_lo, y_f32, head, y_head, head_mul, res_mul,
                         overflow, nullptr, nullptr, 0.0f, stream);
}

// slots per sample of the statistics epilogue = (z classes) * gridDim.x * 4 waves, as f16x3_forward lays its grid out
static int64_t f16x3_stats_slots(const snvc_conv3d_desc &d, const snvc::F16Plan &p) {
    using namespace snvc;
    const bool subgrid = p.kind == FK5D2X || p.kind == FK5D2XQ;
    int nd, nh, nw;
    if (d.transposed) { nd = d.Din; nh = d.Hin; nw = d.Win; }
    else if (subgrid) { nd = (d.Dout + 1) / 2; nh = (d.Hout + 1) / 2; nw = d.Wout; }
    else { nd = d.Dout; nh = d.Hout; nw = d.Wout; }
    const int64_t ntiles = (int64_t)ceil_div(nd, p.TD) * ceil_div(nh, p.TH) * ceil_div(nw, 32);
    if (p.kind == FK3XQ) return ntiles * 4;              // (tile, channel block) jobs on gridDim.x: a slot per tile and wave
    return (d.transposed ? 4 * 2 * ntiles : (subgrid ? 4 : 1) * ntiles) * 4;
}

int64_t snvc_f16x3_conv3d_stats_workspace_bytes(const snvc_conv3d_desc *d) {
    using namespace snvc;
    F16Plan p;
    if (!d || d->N < 0 || make_f16_plan(*d, p, true)) return -1;
    const int64_t groups = (d->Cout + 31) / 32;
    const int64_t slots = f16x3_stats_slots(*d, p);
    return (d->N * slots * groups * 64 + conv_stats_fold_scratch_doubles(d->N, (int)groups, slots) + d->N * (int64_t)d->Cout * 2) * (int64_t)sizeof(double);
}

int snvc_f16x3_conv3d_forward_f32(const snvc_conv3d_desc *d, const void *x_hi, const void *x_lo, const void *packed_weight,
                                  const float *scale, const float *bias, const float *x_mul, const float *res_f32, float *y_f32,
                                  float out_mul, void *stream) {
    using namespace snvc;
    if (!d || !y_f32) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward_f32: null pointer");
    if (d->flags & (SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST)) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward_f32: the residual is res_f32 (added after the activation), no flag");
    snvc_conv3d_desc dd = *d;
    if (res_f32) dd.flags |= SNVC_EPI_ADD_POST;
    return f16x3_forward(&dd, x_hi, x_lo, packed_weight, scale, bias, res_f32, nullptr, nullptr, nullptr, y_f32, nullptr, nullptr, out_mul, 1.0f,
                         nullptr, nullptr, nullptr, 0.0f, stream, nullptr, nullptr, x_mul);
}

int snvc_f16x3_conv3d_forward_stats(const snvc_conv3d_desc *d, const void *x_hi, const void *x_lo, const void *packed_weight,
                                    const float *scale, const float *bias, const float *x_mul, float *y_f32, float head_mul, const float *gamma,
                                    const float *beta, float *bn_scale, float *bn_shift, float *mean, float *var, void *workspace,
                                    float eps, void *stream) {
    using namespace snvc;
    if (!d || !y_f32 || !bn_scale || !bn_shift || !workspace)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv3d_forward_stats: null pointer");
    if (d->N <= 0 || d->flags || d->Cout % 32) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv3d_forward_stats: whole 32-channel groups, no epilogue flags");
    double *stats = static_cast<double *>(workspace);
    int64_t slots = 0;
    int rc = f16x3_forward(d, x_hi, x_lo, packed_weight, scale, bias, nullptr, nullptr, nullptr, nullptr, y_f32, nullptr, nullptr, head_mul, 1.0f,
                           nullptr, nullptr, nullptr, 0.0f, stream, stats, &slots, x_mul);
    if (rc) return rc;
    const int groups = d->Cout / 32;
    double *scratch = stats + d->N * slots * groups * 64;
    double *partial = scratch + conv_stats_fold_scratch_doubles(d->N, groups, slots);
    launch_conv_stats_fold(stats, scratch, partial, d->N, d->Cout, groups, slots, as_stream(stream));
    rc = check_launch("snvc_f16x3_conv3d_forward_stats(fold)");
    if (rc) return rc;
    launch_norm_finalize(partial, gamma, beta, bn_scale, bn_shift, mean, var, d->N, d->Cout, (int64_t)d->Dout * d->Hout * d->Wout, 1, eps,
                         as_stream(stream));
    return check_launch("snvc_f16x3_conv3d_forward_stats(finalize)");
}

int snvc_f16x3_split_scale(const float *x, int64_t n, void *scratch8, float *out_mul, void *stream) {
    using namespace snvc;
    if (n < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_split_scale: negative size");
    if (!out_mul || !scratch8 || (n > 0 && !x)) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_split_scale: null pointer");
    if (reinterpret_cast<uintptr_t>(x) & 15) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_split_scale: x must be 16-byte aligned");
    int blocks = (int)ceil_div<int64_t>(n > 0 ? n : 1, 256 * 4 * 16);
    blocks = blocks < 1 ? 1 : (blocks > 48 ? 48 : blocks);
    split_scale_kernel<<<blocks, 256, 0, as_stream(stream)>>>(x, n, reinterpret_cast<unsigned *>(scratch8), out_mul);
    return check_launch("snvc_f16x3_split_scale");
}

// ---- depth-1 (2D) split layers: 3x7 and 3x3, stride 1, "same" padding, Cin % 8 == 0, Cout % 32 == 0
static int x2q_nq(int kh, int kw) { return (kh * kw + 3) / 4; }

int64_t snvc_f16x3_conv2d_packed_weight_bytes(int cout, int cin, int kh, int kw) {
    if (cout <= 0 || cin <= 0 || cout % 32 || cin % 8 || kh != 3 || (kw != 3 && kw != 7)) return -1;
    return 2 * ((int64_t)(cout / 32) * (cin / 8) * x2q_nq(kh, kw) + 2) * 4 * 64 * 8;      // + the weight ring's read-ahead
}

int snvc_f16x3_conv2d_pack_weights(const float *weight, int cout, int cin, int kh, int kw, void *packed, float wmul, void *stream) {
    using namespace snvc;
    const int64_t bytes = snvc_f16x3_conv2d_packed_weight_bytes(cout, cin, kh, kw);
    if (bytes < 0) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv2d: 3x7 or 3x3 kernels, Cin % 8 == 0, Cout % 32 == 0");
    if (!weight || !packed) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv2d_pack_weights: null pointer");
    if (hipMemsetAsync(packed, 0, (size_t)bytes, as_stream(stream)) != hipSuccess)
        return fail(SNVC_ERR_HIP, "snvc_f16x3_conv2d_pack_weights: hipMemsetAsync failed");
    const int nq = x2q_nq(kh, kw);
    const int64_t total = (int64_t)(cout / 32) * (cin / 8) * nq * 4 * 64 * 8;
    // [cb][chunk][quad][co half][hi | lo][lane][8] over the (kh, kw) raster: the 3x3x3 form's packer with K3 = kh * kw taps
    pack_q16s_weights_kernel<<<(unsigned)ceil_div<int64_t>(total, 256), 256, 0, as_stream(stream)>>>(
        weight, reinterpret_cast<_Float16 *>(packed), cout, cin, kh * kw, 1, 4 * nq, nq, cin / 8, 2, 2, 1, wmul, total);
    return check_launch("snvc_f16x3_conv2d_pack_weights");
}

int snvc_f16x3_conv2d_forward(const void *x_hi, const void *x_lo, const void *packed_weight, const float *scale, const float *bias,
                              float *y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int kh, int kw, float out_mul,
                              const float *x_mul_dev, int flags, void *stream) {
    using namespace snvc;
    if (N < 0 || H <= 0 || W <= 0 || snvc_f16x3_conv2d_packed_weight_bytes((int)Cout, (int)Cin, kh, kw) < 0)
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv2d_forward: 3x7 or 3x3 kernels, Cin % 8 == 0, Cout % 32 == 0");
    if (N == 0) return SNVC_OK;
    if (!x_hi || !x_lo || !packed_weight || !y) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv2d_forward: null pointer");
    if ((scale == nullptr) != (bias == nullptr))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv2d_forward: scale and bias must both be given or both be NULL");
    if (flags & ~SNVC_EPI_RELU) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv2d_forward: only SNVC_EPI_RELU");
    if ((reinterpret_cast<uintptr_t>(x_hi) | reinterpret_cast<uintptr_t>(x_lo) | reinterpret_cast<uintptr_t>(packed_weight)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv2d_forward: C8 tensors must be 16-byte aligned");
    const int64_t hw = H * W;
    if ((Cin / 8 + 2) * hw >= ((int64_t)1 << 31) || N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv2d_forward: tensor too large");
    F16Args a{};
    a.x = reinterpret_cast<const _Float16 *>(x_hi); a.x_lo = reinterpret_cast<const _Float16 *>(x_lo);
    a.wp = reinterpret_cast<const _Float16 *>(packed_weight);
    a.scale = scale; a.bias = bias; a.y_f32 = y; a.head_mul = out_mul; a.x_mul = x_mul_dev;
    a.CGin = (int)(Cin / 8); a.Cout = (int)Cout; a.nchunks = (int)(Cin / 8); a.flags = flags;
    a.Din = a.Dout = 1; a.Hin = a.Hout = (int)H; a.Win = a.Wout = (int)W;
    a.nd = 1; a.nh = (int)H; a.nw = (int)W;
    a.pad_d = 0; a.pad_h = (kh - 1) / 2; a.pad_w = (kw - 1) / 2;
    a.tiles_d = 1; a.tiles_h = (int)ceil_div<int64_t>(H, 16); a.tiles_w = (int)ceil_div<int64_t>(W, 32);
    a.x_bs = 2 * Cin * hw; a.yf_bs = Cout * hw; a.N = (int)N;
    const int64_t jobs = (int64_t)a.tiles_h * a.tiles_w * (Cout / 32);
    if (jobs >= ((int64_t)1 << 30)) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv2d_forward: too many tiles");
    const dim3 grid((unsigned)jobs, 1, (unsigned)N);
    hipStream_t st = as_stream(stream);
    if (kw == 7) {
        using C = X2QCfg<3, 7>;
        static std::atomic<unsigned> at{0};
        if (allow_large_lds(reinterpret_cast<const void *>(&conv2d_x3q_kernel<C>), C::LDS_BYTES, at)) conv2d_x3q_kernel<C><<<grid, 256, C::LDS_BYTES, st>>>(a);
    } else {
        using C = X2QCfg<3, 3>;
        static std::atomic<unsigned> at{0};
        if (allow_large_lds(reinterpret_cast<const void *>(&conv2d_x3q_kernel<C>), C::LDS_BYTES, at)) conv2d_x3q_kernel<C><<<grid, 256, C::LDS_BYTES, st>>>(a);
    }
    return check_launch("snvc_f16x3_conv2d_forward");
}

// The sheared first layer's whole 2D prep in ONE host call (five launches: scale, Rq -> G, Rq' -> G'): behind the step's one host
// sync the GPU is empty, and five Python-level launches of 5-20 us kernels starve it (measured: +0.08-0.14 ms per step against
// the same kernels queued by a host that runs ahead).
extern "C" int snvc_sheared_upsample_split(const float *, void *, void *, const float *, int64_t, int64_t, int64_t, int64_t, int, int64_t, int, void *);
int snvc_sheared_prep_x3(const float *right, int64_t N, int64_t C, int64_t H, int64_t W, int q, int64_t WU, int off, int64_t WU2, int off2,
                         const void *packed_g, const void *packed_col, int64_t Cout3, float out_mul_g, float out_mul_col, void *rq_split,
                         void *rq2_split, void *scratch8, float *mul_dev, float *g, float *gcol, void *stream) {
    using namespace snvc;
    if (!right || !packed_g || !packed_col || !rq_split || !rq2_split || !scratch8 || !mul_dev || !g || !gcol)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_sheared_prep_x3: null pointer");
    if (N <= 0 || C % 8 || Cout3 % 32) return fail(SNVC_ERR_UNSUPPORTED, "snvc_sheared_prep_x3: C % 8 == 0, 3 * Cout % 32 == 0");
    int rc = snvc_f16x3_split_scale(right, N * C * H * W, scratch8, mul_dev, stream);
    if (rc) return rc;
    const int64_t G = C / 8;
    _Float16 *a = reinterpret_cast<_Float16 *>(rq_split), *b = reinterpret_cast<_Float16 *>(rq2_split);
    rc = snvc_sheared_upsample_split(right, a, a + G * H * WU * 8, mul_dev, N, C, H, W, q, WU, off, stream);
    if (rc) return rc;
    rc = snvc_f16x3_conv2d_forward(a, a + G * H * WU * 8, packed_g, nullptr, nullptr, g, N, C, Cout3, H, WU, 3, 7, out_mul_g, mul_dev, 0, stream);
    if (rc) return rc;
    rc = snvc_sheared_upsample_split(right, b, b + G * H * WU2 * 8, mul_dev, N, C, H, W, q, WU2, off2, stream);
    if (rc) return rc;
    return snvc_f16x3_conv2d_forward(b, b + G * H * WU2 * 8, packed_col, nullptr, nullptr, gcol, N, C, Cout3, H, WU2, 3, 7, out_mul_col, mul_dev, 0,
                                     stream);
}

// n_layers depth-1 split layers of ONE fp32 [N][C][H][W] input in one host call: its scale, its split pair (ws: N*2*C*H*W halves), then
// the layers (same kernel size, their own packed weights / output channels / results).  The left half's depth-class planes (one
// 3x3 layer) and the any-shift path's P / Q layers (two 3x3 layers of the right feature) go through here.
extern "C" int snvc_f16x3_from_ncdhw(const float *, void *, void *, int64_t, int64_t, int64_t, int64_t, int64_t, float, const float *, void *);
int snvc_f16x3_conv2d_from_f32(const float *x, int64_t N, int64_t C, int64_t H, int64_t W, int kh, int kw, int n_layers,
                               const void *const *packed, const int64_t *cout, const float *out_mul, float *const *y, void *ws_split,
                               void *scratch8, float *mul_dev, void *stream) {
    using namespace snvc;
    if (!x || !packed || !cout || !out_mul || !y || !ws_split || !scratch8 || !mul_dev || n_layers < 1 || n_layers > 8)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_conv2d_from_f32: null pointer or layer count outside 1..8");
    if (N <= 0 || C % 8) return fail(SNVC_ERR_UNSUPPORTED, "snvc_f16x3_conv2d_from_f32: C % 8 == 0");
    int rc = snvc_f16x3_split_scale(x, N * C * H * W, scratch8, mul_dev, stream);
    if (rc) return rc;
    _Float16 *a = reinterpret_cast<_Float16 *>(ws_split);
    rc = snvc_f16x3_from_ncdhw(x, a, a + C * H * W, N, C, H * W, 0, 0, 1.0f, mul_dev, stream);
    if (rc) return rc;
    for (int i = 0; i < n_layers; ++i) {
        rc = snvc_f16x3_conv2d_forward(a, a + C * H * W, packed[i], nullptr, nullptr, y[i], N, C, cout[i], H, W, kh, kw, out_mul[i], mul_dev, 0, stream);
        if (rc) return rc;
    }
    return SNVC_OK;
}

int64_t snvc_f16x3_tail_packed_weight_bytes(int cin) {
    return cin > 0 && cin % 32 == 0 ? (int64_t)(cin / 32) * 2 * 2 * 64 * 8 * 2 : -1;
}

int snvc_f16x3_tail_pack_weights(const float *weight, int cin, void *packed, float wmul, void *stream) {
    using namespace snvc;
    if (!weight || !packed || cin <= 0 || cin % 32 != 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_tail_pack_weights: weight [Cin][27] with Cin % 32 == 0");
    const int total = (cin / 32) * 2 * 2 * 64 * 8;
    pack_tail_weights_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, as_stream(stream)>>>(weight, reinterpret_cast<_Float16 *>(packed), cin, wmul, total);
    return check_launch("snvc_f16x3_tail_pack_weights");
}

int snvc_f16x3_deconv3d_tail_forward(const snvc_conv3d_desc *d, const void *x_hi, const void *x_lo, const void *packed_weight,
                                     const float *scale, const float *bias, const void *res_hi, const void *res_lo, float res_mul,
                                     const void *tail_packed, float *t_out, float tail_mul, int *overflow, void *stream) {
    using namespace snvc;
    if (!tail_packed || !t_out) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_f16x3_deconv3d_tail_forward: null pointer");
    return f16x3_forward(d, x_hi, x_lo, packed_weight, scale, bias, res_hi, res_lo, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0f, res_mul,
                         overflow, tail_packed, t_out, tail_mul, stream);
}

int snvc_deconv_tail_gather(const float *t, const float *bias, const float *residual, float *y, int64_t n, int64_t nd, int64_t nh,
                            int64_t nw, void *stream) {
    using namespace snvc;
    if (n < 0 || nd <= 0 || nh <= 0 || nw <= 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_deconv_tail_gather: sizes must be positive");
    if (n == 0) return SNVC_OK;
    if (!t || !y) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_deconv_tail_gather: null pointer");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_deconv_tail_gather: y / residual must be 16-byte aligned");
    const int64_t threads = 16 * nd * nh * nw, out_vox = 64 * nd * nh * nw;
    if (threads >= ((int64_t)1 << 31) * 255 || n > 65535 || 27 * 8 * nd * nh * nw >= ((int64_t)1 << 40))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_deconv_tail_gather: tensor too large");
    const dim3 grid((unsigned)ceil_div<int64_t>(threads, 256), (unsigned)n);
    deconv_tail_gather_kernel<<<grid, 256, 0, as_stream(stream)>>>(t, bias, residual, y, (int)nd, (int)nh, (int)nw, 27 * 8 * nd * nh * nw,
                                                                   out_vox, out_vox);
    return check_launch("snvc_deconv_tail_gather");
}

}  // extern "C"
