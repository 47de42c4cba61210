// Deterministic adjoint of the feature -> voxel gather (training step).
//
// Reference: torch autograd through F.grid_sample (grid_sampler_2d_backward scatters with float atomics, as the first
// version here did: order-dependent, and 77 ms when 64 lanes hit one pixel).  This version has no atomics:
//   1. every voxel gets the key of its BASE pixel (x0, y0) = floor of its sample position (all four bilinear taps
//      hang off it), per (sample, camera) group;
//   2. a stable radix sort (hipCUB) of (key, voxel) pairs groups the voxels by base pixel, voxel index ascending;
//   3. every segment is cut into pieces of <= 1024 voxels (an exclusive scan of the piece counts gives each piece its
//      slot, so one pixel that collects a million voxels becomes a thousand workgroups, not one); a workgroup walks its
//      piece in sorted order, 64 voxels at a time: the gradient tile [64 voxels][F channels] goes through LDS, thread
//      (channel, tap) adds its 64 products in voxel order -> P[slot][tap][channel];
//   4. grad[c, pixel] = sum over the four taps (nw, ne, sw, se) of the slots of that tap's base pixel, slots ascending.
// Every sum has one fixed order -> the result is bit-reproducible run to run, and the gradient volume is read once.
// Weights are the forward kernel's own (same make_taps arithmetic, fp contract off).
#include <hipcub/hipcub.hpp>

#include "common.hpp"

namespace snvc {
namespace {

#pragma clang fp contract(off)

struct BaseTap {
    int x0, y0;      // base pixel (may be -1: taps x0+1 / y0+1 still inside)
    float wt[4];     // nw, ne, sw, se
    bool any;        // at least one tap inside the map
};

// same arithmetic as make_taps (voxel_gather.hip): vernier.py:335-338 normalisation, ATen's unnormalise / floor / weights
__device__ __forceinline__ BaseTap base_tap(float px, float py, float res_x, float res_y, int Hf, int Wf) {
    const float gx = px / res_x * 2.0f - 1.0f;
    const float gy = py / res_y * 2.0f - 1.0f;
    const float x = (gx + 1.0f) * ((float)Wf / 2.0f) - 0.5f;
    const float y = (gy + 1.0f) * ((float)Hf / 2.0f) - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float w = x - xf, e = 1.0f - w, n = y - yf, s = 1.0f - n;
    BaseTap t;
    t.wt[0] = s * e; t.wt[1] = s * w; t.wt[2] = n * e; t.wt[3] = n * w;
    const bool fin = xf >= -2.0f && xf <= (float)Wf + 1.0f && yf >= -2.0f && yf <= (float)Hf + 1.0f;
    t.x0 = fin ? (int)xf : -2; t.y0 = fin ? (int)yf : -2;
    t.any = t.x0 >= -1 && t.x0 <= Wf - 1 && t.y0 >= -1 && t.y0 <= Hf - 1;
    return t;
}

// key of group g (= side * N + n), base (x0, y0): g * (K + 1) + (y0 + 1) * (Wf + 1) + (x0 + 1); K = (Hf+1)*(Wf+1);
// voxels without a tap inside the map take the group's last key K (never read back)
__global__ void __launch_bounds__(256)
gather_bwd_keys_kernel(const float *__restrict__ l_pts, const float *__restrict__ r_pts, unsigned *__restrict__ keys,
                       unsigned *__restrict__ vals, int N, int Hf, int Wf, int64_t V, float res_x, float res_y) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int g = blockIdx.y, side = g / N, n = g - side * N;
    const float *pts = (side == 0 ? l_pts : r_pts) + (int64_t)n * 2 * V;
    const BaseTap t = base_tap(pts[v], pts[V + v], res_x, res_y, Hf, Wf);
    const unsigned K = (unsigned)(Hf + 1) * (Wf + 1);
    const unsigned local = t.any ? (unsigned)(t.y0 + 1) * (Wf + 1) + (unsigned)(t.x0 + 1) : K;
    keys[(int64_t)g * V + v] = (unsigned)g * (K + 1) + local;
    vals[(int64_t)g * V + v] = (unsigned)v;
}

// start[k] = first sorted position whose key is >= k, for k in [0, nkeys]; M = number of pairs
__global__ void __launch_bounds__(256)
segment_starts_kernel(const unsigned *__restrict__ skeys, unsigned *__restrict__ start, int64_t M, unsigned nkeys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    const int64_t prev = i == 0 ? -1 : (int64_t)skeys[i - 1];
    const int64_t cur = i == M ? (int64_t)nkeys : (int64_t)skeys[i];
    for (int64_t k = prev + 1; k <= cur; ++k) start[k] = (unsigned)i;
}

constexpr int kPiece = 1024;     // voxels per workgroup of the segment pass

// npieces[k] = ceil(len_k / kPiece) for base-pixel keys, 0 for a group's "no tap inside" bucket
__global__ void __launch_bounds__(256)
piece_counts_kernel(const unsigned *__restrict__ start, unsigned *__restrict__ npieces, unsigned nkeys, unsigned K) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nkeys) return;
    unsigned c = 0;
    if (k < nkeys && k % (K + 1) != K) c = (start[k + 1] - start[k] + kPiece - 1) / kPiece;
    npieces[k] = c;
}

// One workgroup per piece.  tile[j][c]: gradient of channel c at the j-th voxel of the batch; thread
// (c, t) = (tid % FP, tid / FP) sums wt[j][t] * tile[j][c] over j ascending.
template <int FP>   // channels handled per pass: 64 (256 threads = 64 channels x 4 taps)
__global__ void __launch_bounds__(256)
gather_bwd_pieces_kernel(const float *__restrict__ grad_out, const float *__restrict__ l_pts, const float *__restrict__ r_pts,
                         const unsigned *__restrict__ start, const unsigned *__restrict__ slot0, const unsigned *__restrict__ svals,
                         float *__restrict__ P, unsigned nkeys, int N, int F, int Hf, int Wf, int64_t V, float res_x, float res_y) {
    __shared__ float tile[64][FP + 1];
    __shared__ float wts[64][4];
    __shared__ unsigned vox[64];
    __shared__ unsigned key_s;
    const unsigned slot = blockIdx.x;
    if (slot >= slot0[nkeys]) return;                      // the grid is an upper bound of the piece count (block-uniform exit)
    const int tid = threadIdx.x;
    if (tid == 0) {                                        // the key whose slot range [slot0[k], slot0[k+1]) holds this slot
        unsigned lo = 0, hi = nkeys;                       // invariant: slot0[lo] <= slot < slot0[hi]
        while (hi - lo > 1) {
            const unsigned mid = (lo + hi) >> 1;
            if (slot0[mid] <= slot) lo = mid; else hi = mid;
        }
        key_s = lo;
    }
    __syncthreads();
    const unsigned gkey = key_s;
    const unsigned K = (unsigned)(Hf + 1) * (Wf + 1);
    const int g = (int)(gkey / (K + 1)), side = g / N, n = g - side * N;
    const unsigned piece = slot - slot0[gkey];
    const unsigned s0 = start[gkey] + piece * kPiece;
    const unsigned send = start[gkey + 1];
    const unsigned s1 = s0 + kPiece < send ? s0 + kPiece : send;
    const int c_of = tid % FP, t_of = tid / FP;
    const float *pts = (side == 0 ? l_pts : r_pts) + (int64_t)n * 2 * V;
    const float *gbase = grad_out + ((int64_t)n * 2 * F + (int64_t)side * F) * V;
    float *Ps = P + (int64_t)slot * 4 * F;
    for (int c0 = 0; c0 < F; c0 += FP) {
        float acc = 0.0f;
        for (unsigned b = s0; b < s1; b += 64) {
            const int cnt = (int)(s1 - b < 64u ? s1 - b : 64u);
            __syncthreads();                               // previous batch fully consumed
            if (tid < 64 && tid < cnt) {
                const unsigned v = svals[b + tid];
                vox[tid] = v;
                const BaseTap t = base_tap(pts[v], pts[V + v], res_x, res_y, Hf, Wf);
#pragma unroll
                for (int k = 0; k < 4; ++k) wts[tid][k] = t.wt[k];
            }
            __syncthreads();
            // lanes = voxels of the batch (runs of consecutive voxel indices inside a segment read coalesced)
            for (int c = tid >> 6; c < FP; c += 4) {
                const int j = tid & 63;
                if (j < cnt && c0 + c < F) tile[j][c] = gbase[(int64_t)(c0 + c) * V + vox[j]];
            }
            __syncthreads();
            if (t_of < 4 && c0 + c_of < F)
                for (int j = 0; j < cnt; ++j) acc += wts[j][t_of] * tile[j][c_of];
        }
        if (t_of < 4 && c0 + c_of < F) Ps[t_of * F + c0 + c_of] = acc;
    }
}

// grad[n, c, py, px] = sum over the four taps that can land on (py, px), in the order nw, ne, sw, se; per tap the
// pieces of its base pixel in slot order
__global__ void __launch_bounds__(256)
gather_bwd_combine_kernel(const float *__restrict__ P, const unsigned *__restrict__ slot0, float *__restrict__ grad_left,
                          float *__restrict__ grad_right, int N, int F, int Hf, int Wf) {
    const int plane = Hf * Wf;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (c, pixel) of one group
    if (i >= (int64_t)F * plane) return;
    const int g = blockIdx.y, side = g / N, n = g - side * N;
    const int p = (int)(i % plane), c = (int)(i / plane);
    const int py = p / Wf, px = p - py * Wf;
    const unsigned K = (unsigned)(Hf + 1) * (Wf + 1);
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int dy = t >> 1, dx = t & 1;                  // tap t of base (py - dy, px - dx) is this pixel
        const unsigned gkey = (unsigned)g * (K + 1) + (unsigned)(py - dy + 1) * (Wf + 1) + (unsigned)(px - dx + 1);
        for (unsigned j = slot0[gkey]; j < slot0[gkey + 1]; ++j) s += P[((int64_t)j * 4 + t) * F + c];
    }
    (side == 0 ? grad_left : grad_right)[((int64_t)n * F + c) * plane + p] = s;
}

struct BwdLayout {
    int64_t M, nkeys, max_pieces, off_keys_in, off_keys_out, off_vals_in, off_vals_out, off_start, off_np, off_slot0, off_P, off_tmp,
        tmp_bytes, total;
};

inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

int bwd_layout(int64_t N, int64_t F, int64_t Hf, int64_t Wf, int64_t V, BwdLayout &L) {
    L.M = 2 * N * V;
    const int64_t K = (Hf + 1) * (Wf + 1);
    L.nkeys = 2 * N * (K + 1);
    if (L.nkeys >= ((int64_t)1 << 31) || V >= ((int64_t)1 << 32) || L.M >= ((int64_t)1 << 31))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_voxel_gather_backward_det: too many voxels or pixels for 32-bit keys");
    size_t tmp = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const unsigned *)nullptr, (unsigned *)nullptr, (const unsigned *)nullptr,
                                           (unsigned *)nullptr, (int)L.M, 0, 32) != hipSuccess)
        return fail(SNVC_ERR_HIP, "snvc_voxel_gather_backward_det: hipcub temp-storage query failed");
    size_t tmp2 = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, (const unsigned *)nullptr, (unsigned *)nullptr, (int)(L.nkeys + 1)) != hipSuccess)
        return fail(SNVC_ERR_HIP, "snvc_voxel_gather_backward_det: hipcub temp-storage query failed");
    L.tmp_bytes = (int64_t)(tmp > tmp2 ? tmp : tmp2);
    L.max_pieces = L.M / kPiece + L.nkeys;          // sum of ceil(len / kPiece) <= M / kPiece + (number of keys)
    int64_t o = 0;
    L.off_keys_in = o; o += align256(L.M * 4);
    L.off_keys_out = o; o += align256(L.M * 4);
    L.off_vals_in = o; o += align256(L.M * 4);
    L.off_vals_out = o; o += align256(L.M * 4);
    L.off_start = o; o += align256((L.nkeys + 1) * 4);
    L.off_np = o; o += align256((L.nkeys + 1) * 4);
    L.off_slot0 = o; o += align256((L.nkeys + 1) * 4);
    L.off_P = o; o += align256(L.max_pieces * 4 * F * 4);
    L.off_tmp = o; o += align256(L.tmp_bytes);
    L.total = o;
    return SNVC_OK;
}

}  // namespace
}  // namespace snvc

extern "C" {

int64_t snvc_voxel_gather_backward_workspace_bytes(int64_t N, int64_t F, int64_t Hf, int64_t Wf, int64_t V) {
    using namespace snvc;
    if (N <= 0 || F <= 0 || Hf <= 0 || Wf <= 0 || V <= 0) return 256;
    BwdLayout L;
    if (bwd_layout(N, F, Hf, Wf, V, L) != SNVC_OK) return -1;
    return L.total;
}

int snvc_voxel_gather_backward_det(const float *grad_out, const float *l_pts, const float *r_pts, float *grad_left,
                                   float *grad_right, void *workspace, int64_t N, int64_t F, int64_t Hf, int64_t Wf,
                                   int64_t V, float res_x, float res_y, void *stream) {
    using namespace snvc;
    if (N < 0 || F < 0 || Hf < 0 || Wf < 0 || V < 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_backward_det: negative size");
    hipStream_t st = as_stream(stream);
    const int64_t plane = Hf * Wf;
    if (N * F * plane == 0) return SNVC_OK;
    if (!grad_left || !grad_right) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_backward_det: null pointer");
    if (V == 0) {
        if (hipMemsetAsync(grad_left, 0, sizeof(float) * N * F * plane, st) != hipSuccess ||
            hipMemsetAsync(grad_right, 0, sizeof(float) * N * F * plane, st) != hipSuccess)
            return fail(SNVC_ERR_HIP, "snvc_voxel_gather_backward_det: hipMemsetAsync failed");
        return SNVC_OK;
    }
    if (!grad_out || !l_pts || !r_pts || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_voxel_gather_backward_det: null pointer or workspace not 256-byte aligned");
    if (2 * N > 65535 || plane >= ((int64_t)1 << 24))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_voxel_gather_backward_det: batch or feature plane too large");
    BwdLayout L;
    int rc = bwd_layout(N, F, Hf, Wf, V, L);
    if (rc) return rc;
    char *ws = reinterpret_cast<char *>(workspace);
    unsigned *keys_in = reinterpret_cast<unsigned *>(ws + L.off_keys_in), *keys_out = reinterpret_cast<unsigned *>(ws + L.off_keys_out);
    unsigned *vals_in = reinterpret_cast<unsigned *>(ws + L.off_vals_in), *vals_out = reinterpret_cast<unsigned *>(ws + L.off_vals_out);
    unsigned *start = reinterpret_cast<unsigned *>(ws + L.off_start), *npieces = reinterpret_cast<unsigned *>(ws + L.off_np);
    unsigned *slot0 = reinterpret_cast<unsigned *>(ws + L.off_slot0);
    float *P = reinterpret_cast<float *>(ws + L.off_P);
    const int G = (int)(2 * N);
    gather_bwd_keys_kernel<<<dim3((unsigned)ceil_div<int64_t>(V, 256), (unsigned)G), 256, 0, st>>>(l_pts, r_pts, keys_in, vals_in, (int)N,
                                                                                                 (int)Hf, (int)Wf, V, res_x, res_y);
    rc = check_launch("snvc_voxel_gather_backward_det(keys)");
    if (rc) return rc;
    int bits = 1;
    while (((int64_t)1 << bits) < L.nkeys) ++bits;
    size_t tmp = (size_t)L.tmp_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(ws + L.off_tmp, tmp, keys_in, keys_out, vals_in, vals_out, (int)L.M, 0, bits, st) != hipSuccess)
        return fail(SNVC_ERR_HIP, "snvc_voxel_gather_backward_det: radix sort failed");
    segment_starts_kernel<<<(unsigned)ceil_div<int64_t>(L.M + 1, 256), 256, 0, st>>>(keys_out, start, L.M, (unsigned)L.nkeys);
    const unsigned K = (unsigned)((Hf + 1) * (Wf + 1));
    piece_counts_kernel<<<(unsigned)ceil_div<int64_t>(L.nkeys + 1, 256), 256, 0, st>>>(start, npieces, (unsigned)L.nkeys, K);
    tmp = (size_t)L.tmp_bytes;
    if (hipcub::DeviceScan::ExclusiveSum(ws + L.off_tmp, tmp, npieces, slot0, (int)(L.nkeys + 1), st) != hipSuccess)
        return fail(SNVC_ERR_HIP, "snvc_voxel_gather_backward_det: scan failed");
    gather_bwd_pieces_kernel<64><<<dim3((unsigned)L.max_pieces), 256, 0, st>>>(grad_out, l_pts, r_pts, start, slot0, vals_out, P,
                                                                               (unsigned)L.nkeys, (int)N, (int)F, (int)Hf, (int)Wf, V, res_x,
                                                                               res_y);
    gather_bwd_combine_kernel<<<dim3((unsigned)ceil_div<int64_t>(F * plane, 256), (unsigned)G), 256, 0, st>>>(P, slot0, grad_left, grad_right,
                                                                                                              (int)N, (int)F, (int)Hf, (int)Wf);
    return check_launch("snvc_voxel_gather_backward_det");
}

}  // extern "C"
