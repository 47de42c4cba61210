// Plane-sweep volume -> 3D grid resampling for gfx950.
//
// The global scene model that consumes build_cost_volume is not in the public reference
// (snvc/models/__init__.py:1-2); what is shipped are its helpers: project_rect_to_image /
// project_image_to_rect / project_disp_to_depth_new (snvc/utils/torch_utils.py:5-45) and
// disparityregression (snvc/models/submodule.py:76-83).  The resampling step those helpers serve --
// features of a plane-sweep volume [N,C,D,H,W] (D = depth planes, H x W = image) sampled at the
// image projections of 3D voxel centres -- is therefore defined here against the operator the
// DSGN lineage uses for it, 5-D F.grid_sample (trilinear, zeros padding).  PARITY UNPINNED by the
// reference (no caller, no test upstream); the oracle is torch's grid_sample itself (tests/test_gpu_parity.py).
//
//   snvc_rect_to_psv_grid : project_rect_to_image (torch_utils.py:37-45) + normalisation of (u, v, depth) to
//                           grid_sample's [-1, 1] cube, one thread per 3D point
//   snvc_volume_resample  : out[n,c,v] = trilinear sample of x[n,c] at grid[n,v] = (gx -> W, gy -> H, gz -> D);
//                           arithmetic follows ATen's CPU grid_sampler_3d (unnormalise, floor, eight corner weights as
//                           products of three differences, in-bounds corners accumulated in the order tnw, tne, tsw,
//                           tse, bnw, bne, bsw, bse), fp contract off
#include "common.hpp"

namespace snvc {
namespace {

#pragma clang fp contract(off)

__device__ __forceinline__ float unnormalize(float c, int size, bool align) {
    return align ? ((c + 1.0f) / 2.0f) * (float)(size - 1) : ((c + 1.0f) * (float)size - 1.0f) / 2.0f;
}

__global__ void __launch_bounds__(256)
volume_resample_kernel(const float *__restrict__ x, const float *__restrict__ grid, float *__restrict__ out, int C, int D,
                       int H, int W, int64_t V, int align, int64_t x_bs, int64_t o_bs) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int64_t n = blockIdx.y;
    const float *g = grid + (n * V + v) * 3;
    const float ix = unnormalize(g[0], W, align != 0), iy = unnormalize(g[1], H, align != 0), iz = unnormalize(g[2], D, align != 0);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    // corner (dz, dy, dx) carries weight |opposite corner - point| per axis (ATen's expressions, same operand order)
    const float wx0 = (fx + 1.0f) - ix, wx1 = ix - fx, wy0 = (fy + 1.0f) - iy, wy1 = iy - fy, wz0 = (fz + 1.0f) - iz, wz1 = iz - fz;
    const bool fin = fx >= -2.0f && fx <= (float)W + 1.0f && fy >= -2.0f && fy <= (float)H + 1.0f && fz >= -2.0f && fz <= (float)D + 1.0f;
    const int x0 = fin ? (int)fx : -2, y0 = fin ? (int)fy : -2, z0 = fin ? (int)fz : -2;
    float wt[8];
    int off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;       // order tnw, tne, tsw, tse, bnw, bne, bsw, bse
        const int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
        const bool ok = xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D;
        off[k] = ok ? (zz * H + yy) * W + xx : -1;
        wt[k] = ((dx ? wx1 : wx0) * (dy ? wy1 : wy0)) * (dz ? wz1 : wz0);
    }
    const int64_t dhw = (int64_t)D * H * W;
    const float *xn = x + n * x_bs;
    float *o = out + n * o_bs + v;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float *p = xn + c * dhw;
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = p[off[k] < 0 ? 0 : off[k]];
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (off[k] >= 0) acc += t[k] * wt[k];
        o[(int64_t)c * V] = acc;
    }
}

// pts_rect [V,3] (x, y, z in the rectified camera frame) -> grid [V,3] for snvc_volume_resample:
//   (u, v) = project_rect_to_image (torch_utils.py:37-45: [x y z 1] P^T, divide by the third column),
//   gx = (u - u0) / (u_span) * 2 - 1, gy likewise, gz = (z - z0) / z_span * 2 - 1.
__global__ void __launch_bounds__(256)
rect_to_psv_grid_kernel(const float *__restrict__ pts, float *__restrict__ grid, int64_t V, float p00, float p01, float p02,
                        float p03, float p10, float p11, float p12, float p13, float p20, float p21, float p22, float p23,
                        float u0, float u_span, float v0, float v_span, float z0, float z_span) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const float X = pts[3 * i], Y = pts[3 * i + 1], Z = pts[3 * i + 2];
    const float a = ((X * p00 + Y * p01) + Z * p02) + p03;       // torch.mm row: sum in column order
    const float b = ((X * p10 + Y * p11) + Z * p12) + p13;
    const float c = ((X * p20 + Y * p21) + Z * p22) + p23;
    const float u = a / c, v = b / c;
    grid[3 * i] = (u - u0) / u_span * 2.0f - 1.0f;
    grid[3 * i + 1] = (v - v0) / v_span * 2.0f - 1.0f;
    grid[3 * i + 2] = (Z - z0) / z_span * 2.0f - 1.0f;
}

}  // namespace
}  // namespace snvc

extern "C" {

int snvc_volume_resample(const float *x, const float *grid, float *out, int64_t N, int64_t C, int64_t D, int64_t H,
                         int64_t W, int64_t V, int align_corners, int64_t x_batch_stride, int64_t out_batch_stride,
                         void *stream) {
    using namespace snvc;
    if (N < 0 || C < 0 || D <= 0 || H <= 0 || W <= 0 || V < 0)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_volume_resample: bad sizes");
    if (N == 0 || C == 0 || V == 0) return SNVC_OK;
    if (!x || !grid || !out) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_volume_resample: null pointer");
    if (D * H * W >= ((int64_t)1 << 31) || N > 65535 || ceil_div<int64_t>(V, 256) >= ((int64_t)1 << 31))
        return fail(SNVC_ERR_UNSUPPORTED, "snvc_volume_resample: volume, batch or grid too large");
    dim3 g((unsigned)ceil_div<int64_t>(V, 256), (unsigned)N);
    volume_resample_kernel<<<g, 256, 0, as_stream(stream)>>>(x, grid, out, (int)C, (int)D, (int)H, (int)W, V, align_corners,
                                                             x_batch_stride ? x_batch_stride : C * D * H * W,
                                                             out_batch_stride ? out_batch_stride : C * V);
    return check_launch("snvc_volume_resample");
}

int snvc_rect_to_psv_grid(const float *pts_rect, const float *P_host, float *grid, int64_t V, float u0, float u_span,
                          float v0, float v_span, float z0, float z_span, void *stream) {
    using namespace snvc;
    if (V < 0 || !P_host) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_rect_to_psv_grid: bad arguments");
    if (V == 0) return SNVC_OK;
    if (!pts_rect || !grid) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_rect_to_psv_grid: null pointer");
    const float *P = P_host;
    rect_to_psv_grid_kernel<<<(unsigned)ceil_div<int64_t>(V, 256), 256, 0, as_stream(stream)>>>(
        pts_rect, grid, V, P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], P[9], P[10], P[11], u0, u_span, v0, v_span, z0,
        z_span);
    return check_launch("snvc_rect_to_psv_grid");
}

}  // extern "C"
