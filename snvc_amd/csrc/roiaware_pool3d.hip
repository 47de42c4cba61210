// RoI-aware point pooling for gfx950.
//
// Reference: snvc/extension/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-359 (kernels),
//            snvc/extension/roiaware_pool3d/src/roiaware_pool3d.cpp:29-168 (launchers, CPU variant).
// Same results (bit-exact voxel codes, point lists and argmax), different machine mapping:
//   * the reference builds the per-voxel point lists with ONE THREAD per box scanning all P
//     points serially (roiaware_pool3d_kernel.cu:78-108).  Here one 64-lane wavefront owns a
//     box, tests 64 points per step, and appends the survivors with a ballot-based stable
//     ranking (points keep their ascending order inside every voxel, which is what makes
//     truncation at 127 points and argmax ties deterministic);
//   * no per-call device malloc (the reference cudaMalloc's / frees the mask, :203-205,228):
//     the caller passes the B*P int32 workspace;
//   * pooling threads are channel-fastest, so point-feature reads and pooled writes coalesce.
// The rotation uses correctly rounded float cos/sin (computed in double), see the note in
// oracle/roiaware_pool3d_ref.c.
#include "common.hpp"

namespace snvc {
namespace {

#pragma clang fp contract(off)

__host__ __device__ inline int point_in_box(const float *pt, const float *box, float margin, float &lx,
                                            float &ly) {
    const float x = pt[0], y = pt[1], z = pt[2];
    const float cx = box[0], cy = box[1], cz = box[2];
    const float dx = box[3], dy = box[4], dz = box[5], rz = box[6];
    if ((double)fabsf(z - cz) > (double)dz / 2.0) return 0;
    const float cosa = (float)cos((double)(-rz)), sina = (float)sin((double)(-rz));
    const float sx = x - cx, sy = y - cy;
    lx = sx * cosa + sy * (-sina);
    ly = sx * sina + sy * cosa;
    return (int)((double)fabsf(lx) < (double)dx / 2.0 + (double)margin) &
           (int)((double)fabsf(ly) < (double)dy / 2.0 + (double)margin);
}

__device__ __forceinline__ unsigned clamp_index(float v, int out) {
    const unsigned idx = (unsigned)(int)v, hi = (unsigned)(out - 1);
    return idx < hi ? idx : hi;
}

// grid (ceil(P/256), B)
__global__ void __launch_bounds__(256)
roiaware_mask_kernel(const float *__restrict__ rois, const float *__restrict__ pts, int32_t *__restrict__ mask,
                     int P, int ox, int oy, int oz) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (p >= P) return;
    const float *box = rois + 7 * b, *pt = pts + 3 * (int64_t)p;
    float lx = 0, ly = 0;
    int32_t code = -1;
    if (point_in_box(pt, box, 1e-5f, lx, ly)) {
        const float lz = pt[2] - box[2];
        const float dx = box[3], dy = box[4], dz = box[5];
        const float xr = dx / ox, yr = dy / oy, zr = dz / oz;
        const unsigned xi = clamp_index((lx + dx / 2) / xr, ox);
        const unsigned yi = clamp_index((ly + dy / 2) / yr, oy);
        const unsigned zi = clamp_index((lz + dz / 2) / zr, oz);
        code = (int32_t)((xi << 16) + (yi << 8) + zi);
    }
    mask[(int64_t)b * P + p] = code;
}

// One wavefront per box.  Every step looks at 64 consecutive points; lanes holding an inside
// point are grouped by voxel code with a ballot loop, ranked inside their group by lane order
// (= ascending point index) and appended behind the voxel's running count.
__global__ void __launch_bounds__(64)
roiaware_collect_kernel(const int32_t *__restrict__ mask, int32_t *__restrict__ lists, int P, int max_pts,
                        int ox, int oy, int oz) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int cap = max_pts - 1;
    int32_t *box_lists = lists + (int64_t)b * ox * oy * oz * max_pts;
    const int32_t *m = mask + (int64_t)b * P;
    for (int base = 0; base < P; base += 64) {  // trip count is wave-uniform
        const int k = base + lane;
        const int32_t code = k < P ? m[k] : -1;
        unsigned long long pending = __ballot(code != -1);
        while (pending) {  // wave-uniform loop: one voxel code per iteration
            const int leader = __ffsll((long long)pending) - 1;
            const int32_t lead_code = __shfl(code, leader, 64);
            const unsigned long long grp = __ballot(code == lead_code) & pending;
            // the group's leader reserves the slots with one L2 atomic (coherent by construction;
            // counts may overshoot the cap here and are clamped by roiaware_clamp_kernel)
            int cnt = 0;
            if (lane == leader) {
                const unsigned uc = (unsigned)code;
                const unsigned xi = (uc >> 16) & 0xFF, yi = (uc >> 8) & 0xFF, zi = uc & 0xFF;
                int32_t *cell = box_lists + ((int64_t)(xi * oy + yi) * oz + zi) * max_pts;
                cnt = atomicAdd(cell, __popcll(grp));
            }
            cnt = __shfl(cnt, leader, 64);
            if ((grp >> lane) & 1ull) {
                const unsigned uc = (unsigned)code;
                const unsigned xi = (uc >> 16) & 0xFF, yi = (uc >> 8) & 0xFF, zi = uc & 0xFF;
                int32_t *cell = box_lists + ((int64_t)(xi * oy + yi) * oz + zi) * max_pts;
                const int rank = __popcll(grp & ((1ull << lane) - 1ull));
                const int slot = cnt + rank;
                if (slot < cap) cell[slot + 1] = k;
            }
            pending &= ~grp;
        }
    }
}

// slot 0 saturates at max_pts - 1 in the reference (roiaware_pool3d_kernel.cu:97-100)
__global__ void __launch_bounds__(256)
roiaware_clamp_kernel(int32_t *__restrict__ lists, int max_pts, int64_t cells) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const int cap = max_pts - 1;
    if (lists[i * max_pts] > cap) lists[i * max_pts] = cap;
}

// one thread per (box, voxel, channel), channel fastest
__global__ void __launch_bounds__(256)
roiaware_pool_kernel(const float *__restrict__ feat, const int32_t *__restrict__ lists,
                     float *__restrict__ pooled, int32_t *__restrict__ argmax, int C, int max_pts,
                     int64_t total, int pool_method) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int64_t cellidx = i / C;
    const int32_t *cell = lists + cellidx * max_pts;
    const int n = cell[0];
    if (pool_method == 0) {
        int32_t best = -1;
        float best_val = -INFINITY;
        for (int k = 1; k <= n; ++k) {
            const float f = feat[(int64_t)cell[k] * C + c];
            if (f > best_val) { best_val = f; best = cell[k]; }
        }
        if (best != -1) pooled[i] = best_val;
        argmax[i] = best;
    } else {
        float sum = 0;
        for (int k = 1; k <= n; ++k) sum += feat[(int64_t)cell[k] * C + c];
        if (n > 0) pooled[i] = sum / n;
    }
}

__global__ void __launch_bounds__(256)
roiaware_pool_bwd_kernel(const int32_t *__restrict__ lists, const int32_t *__restrict__ argmax,
                         const float *__restrict__ grad_out, float *__restrict__ grad_in, int C, int max_pts,
                         int64_t total, int pool_method) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    if (pool_method == 0) {
        const int32_t a = argmax[i];
        if (a == -1) return;
        atomicAdd(grad_in + (int64_t)a * C + c, grad_out[i] * 1);
    } else {
        const int32_t *cell = lists + (i / C) * max_pts;
        const int n = cell[0];
        const float share = 1 / fmaxf((float)n, 1.0f);
        for (int k = 1; k <= n; ++k) atomicAdd(grad_in + (int64_t)cell[k] * C + c, grad_out[i] * share);
    }
}

// grid (ceil(M/256), Bs)
__global__ void __launch_bounds__(256)
points_in_boxes_kernel(const float *__restrict__ boxes, const float *__restrict__ pts, int32_t *__restrict__ out,
                       int T, int M) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (m >= M) return;
    const float *pt = pts + ((int64_t)s * M + m) * 3;
    float lx = 0, ly = 0;
    for (int k = 0; k < T; ++k)
        if (point_in_box(pt, boxes + ((int64_t)s * T + k) * 7, 1e-5f, lx, ly)) {
            out[(int64_t)s * M + m] = k;
            break;
        }
}

}  // namespace
}  // namespace snvc

extern "C" {

int snvc_roiaware_pool3d_forward(const float *rois, const float *pts, const float *feat, int32_t *argmax,
                                 int32_t *pts_idx_of_voxels, float *pooled, int32_t *workspace, int B, int P,
                                 int C, int max_pts, int ox, int oy, int oz, int pool_method, void *stream) {
    using namespace snvc;
    if (B < 0 || P < 0 || C < 0 || max_pts < 1 || ox < 1 || oy < 1 || oz < 1)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_roiaware_pool3d_forward: bad sizes");
    if (!(ox < 256 && oy < 256 && oz < 256))  // roiaware_pool3d.cpp:53, indices are packed in 8 bits
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_roiaware_pool3d_forward: out sizes must be < 256");
    if (pool_method != 0 && pool_method != 1)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_roiaware_pool3d_forward: pool_method must be 0 (max) or 1 (avg)");
    if (B == 0) return SNVC_OK;
    if (B > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_roiaware_pool3d_forward: more than 65535 boxes");
    if (!rois || !pts_idx_of_voxels || !pooled || (pool_method == 0 && !argmax) || (P > 0 && (!pts || !feat || !workspace)))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_roiaware_pool3d_forward: null pointer");
    hipStream_t st = as_stream(stream);
    int rc;
    if (P > 0) {
        roiaware_mask_kernel<<<dim3((unsigned)ceil_div(P, 256), (unsigned)B), 256, 0, st>>>(rois, pts, workspace, P, ox, oy, oz);
        if ((rc = check_launch("snvc_roiaware_pool3d_forward(mask)"))) return rc;
        roiaware_collect_kernel<<<dim3((unsigned)B), 64, 0, st>>>(workspace, pts_idx_of_voxels, P, max_pts, ox, oy, oz);
        if ((rc = check_launch("snvc_roiaware_pool3d_forward(collect)"))) return rc;
        const int64_t cells = (int64_t)B * ox * oy * oz;
        roiaware_clamp_kernel<<<dim3((unsigned)ceil_div<int64_t>(cells, 256)), 256, 0, st>>>(pts_idx_of_voxels, max_pts, cells);
        if ((rc = check_launch("snvc_roiaware_pool3d_forward(clamp)"))) return rc;
    }
    const int64_t total = (int64_t)B * ox * oy * oz * C;
    if (total > 0) {
        roiaware_pool_kernel<<<dim3((unsigned)ceil_div<int64_t>(total, 256)), 256, 0, st>>>(
            feat, pts_idx_of_voxels, pooled, argmax, C, max_pts, total, pool_method);
        if ((rc = check_launch("snvc_roiaware_pool3d_forward(pool)"))) return rc;
    }
    return SNVC_OK;
}

int snvc_roiaware_pool3d_backward(const int32_t *pts_idx_of_voxels, const int32_t *argmax, const float *grad_out,
                                  float *grad_in, int B, int C, int max_pts, int ox, int oy, int oz, int pool_method,
                                  void *stream) {
    using namespace snvc;
    if (B < 0 || C < 0 || max_pts < 1 || ox < 1 || oy < 1 || oz < 1)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_roiaware_pool3d_backward: bad sizes");
    if (pool_method != 0 && pool_method != 1)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_roiaware_pool3d_backward: pool_method must be 0 (max) or 1 (avg)");
    const int64_t total = (int64_t)B * ox * oy * oz * C;
    if (total == 0) return SNVC_OK;
    if (!pts_idx_of_voxels || !grad_out || !grad_in || (pool_method == 0 && !argmax))
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_roiaware_pool3d_backward: null pointer");
    roiaware_pool_bwd_kernel<<<dim3((unsigned)ceil_div<int64_t>(total, 256)), 256, 0, as_stream(stream)>>>(
        pts_idx_of_voxels, argmax, grad_out, grad_in, C, max_pts, total, pool_method);
    return check_launch("snvc_roiaware_pool3d_backward");
}

int snvc_points_in_boxes_gpu(const float *boxes, const float *pts, int32_t *out, int Bs, int T, int M, void *stream) {
    using namespace snvc;
    if (Bs < 0 || T < 0 || M < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_points_in_boxes_gpu: negative size");
    if (Bs == 0 || M == 0 || T == 0) return SNVC_OK;
    if (Bs > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_points_in_boxes_gpu: batch > 65535");
    if (!boxes || !pts || !out) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_points_in_boxes_gpu: null pointer");
    points_in_boxes_kernel<<<dim3((unsigned)ceil_div(M, 256), (unsigned)Bs), 256, 0, as_stream(stream)>>>(boxes, pts, out, T, M);
    return check_launch("snvc_points_in_boxes_gpu");
}

// The reference's points_in_boxes_cpu is a host function operating on CPU tensors
// (roiaware_pool3d.cpp:137-168); it stays a host function here, margin 1e-2 (:131).
int snvc_points_in_boxes_cpu(const float *boxes_host, const float *pts_host, int32_t *out_host, int T, int M) {
    using namespace snvc;
    if (T < 0 || M < 0) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_points_in_boxes_cpu: negative size");
    if (T * (int64_t)M == 0) return SNVC_OK;
    if (!boxes_host || !pts_host || !out_host) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_points_in_boxes_cpu: null pointer");
    float lx = 0, ly = 0;
    for (int i = 0; i < T; ++i)
        for (int j = 0; j < M; ++j)
            out_host[(int64_t)i * M + j] = point_in_box(pts_host + 3 * (int64_t)j, boxes_host + 7 * (int64_t)i, 1e-2f, lx, ly);
    return SNVC_OK;
}

}  // extern "C"
