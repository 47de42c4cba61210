// Projection of the Vernier 3D sampling grid into the left / right RoI crops, on the device
// (SURVEY.md section 8a row a11, 8f N2).
//
// Reference (host, numpy float64, per instance 786 k points x 2 cameras, then a 6.3 MB/side
// host->device copy): refinementDataset._init_3d_grid / _to_cam / _generate_grid_proj
// (snvc/dataset/KITTIRefinement_dataset.py:267-282,828-868), Calibration.project_rect_to_image
// (snvc/dataset/kitti_util.py:282-293), affine_transform (snvc/utils/img_proc.py:71-74).
//
// Here one thread owns one grid point of one instance: linspace grid point -> rotate about y by
// ry + pi/2 and translate to the proposal centre -> P2 / P3 projection and perspective divide ->
// 2x3 crop affine -> float32 (x row, y row), exactly the [N,2,V] tensors the gather kernel eats.
// All arithmetic is fp64 in the reference's operation order (products summed k-ascending the way a
// BLAS micro-kernel does, with FMA), so the float32 results agree bit for bit except where the
// fp64 value sits within rounding noise of a float32 tie.
#include "common.hpp"

namespace snvc {
namespace {

struct GridSpec {
    double x0, xs, x1, y0, ys, y1, z0, zs, z1;   // start, step, stop per axis (numpy.linspace)
    int nh, nw, nl;
};

__device__ __forceinline__ double lin(int i, int n, double start, double step, double stop) {
#pragma clang fp contract(off)
    if (n > 1 && i == n - 1) return stop;     // numpy pins the end point
    return (double)i * step + start;
}

__global__ void __launch_bounds__(256)
grid_projection_kernel(const double *__restrict__ samples, const double *__restrict__ P_left,
                       const double *__restrict__ P_right, const double *__restrict__ trans_l,
                       const double *__restrict__ trans_r, float *__restrict__ out_l, float *__restrict__ out_r,
                       double *__restrict__ grid_cam, GridSpec g, int64_t V) {
    const int64_t n = blockIdx.y;
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int il = (int)(v % g.nl), iw = (int)((v / g.nl) % g.nw), ih = (int)(v / ((int64_t)g.nl * g.nw));
    const double gx = lin(iw, g.nw, g.x0, g.xs, g.x1);
    const double gy = lin(ih, g.nh, g.y0, g.ys, g.y1);
    const double gz = lin(il, g.nl, g.z0, g.zs, g.z1);
    const double *s = samples + n * 7;
    double ry, cx, cy, cz;
    {
#pragma clang fp contract(off)
        ry = s[6] + 0.5 * 3.141592653589793;
        cx = s[3];
        cy = s[4] - s[0] * 0.5;
        cz = s[5];
    }
    const double c = cos(ry), sn = sin(ry);
    // rot @ pts (k-ascending FMA chain), then + translation (separately rounded add)
    double X = fma(sn, gz, fma(0.0, gy, c * gx));
    double Y = fma(0.0, gz, fma(1.0, gy, 0.0 * gx));
    double Z = fma(c, gz, fma(0.0, gy, (-sn) * gx));
    {
#pragma clang fp contract(off)
        X = X + cx; Y = Y + cy; Z = Z + cz;
    }
    if (grid_cam) {
        double *gc = grid_cam + (n * V + v) * 3;
        gc[0] = X; gc[1] = Y; gc[2] = Z;
    }
#pragma unroll
    for (int cam = 0; cam < 2; ++cam) {
        const double *P = cam == 0 ? P_left : P_right;
        const double *T = (cam == 0 ? trans_l : trans_r) + n * 6;
        const double pu = fma(1.0, P[3], fma(Z, P[2], fma(Y, P[1], X * P[0])));
        const double pv = fma(1.0, P[7], fma(Z, P[6], fma(Y, P[5], X * P[4])));
        const double pw = fma(1.0, P[11], fma(Z, P[10], fma(Y, P[9], X * P[8])));
        const double u = pu / pw, w = pv / pw;
        const double ox = fma(T[2], 1.0, fma(T[1], w, T[0] * u));
        const double oy = fma(T[5], 1.0, fma(T[4], w, T[3] * u));
        float *o = (cam == 0 ? out_l : out_r) + n * 2 * V;
        o[v] = (float)ox;
        o[V + v] = (float)oy;
    }
}

}  // namespace
}  // namespace snvc

extern "C" {

int snvc_grid_projection(const double *samples, const double *P_left, const double *P_right, const double *trans_l,
                         const double *trans_r, const double *ranges_host, int nh, int nw, int nl, float *out_l,
                         float *out_r, double *grid_cam, int N, void *stream) {
    using namespace snvc;
    if (N < 0 || nh < 1 || nw < 1 || nl < 1) return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_grid_projection: bad sizes");
    if (N == 0) return SNVC_OK;
    if (N > 65535) return fail(SNVC_ERR_UNSUPPORTED, "snvc_grid_projection: more than 65535 instances");
    if (!samples || !P_left || !P_right || !trans_l || !trans_r || !ranges_host || !out_l || !out_r)
        return fail(SNVC_ERR_INVALID_ARGUMENT, "snvc_grid_projection: null pointer");
    GridSpec g;
    g.nh = nh; g.nw = nw; g.nl = nl;
    // numpy.linspace: step = (stop - start) / (num - 1)
    g.x0 = ranges_host[0]; g.x1 = ranges_host[1]; g.xs = nw > 1 ? (g.x1 - g.x0) / (double)(nw - 1) : 0.0;
    g.y0 = ranges_host[2]; g.y1 = ranges_host[3]; g.ys = nh > 1 ? (g.y1 - g.y0) / (double)(nh - 1) : 0.0;
    g.z0 = ranges_host[4]; g.z1 = ranges_host[5]; g.zs = nl > 1 ? (g.z1 - g.z0) / (double)(nl - 1) : 0.0;
    const int64_t V = (int64_t)nh * nw * nl;
    dim3 grid((unsigned)ceil_div<int64_t>(V, 256), (unsigned)N);
    grid_projection_kernel<<<grid, 256, 0, as_stream(stream)>>>(samples, P_left, P_right, trans_l, trans_r, out_l, out_r,
                                                                grid_cam, g, V);
    return check_launch("snvc_grid_projection");
}

}  // extern "C"
