// Microbenchmark: do global loads issued by OTHER waves of the same CU slow fp32 MFMA waves down?
// 12 MFMA waves (3 per SIMD) + 4 load waves (1 per SIMD) per workgroup, one workgroup per CU.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_vs_loads.hip -o /tmp/m2 && /tmp/m2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(1024, 1) k(const f32x4 *src, size_t n4, float *out, int mfma_iters, int load_iters, int mode) {
    const int wave = threadIdx.x >> 6;
    if (wave < 12) {
        if (!(mode & 1)) return;
        f32x16 acc[3];
        for (int p = 0; p < 3; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
        float a = threadIdx.x, b = 2.f;
        for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int p = 0; p < 3; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0);
        float s = 0; for (int p = 0; p < 3; ++p) for (int r = 0; r < 16; ++r) s += acc[p][r];
        if (s == 1234.5f) out[0] = s;
    } else {
        if (!(mode & 2)) return;
        const size_t t = (size_t)blockIdx.x * 256 + (threadIdx.x - 768), stride = (size_t)gridDim.x * 256;
        f32x4 s = {0, 0, 0, 0};
        size_t i = t;
        for (int it = 0; it < load_iters; ++it) {
            f32x4 v[13];
#pragma unroll
            for (int j = 0; j < 13; ++j) { v[j] = src[i % n4]; i += stride; }
#pragma unroll
            for (int j = 0; j < 13; ++j) s += v[j];
        }
        if (s[0] + s[1] + s[2] + s[3] == 1234.5f) out[1] = s[0];
    }
}
int main() {
    const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    f32x4 *src; float *out;
    hipMalloc(&src, bytes); hipMemset(src, 0, bytes); hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256, mi = 3000, li = 300;
    for (int mode = 1; mode <= 3; ++mode) {
        k<<<grid, 1024>>>(src, n4, out, 10, 10, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<grid, 1024>>>(src, n4, out, mi, li, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)grid * 12 * mi * 24 * 4096.0, by = (double)grid * 256 * li * 13 * 16;
        printf("mode %d (%s): %8.3f ms", mode, mode == 1 ? "MFMA waves only" : mode == 2 ? "load waves only" : "both", ms);
        if (mode & 1) printf("   %7.1f TFLOP/s", fl / ms / 1e9);
        if (mode & 2) printf("   %7.1f GB/s loaded", by / ms / 1e6);
        printf("\n");
    }
    return 0;
}
