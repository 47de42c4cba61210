// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate with 6 independent accumulators per wave, 3 waves per SIMD,
// and V extra VALU operations per group of 6 MFMAs (the shape of the Winograd kernels' inner step).
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_issue.hip -o gpurun_out/mfma_issue && gpurun_out/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// same, with V packed (two-float) VALU operations per group
template <int V>
__global__ void __launch_bounds__(256, 3) kpk(float *out, int iters, float a0, float b0) {
    f32x16 acc[6];
    for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float a = a0 + threadIdx.x;
    f32x2 t[4];
    for (int i = 0; i < 4; ++i) { t[i][0] = b0 + i; t[i][1] = b0 - i; }
    const f32x2 c = {1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int v = 0; v < V; ++v) t[v & 3] = __builtin_elementwise_fma(t[v & 3], c, t[(v + 1) & 3]);
#pragma unroll
            for (int p = 0; p < 6; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t[p & 3][p >> 2], acc[p], 0, 0, 0);
        }
    }
    float s = 0; for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) s += acc[p][r];
    if (s == 1234.5f) out[0] = s;
}
template <int V, int L>
__global__ void __launch_bounds__(256, 3) k(float *out, int iters, float a0, float b0) {
    extern __shared__ float lds[];
    f32x16 acc[6];
    for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0, t[8];
    for (int i = 0; i < 8; ++i) t[i] = b0 + i;
    if (L) { for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i; __syncthreads(); }
    const float *lp = lds + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (L) {
#pragma unroll
                for (int j = 0; j < L; ++j) { float4 q = *reinterpret_cast<const float4 *>(lp + ((u * L + j) & 7) * 256); t[j & 7] += q.x + q.w; }
            }
#pragma unroll
            for (int v = 0; v < V; ++v) t[v & 7] = __builtin_fmaf(t[v & 7], 1.0001f, t[(v + 1) & 7]);
#pragma unroll
            for (int p = 0; p < 6; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t[p], acc[p], 0, 0, 0);
        }
    }
    float s = 0; for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) s += acc[p][r];
    if (s == 1234.5f) out[0] = s;
}
template <int V, int L> void run(const char *name) {
    float *out; hipMalloc(&out, 4);
    const int iters = 4000, grid = 256 * 3 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<V, L><<<grid, 256, 16384>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<V, L><<<grid, 256, 16384>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)grid * 4 /*waves*/ * iters * 4 * 6 * 4096.0;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}
template <int V> void runpk(const char *name) {
    float *out; hipMalloc(&out, 4);
    const int iters = 4000, grid = 256 * 3 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kpk<V><<<grid, 256, 0>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kpk<V><<<grid, 256, 0>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)grid * 4 * iters * 4 * 6 * 4096.0;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}
int main() {
    runpk<4>("6 MFMA + 4 pk VALU");
    runpk<8>("6 MFMA + 8 pk VALU");
    runpk<12>("6 MFMA + 12 pk VALU");
    run<0, 0>("6 MFMA");
    run<8, 0>("6 MFMA + 8 VALU");
    run<16, 0>("6 MFMA + 16 VALU");
    run<24, 0>("6 MFMA + 24 VALU");
    run<32, 0>("6 MFMA + 32 VALU");
    run<16, 3>("6 MFMA + 16 VALU + 3 b128");
    run<16, 6>("6 MFMA + 16 VALU + 6 b128");
    return 0;
}
