// Microbenchmark: fp32 MFMAs whose A and B operands come from LDS (the direct convolution kernels' inner loop):
// per MFMA one ds_read_b32 for A (+ one for B in mode 2), no VALU.  3 workgroups of 4 waves per CU.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_lds_feed.hip -o /tmp/m3 && /tmp/m3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>   // 0: operands in registers; 1: A from LDS; 2: A and B from LDS; 3: A b32 + B via b128 (4 MFMAs per read)
__global__ void __launch_bounds__(256, 3) k(float *out, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 0.001f;
    __syncthreads();
    f32x16 acc[6];
    for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    const float *pa = lds + (threadIdx.x & 63), *pb = lds + 4096 + (threadIdx.x & 63) * (MODE == 3 ? 4 : 1);
    float a = threadIdx.x, b = 2.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            if (MODE == 3) {
                const float4 bv = *reinterpret_cast<const float4 *>(pb + (u & 3) * 256);
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    const float av = pa[(u * 6 + p) * 64];
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, p & 1 ? bv.x : bv.y, acc[p], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    const float av = MODE >= 1 ? pa[(u * 6 + p) * 64] : a;
                    const float bv = MODE >= 2 ? pb[((u * 6 + p) & 31) * 64] : b;
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[p], 0, 0, 0);
                }
            }
        }
    }
    float s = 0; for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) s += acc[p][r];
    if (s == 1234.5f) out[0] = s;
}
template <int MODE> void run(const char *name) {
    float *out; hipMalloc(&out, 4);
    const int iters = 2000, grid = 256 * 3 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256, 32768>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256, 32768>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)grid * 4 * iters * 54 * 4096.0;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}
int main() {
    run<0>("operands in registers");
    run<1>("A from LDS (1 ds_read_b32 per MFMA)");
    run<2>("A and B from LDS (2 ds_read_b32 per MFMA)");
    run<3>("A b32 per MFMA + B b128 per 6 MFMAs");
    return 0;
}
