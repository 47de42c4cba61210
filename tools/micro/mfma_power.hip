// Microbenchmark: sustained half-precision MFMA rate under the chip's power limit, by instruction shape and operand content.
// Every wave runs a long loop of independent MFMAs on register-resident operands (no memory traffic in the loop):
//   shape 32x32x16 (8 passes, 1024 accumulators per 16 k) against 16x16x32 (4 passes, 256 accumulators per 32 k);
//   operands: random halves, all zeros, and "B changes every MFMA / A every 4th" (the conv kernels' pattern) against both fixed.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o gpurun_out/mfma_power && gpurun_out/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NB independent accumulators; per outer step: 4 A fragments x NB B fragments as in the kernels (A reused over the B's)
template <int SHAPE, int OCC>
__global__ void __launch_bounds__(256, OCC) k(const h8 *__restrict__ src, float *out, int iters) {
    constexpr int NB = 4;
    h8 a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = src[(threadIdx.x + 256 * i) & 4095];
    for (int i = 0; i < 8; ++i) b[i] = src[(threadIdx.x * 3 + 64 * i + 1024) & 4095];
    if constexpr (SHAPE == 32) {
        f32x16 acc[NB];
        for (int p = 0; p < NB; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int p = 0; p < NB; ++p)
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[(u + 2 * p) & 7], acc[p], 0, 0, 0);
        }
        float s = 0; for (int p = 0; p < NB; ++p) for (int r = 0; r < 16; ++r) s += acc[p][r];
        if (s == 1234.5f) out[0] = s;
    } else {
        f32x4 acc[NB * 2];
        for (int p = 0; p < NB * 2; ++p) for (int r = 0; r < 4; ++r) acc[p][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int p = 0; p < NB * 2; ++p)     // twice the instructions for the same flops per outer step
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[(u + p) & 7], acc[p], 0, 0, 0);
        }
        float s = 0; for (int p = 0; p < NB * 2; ++p) for (int r = 0; r < 4; ++r) s += acc[p][r];
        if (s == 1234.5f) out[0] = s;
    }
}

template <int SHAPE, int OCC> void run(const char *name, const h8 *src) {
    float *out; hipMalloc(&out, 4);
    const int iters = 20000, grid = 256 * OCC * 4;      // OCC workgroups of 4 waves per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, OCC><<<grid, 256>>>(src, out, 2000);        // warm-up (clock ramp)
    k<SHAPE, OCC><<<grid, 256>>>(src, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SHAPE, OCC><<<grid, 256>>>(src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 /*waves*/ * iters * 16 /*MFMA-equivalents of 32768 flop per outer step*/ * 32768.0;
    printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    h8 *rnd, *zero;
    hipMalloc(&rnd, 4096 * 16); hipMalloc(&zero, 4096 * 16);
    _Float16 *h = (_Float16 *)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
    hipMemcpy(rnd, h, 4096 * 16, hipMemcpyHostToDevice);
    hipMemset(zero, 0, 4096 * 16);
    run<32, 2>("32x32x16, 2 waves/SIMD, random operands", rnd);
    run<32, 2>("32x32x16, 2 waves/SIMD, zero operands", zero);
    run<16, 2>("16x16x32, 2 waves/SIMD, random operands", rnd);
    run<16, 2>("16x16x32, 2 waves/SIMD, zero operands", zero);
    run<32, 1>("32x32x16, 1 wave/SIMD, random operands", rnd);
    run<16, 1>("16x16x32, 1 wave/SIMD, random operands", rnd);
    run<32, 2>("32x32x16, 2 waves/SIMD, random operands (again)", rnd);
    return 0;
}
