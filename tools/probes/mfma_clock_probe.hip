// Development probe: what does a pure v_mfma_f32_32x32x2_f32 stream sustain on this part, and at what
// shader clock?  clock64() counts shader cycles (s_memtime), wall_clock64() a constant 100 MHz.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_clock_probe mfma_clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256, 2) probe(float *out, long long *clk, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main(int argc, char **argv) {
    const int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 2;
    const int nblk = 256 * blocks_per_cu;
    float *out; long long *clk;
    hipMalloc(&out, nblk * 256 * 4); hipMalloc(&clk, nblk * 16);
    long long *h = (long long *)malloc(nblk * 16);
    for (int iters : {100, 1000, 10000, 40000, 40000, 40000}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        probe<8><<<nblk, 256>>>(out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, clk, nblk * 16, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int i = 0; i < nblk; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
        cyc /= nblk; wall /= nblk;
        const double mfma_per_simd = (double)iters * 8 * blocks_per_cu;   // one wave of each block per SIMD
        const double tflops = (double)nblk * 4 * iters * 8 * 4096.0 / (ms * 1e-3) / 1e12;
        printf("blocks/CU %d iters %6d: %.3f ms  %.1f TFLOP/s  shader clock %.3f GHz  cycles/MFMA/SIMD %.1f\n",
               blocks_per_cu, iters, ms, tflops, cyc / (wall * 10.0) , cyc / mfma_per_simd);
    }
    return 0;
}
