// Microbenchmark: does the ORDER of a quarter-step's 12 MFMAs matter under the power limit?  (conv3d_x3q_kernel: per tile row 4 A
// fragments -- weights hi / lo of the two channel halves -- and 4 B fragments -- the row's two halves x (hi | lo) planes; 12 of the
// 16 pairs are used.)  "kernel order" = term, then row half, then channel half (both operands change almost every instruction);
// "gray order" changes exactly one operand from one MFMA to the next.  Register-resident random operands, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_order.hip -o gpurun_out/mfma_order && gpurun_out/mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ORDER>
__global__ void __launch_bounds__(256, 2) k(const h8 *__restrict__ src, float *out, int iters) {
    h8 a[4], b[4];      // a: [2 h + (0: hi | 1: lo)], b: [2 ph + (0: hi | 1: lo)]
    for (int i = 0; i < 4; ++i) a[i] = src[(threadIdx.x + 256 * i) & 4095];
    for (int i = 0; i < 4; ++i) b[i] = src[(threadIdx.x * 3 + 64 * i + 1024) & 4095];
    f32x4 acc[4][2][2];
    for (int r = 0; r < 4; ++r) for (int p = 0; p < 2; ++p) for (int h = 0; h < 2; ++h) acc[r][p][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#define M(r, ph, h, A, B) acc[r][ph][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[A], b[B], acc[r][ph][h], 0, 0, 0)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (ORDER == 0) {      // term 0: w_lo x_hi; term 1: w_hi x_lo; term 2: w_hi x_hi -- (ph, h) inside
                M(r, 0, 0, 1, 0); M(r, 0, 1, 3, 0); M(r, 1, 0, 1, 2); M(r, 1, 1, 3, 2);
                M(r, 0, 0, 0, 1); M(r, 0, 1, 2, 1); M(r, 1, 0, 0, 3); M(r, 1, 1, 2, 3);
                M(r, 0, 0, 0, 0); M(r, 0, 1, 2, 0); M(r, 1, 0, 0, 2); M(r, 1, 1, 2, 2);
            } else {                          // one operand changes per step
                M(r, 0, 0, 1, 0); M(r, 1, 0, 1, 2); M(r, 1, 1, 3, 2); M(r, 0, 1, 3, 0);
                M(r, 0, 1, 2, 0); M(r, 1, 1, 2, 2); M(r, 1, 0, 0, 2); M(r, 0, 0, 0, 0);
                M(r, 0, 0, 0, 1); M(r, 1, 0, 0, 3); M(r, 1, 1, 2, 3); M(r, 0, 1, 2, 1);
            }
        }
        // rotate the operands a little so that nothing is loop-invariant for the hardware's data either
        const h8 t = b[0]; b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = t;
    }
    float s = 0;
    for (int r = 0; r < 4; ++r) for (int p = 0; p < 2; ++p) for (int h = 0; h < 2; ++h) for (int e = 0; e < 4; ++e) s += acc[r][p][h][e];
    if (s == 1234.5f) out[0] = s;
}

template <int ORDER> double run(const char *name, const h8 *src) {
    float *out; hipMalloc(&out, 4);
    const int iters = 20000, grid = 256 * 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<ORDER><<<grid, 256>>>(src, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<ORDER><<<grid, 256>>>(src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 48 * 16384.0;
    printf("%-28s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
    hipFree(out);
    return ms;
}

int main() {
    h8 *rnd;
    hipMalloc(&rnd, 4096 * 16);
    _Float16 *h = (_Float16 *)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
    hipMemcpy(rnd, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int r = 0; r < 3; ++r) {
        run<0>("kernel order, random", rnd);
        run<1>("gray order, random", rnd);
    }
    return 0;
}
