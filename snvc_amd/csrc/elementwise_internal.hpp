// Launchers of elementwise.hip that other translation units build on.
#pragma once
#include "common.hpp"

namespace snvc {

// Batch statistics -> scale / shift [C] (+ mean / var [C]) from fp64 partial pairs partial[((n*C + c) * splits + k)][2] =
// (sum, sum of squares) over S elements per (n, c): the second pass of snvc_norm_stats (per_sample = 0, groups = C).
void launch_norm_finalize(const double *partial, const float *gamma, const float *beta, float *scale, float *shift, float *mean,
                          float *var, int64_t N, int64_t C, int64_t S, int splits, float eps, hipStream_t st);

// conv3d.hip: stats[((n * tiles + t) * groups + cg)][32][2] -> partial[(n * C + c)][2] in a fixed order (the statistics epilogues of the
// convolution kernels leave one (sum, sum of squares) pair per workgroup slot and 32-channel group)
// `scratch`: conv_stats_fold_scratch_doubles(N, groups, tiles) doubles (0 for up to 4096 tiles: the per-channel kernel alone); a first round
// sums 64 slots per workgroup with consecutive threads on consecutive doubles (one workgroup per channel walking 46 k slots 512 bytes
// apart took 106 us at cfg4's transposed layer)
int64_t conv_stats_fold_scratch_doubles(int64_t N, int groups, int64_t tiles);
void launch_conv_stats_fold(const double *stats, double *scratch, double *partial, int64_t N, int C, int groups, int64_t tiles, hipStream_t st);

}  // namespace snvc
