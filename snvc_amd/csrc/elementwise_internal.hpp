// Launchers of elementwise.hip that other translation units build on.
#pragma once
#include "common.hpp"

namespace snvc {

// Batch statistics -> scale / shift [C] (+ mean / var [C]) from fp64 partial pairs partial[((n*C + c) * splits + k)][2] =
// (sum, sum of squares) over S elements per (n, c): the second pass of snvc_norm_stats (per_sample = 0, groups = C).
void launch_norm_finalize(const double *partial, const float *gamma, const float *beta, float *scale, float *shift, float *mean,
                          float *var, int64_t N, int64_t C, int64_t S, int splits, float eps, hipStream_t st);

}  // namespace snvc
