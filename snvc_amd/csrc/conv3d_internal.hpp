// Launchers shared between the conv3d translation units (conv3d.hip dispatches, the small-layer
// kernels live in conv3d_small.hip so that they build in seconds).
#pragma once
#include "common.hpp"

namespace snvc {

// ConvTranspose3d(Cin, 1, k3, s2, p1, op1) + epilogue on the VALU (conv3d_small.hip).  `w` is the layer's raw
// [Cin][1][3][3][3] weight.  Returns false when the shape does not qualify (the caller then takes the MFMA kernel).
bool deconv3d_cout1_qualifies(const snvc_conv3d_desc &d, const float *x, const float *y, const float *res,
                              int64_t x_bs, int64_t y_bs, int64_t r_bs);
void deconv3d_cout1_launch(const snvc_conv3d_desc &d, const float *x, const float *w, const float *scale,
                           const float *bias, const float *res, float *y, int64_t x_bs, int64_t y_bs, int64_t r_bs,
                           hipStream_t st);

}  // namespace snvc
