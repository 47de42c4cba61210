#include "common.hpp"

#include <cstring>

namespace snvc {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace snvc

extern "C" {
const char *snvc_last_error_string(void) { return snvc::g_err; }
int snvc_abi_version(void) { return 6; }   // 6: *_twin passes, snvc_split_scale_bound, fp32 residual on the split kernels' fp32 output (r6); 5: *_amax entry points (r6); 2: snvc_conv3d_desc.ksize_d, fp16-storage mode, volume resampling; 3: side head, one-channel transposed layers; 4: tail projection
}
