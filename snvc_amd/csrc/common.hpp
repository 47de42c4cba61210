// Shared host-side helpers for libsnvc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "snvc_hip.h"

namespace snvc {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

inline int fail(int code, const char *msg) {
    set_error("%s", msg);
    return code;
}

// Set by allow_large_lds when the attribute call failed (the launch that follows is skipped by its caller and
// check_launch reports the failure, message already recorded by set_error).
inline thread_local bool g_launch_aborted = false;

// Called after every launch: hipGetLastError is a host-side query (no device sync).
inline int check_launch(const char *what) {
    if (g_launch_aborted) {
        g_launch_aborted = false;
        (void)hipGetLastError();
        return SNVC_ERR_HIP;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SNVC_ERR_HIP;
    }
    return SNVC_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Kernels that need more than 48 KB of dynamic LDS must say so once per device (one process may drive several,
// from several host threads: nn.DataParallel replicas).  `done_mask` has one bit per device; the bit is set only
// after the attribute call succeeded, so a racing thread at worst repeats the (idempotent) call.
inline bool allow_large_lds(const void *func, int bytes, std::atomic<unsigned> &done_mask) {
    if (bytes <= 48 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    if (done_mask.load(std::memory_order_acquire) & (1u << dev)) return true;
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d): %s", bytes, hipGetErrorString(e));
        g_launch_aborted = true;
        return false;
    }
    done_mask.fetch_or(1u << dev, std::memory_order_release);
    return true;
}

// Compute units of the current device, rounded down to whole XCD octets (persistent-grid sizing).
inline int device_cu_count() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cached[dev] = n / 8 * 8;
    }
    return cached[dev];
}

template <typename T>
__host__ __device__ inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

constexpr int kWave = 64;  // gfx950 wavefront

}  // namespace snvc
