// rp_common.h -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "rectipy_hip.h"
#include "rectipy_math.h"

#define RP_WAVE 64

extern thread_local char rp_err_buf[512];

#define RP_FAIL(code, ...)                                   \
    do {                                                     \
        snprintf(rp_err_buf, sizeof rp_err_buf, __VA_ARGS__); \
        return (code);                                       \
    } while (0)

#define RP_HIP_CHECK(expr)                                                            \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            RP_FAIL(RP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                              \
    } while (0)

typedef float rp_f4 __attribute__((ext_vector_type(4)));

// 16-byte streaming load of the once-read coupling matrix (non-temporal: keep the
// per-XCD L2 for the re-read source vector).
__device__ __forceinline__ rp_f4 rp_ld4_stream(const float* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const rp_f4*>(p));
}
__device__ __forceinline__ rp_f4 rp_ld4(const float* p) {
    return *reinterpret_cast<const rp_f4*>(p);
}

// Fixed butterfly: after it every lane holds the same total.  s[l] += s[l ^ off]
// for off = 32,16,8,4,2,1 (the order is part of the summation spec).
__device__ __forceinline__ float rp_wave_allreduce(float s) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, RP_WAVE);
    return s;
}

__device__ __forceinline__ float rp_lane_sum4(rp_f4 a) { return (a.x + a.y) + (a.z + a.w); }
