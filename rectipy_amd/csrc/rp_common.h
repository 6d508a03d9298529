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

// ---- per-device host-side caches: one process may drive several GPUs (engine.py: HipBackend._on) ----------------
#define RP_MAX_DEVICES 64

// number of CUs of the CURRENT device (0 on error)
inline int rp_cu_count() {
    static int cu[RP_MAX_DEVICES];  // benign race: every writer stores the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RP_MAX_DEVICES) return 0;
    if (cu[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cu[dev] = prop.multiProcessorCount;
    }
    return cu[dev];
}

// LDS bytes one workgroup may allocate on the CURRENT device (0 on error); 160 KiB on gfx950
inline size_t rp_lds_limit() {
    static size_t lim[RP_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RP_MAX_DEVICES) return 0;
    if (lim[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return 0;
        lim[dev] = (size_t)v;
    }
    return lim[dev];
}

// raise the dynamic-LDS limit of `Kernel` on the CURRENT device (once per kernel and device)
template <auto Kernel>
inline int rp_allow_dynamic_lds(size_t bytes) {
    static unsigned char done[RP_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RP_MAX_DEVICES) return RP_ERR_HIP;
    if (!done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bytes) != hipSuccess)
            return RP_ERR_HIP;
        done[dev] = 1;
    }
    return RP_OK;
}

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
// The value of lane l ^ OFF, without the LDS crossbar (ds_bpermute) wherever gfx950 has a register form: the half / quarter
// swaps v_permlane32_swap / v_permlane16_swap (OFF = 32, 16: rp_pair_sum below), DPP row_ror:8, two bank-masked DPP
// row shifts (OFF = 4), DPP quad permutations (OFF = 2, 1).  Lane patterns measured on the chip: scripts/probes/lane_ops_probe.hip.
typedef unsigned rp_u2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ float rp_lane_xor(float v) {
    static_assert(OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "register forms of lane ^ OFF inside a row of 16");
    const unsigned u = __float_as_uint(v);
    if constexpr (OFF == 8) return __uint_as_float(__builtin_amdgcn_update_dpp(0u, u, 0x128, 0xf, 0xf, false));       // row_ror:8
    else if constexpr (OFF == 2) return __uint_as_float(__builtin_amdgcn_update_dpp(0u, u, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    else if constexpr (OFF == 1) return __uint_as_float(__builtin_amdgcn_update_dpp(0u, u, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    else {
        unsigned r = __builtin_amdgcn_update_dpp(0u, u, 0x104, 0xf, 0x5, false);   // row_shl:4 into banks 0, 2: lane + 4
        r = __builtin_amdgcn_update_dpp(r, u, 0x114, 0xf, 0xa, false);             // row_shr:4 into banks 1, 3: lane - 4
        return __uint_as_float(r);
    }
}
// a[l] + a[l ^ OFF] in the lanes with (l & OFF) == 0 and b[l] + b[l ^ OFF] in the others, OFF = 32 | 16: one swap of the
// halves (quarters) of the two registers and ONE addition -- the additions of `keep + shuffled(send)` (a + b == b + a)
template <int OFF>
__device__ __forceinline__ float rp_pair_sum(float a, float b) {
    static_assert(OFF == 32 || OFF == 16, "half / quarter swaps");
    rp_u2 r;
    if constexpr (OFF == 32) r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    else r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

__device__ __forceinline__ float rp_wave_allreduce(float s) {
    s = rp_pair_sum<32>(s, s);
    s = rp_pair_sum<16>(s, s);
    s = s + rp_lane_xor<8>(s);
    s = s + rp_lane_xor<4>(s);
    s = s + rp_lane_xor<2>(s);
    s = s + rp_lane_xor<1>(s);
    return s;
}

__device__ __forceinline__ float rp_lane_sum4(rp_f4 a) { return (a.x + a.y) + (a.z + a.w); }

// c + p * s per component, every component one fused multiply-add (fmaf): two v_pk_fma_f32 instead of four v_fma_f32
typedef float rp_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rp_f4 rp_fma4(rp_f4 p, rp_f4 x, rp_f4 c) {
    rp_f4 r;
    r.xy = __builtin_elementwise_fma(p.xy, x.xy, c.xy);
    r.zw = __builtin_elementwise_fma(p.zw, x.zw, c.zw);
    return r;
}
__device__ __forceinline__ rp_f4 rp_fma4s(rp_f4 p, float s, rp_f4 c) {
    const rp_f2 s2 = {s, s};
    rp_f4 r;
    r.xy = __builtin_elementwise_fma(p.xy, s2, c.xy);
    r.zw = __builtin_elementwise_fma(p.zw, s2, c.zw);
    return r;
}

// The butterflies of R rows at once (a "transposed" butterfly): sums, over the 64 lanes, R per-lane values; afterwards lane l
// holds the total of the row spelled by its lane bits, high bit first (rp_rows_allreduce_row).  Same pairings (l, l ^ 32),
// (l, l ^ 16), ... as rp_wave_allreduce -- the same additions per row, R - 1 + log2(64 / R) shuffles instead of 6 R.
// (one level per template instance: every array index is a compile-time constant whatever R is -- a loop over the levels
// is not always unrolled for R = 16 / 32 and then indexes the array at run time, i.e. through scratch)
template <int R, int W, int OFF>
__device__ __forceinline__ void rp_rows_level(float (&v)[R], int lane) {
    if constexpr (W > 1) {
        if constexpr (OFF >= 16) {
#pragma unroll
            for (int q = 0; q < W / 2; ++q) v[q] = rp_pair_sum<OFF>(v[q], v[q + W / 2]);
        } else {
            const bool hi = (lane & OFF) != 0;
#pragma unroll
            for (int q = 0; q < W / 2; ++q) {
                const float send = hi ? v[q] : v[q + W / 2];
                const float keep = hi ? v[q + W / 2] : v[q];
                v[q] = keep + rp_lane_xor<OFF>(send);
            }
        }
        rp_rows_level<R, W / 2, OFF / 2>(v, lane);
    }
}

template <int OFF>
__device__ __forceinline__ float rp_tail_allreduce(float s) {
    if constexpr (OFF >= 1) {
        if constexpr (OFF >= 16) s = rp_pair_sum<OFF>(s, s);
        else s = s + rp_lane_xor<OFF>(s);
        return rp_tail_allreduce<OFF / 2>(s);
    } else {
        return s;
    }
}

template <int R>
__device__ __forceinline__ float rp_rows_allreduce(float (&v)[R], int lane) {
    // sums over the 64 lanes of R per-lane values at once; afterwards lane l holds the total of value
    // (bits of l selected level by level); same pairings (l, l ^ 32), (l, l ^ 16), ... as rp_wave_allreduce
    static_assert((R & (R - 1)) == 0 && R >= 1 && R <= 64, "R is a power of two");
    rp_rows_level<R, R, 32>(v, lane);
    return rp_tail_allreduce<32 / R>(v[0]);
}


template <int R>
__device__ __forceinline__ int rp_rows_allreduce_row(int lane) {
    int r = 0;
    int width = R, off = 32;
#pragma unroll
    for (; width > 1; width >>= 1, off >>= 1) r += (lane & off) ? width / 2 : 0;
    return r;
}

// out[i] = (prev ? prev[i] : 0 +) (M s)_i for a matrix M [rows x n] in packed sparse form (rectipy_hip.h: rp_packed): the sparse
// step kernels' row dots with a pass-through "epilogue" (rnn_forward.hip; used by rp_forward_coupled and rp_backward_coupled)
struct rp_packed;
int rp_internal_packed_matvec(const rp_packed* M, int n, int rows, const float* s, const float* prev, float* out, hipStream_t st);

// edges with memory (rp_edge_mem, rnn_edges.hip): argument check; one step of the stage on the vector x -- *u receives the
// stage's output vector (DELAY / DELAY_FILTER: M->u; FILTER: the new state); the row-dot mat-vec of the filter (rnn_forward.hip)
struct rp_edge_mem;
int rp_internal_edge_mem_check(const rp_edge_mem* M);
int rp_internal_edge_mem_step(rp_edge_mem* M, const float* x, const float** u, hipStream_t st, int save_row);
int rp_internal_filter_matvec(const float* F, int64_t ldf, int n, const float* src, const float* x, float* dst, hipStream_t st);
