// rp_resident.h -- the machinery of the resident multi-step kernels (launch-bound sizes: n <= 1024): the forward
// step (rnn_forward.hip: resident_kernel) and the adjoint sweeps on the transposed copy (rnn_backward.hip: resident_adj_kernel for
// the rate models, resident_spk_adj_kernel for the spiking ones).
//
// Every wave keeps RW rows of the matrix in REGISTERS for the whole call; between steps the waves hand the vector the rows are
// multiplied with around themselves: the lane that finished element i stores ONE 8-byte granule {tag = step + 1, value}
// (agent scope, write-through), and every wave sweeps the n granules of that parity until all tags match -- the data is its
// own flag: no barrier, no separate flag, no fence; two granule buffers alternate, so a granule is only overwritten once every
// wave has consumed it.  Placement is for speed only (every shared word is an agent-scope access wherever the workgroups
// sit): with up to 8 workgroups the hand-off is 0.9 us per step faster when all of them are on ONE XCD
// (scripts/probes/granule_probe.hip: n = 512, no arithmetic: 1.29 against 2.21 us per step; 24 or 32 workgroups on one XCD
// are slower than spread out).  Then eight times the workgroups are launched; the first to arrive names its XCD
// (HW_REG_XCC_ID), the workgroups on that XCD take tickets -- the ticket is the work index -- and count themselves arrived; the
// others count themselves arrived and wait: they leave once all tickets are gone, or take what is left once EVERY workgroup
// has arrived (the XCD got too few: the roster completes anyway, a little slower per step).  Nothing is written before the
// roster is complete.  Every spin is bounded: a wave that waited for seconds (the GPU never ran all workgroups together) raises
// `fail` (the launch's own word: every wave that sees it leaves) and sets a sticky word in pinned host memory; the host learns of
// it without synchronising -- the library's next calls (rp_resident_poll) and rp_async_status look at that word.
#pragma once

#include <atomic>
#include <map>
#include <mutex>
#include <utility>

#include "rp_common.h"
#include "rp_dot.h"

struct rp_resident_ctl {
    unsigned long long* gran[2];  // [n] granules per parity
    int* fail;
    int* host_fail;               // pinned host memory, mapped: set (never cleared by the device) when a wave of ANY launch timed out
    unsigned* ctl;                // [0] the XCD (+ 1) of the first workgroup to arrive, [1] tickets taken, [2] workgroups arrived
    int need_blocks, one_xcd;
    unsigned spin_cap;
    int test_fault;               // (RECTIPY_AMD_RESIDENT_TEST_FAULT: element 0 is never handed on -- the tests' way to a timeout)
};

typedef __attribute__((address_space(1))) unsigned long long rp_gu64;
typedef __attribute__((address_space(1))) unsigned rp_gu32;
typedef __attribute__((address_space(1))) int rp_gi32;

constexpr int kResidentWaves = 4;                     // waves per workgroup
constexpr int kResidentUnavailable = 1;               // (internal: no stream-ordered scratch -- nothing launched, the caller launches per step)
constexpr int rp_resident_chunks(int n) { return (n + 255) >> 8; }   // 256-column chunks, the last one possibly partial
constexpr int rp_resident_rows_per_wave(int n) { return rp_resident_chunks(n) <= 2 ? 16 : (rp_resident_chunks(n) <= 4 ? 8 : 4); }

// the work index of this workgroup, or -1 (it leaves).  Ends with the workgroup's barrier.
__device__ __forceinline__ int rp_resident_join(const rp_resident_ctl& C) {
    __shared__ int block_s;
    if (threadIdx.x == 0) {
        int b = (int)blockIdx.x;
        if (C.one_xcd) {
            rp_gu32* const ctl = (rp_gu32*)C.ctl;
            rp_gi32* const fail = (rp_gi32*)C.fail;
            const unsigned need = (unsigned)C.need_blocks;
            const unsigned xcc = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu) + 1u;
            unsigned chosen = 0u;
            __hip_atomic_compare_exchange_strong(ctl, &chosen, xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (chosen == 0u) chosen = xcc;   // (this workgroup was the first)
            bool take = chosen == xcc;
            if (C.one_xcd == 2) take = take && (blockIdx.x & 31u) == 0u;   // (test seam: the XCD "gets" too few workgroups)
            if (take) b = (int)__hip_atomic_fetch_add(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ticket, THEN arrival
            __hip_atomic_fetch_add(ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!take) {   // another XCD: leave when the roster is complete, step in when everyone is here and it is not
                b = C.need_blocks;
                for (unsigned spins = 0;; ++spins) {
                    if (__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
                    if (__hip_atomic_load(ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) {
                        b = (int)__hip_atomic_fetch_add(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    if (spins > C.spin_cap || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (b < C.need_blocks) {   // the gate: all tickets taken
                for (unsigned spins = 0;; ++spins) {
                    if (__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
                    if (spins > C.spin_cap || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (C.host_fail) __hip_atomic_store(C.host_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        b = C.need_blocks;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        block_s = b;
    }
    __syncthreads();
    return block_s < C.need_blocks ? block_s : -1;
}

// the n values tagged `want` in the lane layout of the row dots (lane l: columns 256 m + 4 l .. + 3; zeros past n -- n % 4 == 0,
// so a lane's four columns are all there or all past the end); false: give up
template <int NC>
__device__ __forceinline__ bool rp_resident_sweep(const rp_resident_ctl& C, int parity, unsigned want, rp_f4 (&r)[NC], int lane, int n) {
    const rp_gu64* g = (const rp_gu64*)C.gran[parity] + 4 * lane;
    rp_gi32* const fail = (rp_gi32*)C.fail;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int m = 0; m < NC; ++m) {
            if (256 * m + 4 * lane < n) {
                unsigned long long q[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) q[c] = __hip_atomic_load(g + 256 * m + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int c = 0; c < 4; ++c) ok = ok && (unsigned)(q[c] >> 32) == want;
                r[m].x = __uint_as_float((unsigned)q[0]);
                r[m].y = __uint_as_float((unsigned)q[1]);
                r[m].z = __uint_as_float((unsigned)q[2]);
                r[m].w = __uint_as_float((unsigned)q[3]);
            } else {
                r[m] = rp_f4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (__all(ok)) return true;
        if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;   // (wave-uniform)
        if (spins > C.spin_cap) {
            if (lane == 0) {
                __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (C.host_fail) __hip_atomic_store(C.host_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return false;   // (every wave sees the flag at its next sweep and leaves)
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// the vector as it lies in memory, in the same layout (the first step of a call)
template <int NC>
__device__ __forceinline__ void rp_resident_load_vec(rp_f4 (&r)[NC], const float* v, int lane, int n) {
#pragma unroll
    for (int m = 0; m < NC; ++m) r[m] = 256 * m + 4 * lane < n ? rp_ld4(v + 256 * m + 4 * lane) : rp_f4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ void rp_resident_publish(const rp_resident_ctl& C, int parity, int row, unsigned tag, float value) {
    if (C.test_fault && row == 0) return;
    __hip_atomic_store((rp_gu64*)C.gran[parity] + row, ((unsigned long long)tag << 32) | __float_as_uint(value), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// eight rows (one stagger group: the same start chunk M0) against the chunks of the vector, in the contract's order (rp_dot.h:
// the MF full chunks from M0 on, wrapping; then the partial last chunk -- its lanes past n hold zeros: fma(0, 0, a) == a)
template <int NC, int MF, int M0, int ROWS = 8>
__device__ __forceinline__ void rp_resident_dot8(const rp_f4 (*w)[NC], const rp_f4 (&r)[NC], float* rs) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        rp_f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = (M0 + i) % (MF > 0 ? MF : 1);
            a = rp_fma4(w[j][m], r[m], a);
        }
        if constexpr (MF < NC) a = rp_fma4(w[j][NC - 1], r[NC - 1], a);
        rs[j] = rp_lane_sum4(a);
    }
}

// rows past n and the lanes of a partial chunk past n hold zeros
template <int NC, int RW>
__device__ __forceinline__ void rp_resident_load_rows(rp_f4 (&w)[RW][NC], const float* W, int64_t ldw, int row_w0, int lane, int n) {
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int m = 0; m < NC; ++m)
            w[j][m] = (row_w0 + j < n && 256 * m + 4 * lane < n) ? rp_ld4(W + (int64_t)(row_w0 + j) * ldw + 256 * m + 4 * lane)
                                                                 : rp_f4{0.f, 0.f, 0.f, 0.f};
}

// the totals of the wave's RW rows: afterwards lane l holds the total of row rp_rows_allreduce_row<RW>(l) of the wave.
// PARTIAL: the last of the NC chunks is a partial one (n % 256 != 0)
// STAGGER = false: every row starts at chunk 0 (the RLS passes)
template <int NC, int RW, bool PARTIAL, bool STAGGER = true>
__device__ __forceinline__ float rp_resident_dots(const rp_f4 (&w)[RW][NC], const rp_f4 (&r)[NC], int global_row_w0, int lane) {
    static_assert(RW == 4 || RW == 8 || RW == 16, "half a stagger group of eight rows, one, or two per wave");
    static_assert(NC <= 8, "start chunks 0 .. 7");
    constexpr int MF = PARTIAL ? NC - 1 : NC;   // full chunks
    constexpr int G = RW < 8 ? RW : 8;          // rows that share a start chunk (a wave's rows never straddle a group of eight)
    float rs[RW];
#pragma unroll
    for (int h = 0; h < RW / G; ++h) {
        if constexpr (MF <= 1 || !STAGGER) {
            rp_resident_dot8<NC, MF, 0, G>(w + G * h, r, rs + G * h);
        } else {
            const int m0 = (int)(((unsigned)((global_row_w0 + G * h) >> 3) * kStagger) % (unsigned)MF);   // wave-uniform
            switch (m0) {   // (m0 < MF: the cases past MF - 1 are never taken)
                case 0: rp_resident_dot8<NC, MF, 0, G>(w + G * h, r, rs + G * h); break;
                case 1: rp_resident_dot8<NC, MF, 1, G>(w + G * h, r, rs + G * h); break;
                case 2: rp_resident_dot8<NC, MF, 2 % MF, G>(w + G * h, r, rs + G * h); break;
                case 3: rp_resident_dot8<NC, MF, 3 % MF, G>(w + G * h, r, rs + G * h); break;
                case 4: rp_resident_dot8<NC, MF, 4 % MF, G>(w + G * h, r, rs + G * h); break;
                case 5: rp_resident_dot8<NC, MF, 5 % MF, G>(w + G * h, r, rs + G * h); break;
                case 6: rp_resident_dot8<NC, MF, 6 % MF, G>(w + G * h, r, rs + G * h); break;
                default: rp_resident_dot8<NC, MF, 7 % MF, G>(w + G * h, r, rs + G * h); break;
            }
        }
    }
    return rp_rows_allreduce<RW>(rs, lane);
}

// ---- coupled circuits (rnn_forward.hip: resident_coupled_kernel): a wave keeps its rows of SEVERAL matrices -- the member's own
// and its in-edges' --, each in a slot of NC chunks whatever the matrix's own width (<= 256 NC columns) -------------------------
// rows past `rows` and columns past `cols` hold zeros
template <int NC, int RW>
__device__ __forceinline__ void rp_resident_load_rows_rect(rp_f4 (&w)[RW][NC], const float* W, int64_t ldw, int row_w0, int lane, int rows,
                                                           int cols) {
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int m = 0; m < NC; ++m)
            w[j][m] = (row_w0 + j < rows && 256 * m + 4 * lane < cols) ? rp_ld4(W + (int64_t)(row_w0 + j) * ldw + 256 * m + 4 * lane)
                                                                       : rp_f4{0.f, 0.f, 0.f, 0.f};
}
// the row totals of a slot that holds a matrix of `cols` columns, in the order of THAT width's contract: a slot chunk past the
// matrix's last one holds zeros and may come last (fma(0, 0, a) == a); the full chunks rotate among themselves
// (slots of three or four chunks: exactly the chunks the matrix has -- MF full ones from the row group's start chunk on, wrapping,
// then the partial one at index MF; the slot's empty chunks are not visited)
template <int NC, int MF, int M0, int ROWS, bool HASP>
__device__ __forceinline__ void rp_resident_slot_rows(const rp_f4 (*w)[NC], const rp_f4 (&r)[NC], float* rs) {
    static_assert(MF + (HASP ? 1 : 0) <= NC, "the matrix fits the slot");
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        rp_f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = (M0 + i) % (MF > 0 ? MF : 1);
            a = rp_fma4(w[j][m], r[m], a);
        }
        if constexpr (HASP) a = rp_fma4(w[j][MF], r[MF], a);
        rs[j] = rp_lane_sum4(a);
    }
}
template <int NC, int RW, int MF, bool HASP>
__device__ __forceinline__ float rp_resident_slot_dots(const rp_f4 (&w)[RW][NC], const rp_f4 (&r)[NC], int global_row_w0, int lane) {
    static_assert(RW == 4 || RW == 8, "half a stagger group of eight rows or one per wave");
    float rs[RW];
    if constexpr (MF <= 1) {
        rp_resident_slot_rows<NC, MF, 0, RW, HASP>(w, r, rs);
    } else {
        const int m0 = (int)(((unsigned)(global_row_w0 >> 3) * kStagger) % (unsigned)MF);   // wave-uniform
        switch (m0) {   // (m0 < MF)
            case 0: rp_resident_slot_rows<NC, MF, 0, RW, HASP>(w, r, rs); break;
            case 1: rp_resident_slot_rows<NC, MF, 1, RW, HASP>(w, r, rs); break;
            case 2: rp_resident_slot_rows<NC, MF, 2 % MF, RW, HASP>(w, r, rs); break;
            default: rp_resident_slot_rows<NC, MF, 3 % MF, RW, HASP>(w, r, rs); break;
        }
    }
    return rp_rows_allreduce<RW>(rs, lane);
}
template <int NC, int RW>
__device__ __forceinline__ float rp_resident_dots_slot(const rp_f4 (&w)[RW][NC], const rp_f4 (&r)[NC], int cols, int global_row_w0, int lane) {
    static_assert(NC >= 1 && NC <= 4, "slots of one to four chunks");
    if constexpr (NC == 1) {
        return rp_resident_dots<1, RW, false>(w, r, global_row_w0, lane);
    } else if constexpr (NC == 2) {
        if (cols == 512) return rp_resident_dots<2, RW, false>(w, r, global_row_w0, lane);   // two full chunks: the start chunk rotates
        return rp_resident_dots<2, RW, true>(w, r, global_row_w0, lane);                      // chunk 0, then the partial (or empty) chunk 1
    } else {
        const int mf = cols >> 8;
        const bool hasp = (cols & 255) != 0;   // (wave-uniform, as mf)
#define RP_SLOT_(MF_, P_) return rp_resident_slot_dots<NC, RW, MF_, P_>(w, r, global_row_w0, lane)
        if (hasp) {
            switch (mf) {
                case 0: RP_SLOT_(0, true);
                case 1: RP_SLOT_(1, true);
                case 2: RP_SLOT_(2, true);
                default:
                    if constexpr (NC == 4) RP_SLOT_(3, true);
                    else RP_SLOT_(2, true);   // (NC == 3: never reached -- three full chunks have no partial one behind them)
            }
        } else {
            switch (mf) {
                case 1: RP_SLOT_(1, false);
                case 2: RP_SLOT_(2, false);
                case 3: RP_SLOT_(3, false);
                default:
                    if constexpr (NC == 4) RP_SLOT_(4, false);
                    else RP_SLOT_(3, false);
            }
        }
#undef RP_SLOT_
    }
}

// ---- the two row-dot contracts as layouts of a wave's registers ------------------------------------------------------------------
// 16-byte path (rp_dot_rows_v4: n % 4 == 0, ld % 4 == 0): lane l holds columns 256 m + 4 l .. + 3 of NC chunks
template <int NC, int RW_, bool PARTIAL>
struct rp_layout_v4 {
    static constexpr int RW = RW_;
    struct Vec {
        rp_f4 c[NC];
    };
    struct Rows {
        rp_f4 w[RW_][NC];
    };
    static __device__ __forceinline__ void load_rows(Rows& R, const float* W, int64_t ldw, int row_w0, int lane, int n) {
        rp_resident_load_rows<NC, RW_>(R.w, W, ldw, row_w0, lane, n);
    }
    static __device__ __forceinline__ void load_vec(Vec& v, const float* src, int lane, int n) { rp_resident_load_vec<NC>(v.c, src, lane, n); }
    static __device__ __forceinline__ bool sweep(const rp_resident_ctl& C, int parity, unsigned want, Vec& v, int lane, int n) {
        return rp_resident_sweep<NC>(C, parity, want, v.c, lane, n);
    }
    static __device__ __forceinline__ float dots(const Rows& R, const Vec& v, int global_row_w0, int lane) {
        return rp_resident_dots<NC, RW_, PARTIAL>(R.w, v.c, global_row_w0, lane);
    }
    // ---- the RLS sweep's operations in this layout ----
    static __device__ __forceinline__ float dots_plain(const Rows& R, const Vec& v, int lane) {   // every row from chunk 0 on
        return rp_resident_dots<NC, RW_, PARTIAL, false>(R.w, v.c, 0, lane);
    }
    // rls_scalars_kernel's dot product: leaf j = fmaf(u_j, x_j, 0), then the 1024-leaf tree -- levels 512 / 256: a lane's own
    // chunks, levels 128 .. 4: the wave butterfly, levels 2 / 1: (x + z) + (y + w) of the lane's components
    static __device__ __forceinline__ float tree_dot(const Vec& u, const Vec& x) {
        const rp_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
        rp_f4 a[NC];
#pragma unroll
        for (int m = 0; m < NC; ++m) a[m] = rp_fma4(u.c[m], x.c[m], zero4);
        rp_f4 b0 = a[0], b1 = zero4;
        if constexpr (NC > 2) b0 = a[0] + a[2];
        if constexpr (NC > 1) b1 = a[1];
        if constexpr (NC > 3) b1 = a[1] + a[3];
        rp_f4 c = b0;
        if constexpr (NC > 1) c = b0 + b1;
        c.x = rp_wave_allreduce(c.x);
        c.y = rp_wave_allreduce(c.y);
        c.z = rp_wave_allreduce(c.z);
        c.w = rp_wave_allreduce(c.w);
        return (c.x + c.z) + (c.y + c.w);
    }
    static __device__ __forceinline__ void add_scaled(Vec& w, float a, const Vec& z) {   // w = w + a z (product, then sum)
#pragma unroll
        for (int m = 0; m < NC; ++m) {
            w.c[m].x = w.c[m].x + a * z.c[m].x;
            w.c[m].y = w.c[m].y + a * z.c[m].y;
            w.c[m].z = w.c[m].z + a * z.c[m].z;
            w.c[m].w = w.c[m].w + a * z.c[m].w;
        }
    }
    static __device__ __forceinline__ void row_sub_scaled(Rows& R, int j, float a, const Vec& z) {   // row j: p = p - a z
#pragma unroll
        for (int m = 0; m < NC; ++m) {
            R.w[j][m].x = R.w[j][m].x - a * z.c[m].x;
            R.w[j][m].y = R.w[j][m].y - a * z.c[m].y;
            R.w[j][m].z = R.w[j][m].z - a * z.c[m].z;
            R.w[j][m].w = R.w[j][m].w - a * z.c[m].w;
        }
    }
    static __device__ __forceinline__ void zero_vec(Vec& v) {
#pragma unroll
        for (int m = 0; m < NC; ++m) v.c[m] = rp_f4{0.f, 0.f, 0.f, 0.f};
    }
    static __device__ __forceinline__ void store_vec(const Vec& v, float* dst, int lane, int n) {
#pragma unroll
        for (int m = 0; m < NC; ++m)
            if (256 * m + 4 * lane < n) *reinterpret_cast<rp_f4*>(dst + 256 * m + 4 * lane) = v.c[m];
    }
    static __device__ __forceinline__ void store_rows(const Rows& R, float* P, int64_t ldp, int row_w0, int lane, int n) {
#pragma unroll
        for (int j = 0; j < RW_; ++j)
#pragma unroll
            for (int m = 0; m < NC; ++m)
                if (row_w0 + j < n && 256 * m + 4 * lane < n) *reinterpret_cast<rp_f4*>(P + (int64_t)(row_w0 + j) * ldp + 256 * m + 4 * lane) = R.w[j][m];
    }
};

// 4-byte path (rp_dot_rows_s1: any n and ld): lane l holds columns l + 64 k of KC strides (zeros past n), ONE accumulator per row
// fed in ascending k, then the wave butterfly
template <int KC, int RW_>
struct rp_layout_s1 {
    static constexpr int RW = RW_;
    struct Vec {
        float c[KC];
    };
    struct Rows {
        float w[RW_][KC];
    };
    static __device__ __forceinline__ void load_rows(Rows& R, const float* W, int64_t ldw, int row_w0, int lane, int n) {
#pragma unroll
        for (int j = 0; j < RW_; ++j)
#pragma unroll
            for (int k = 0; k < KC; ++k)
                R.w[j][k] = (row_w0 + j < n && lane + 64 * k < n) ? W[(int64_t)(row_w0 + j) * ldw + lane + 64 * k] : 0.f;
    }
    static __device__ __forceinline__ void load_vec(Vec& v, const float* src, int lane, int n) {
#pragma unroll
        for (int k = 0; k < KC; ++k) v.c[k] = lane + 64 * k < n ? src[lane + 64 * k] : 0.f;
    }
    static __device__ __forceinline__ bool sweep(const rp_resident_ctl& C, int parity, unsigned want, Vec& v, int lane, int n) {
        const rp_gu64* g = (const rp_gu64*)C.gran[parity] + lane;
        rp_gi32* const fail = (rp_gi32*)C.fail;
        for (unsigned spins = 0;; ++spins) {
            bool ok = true;
            unsigned long long q[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k)
                q[k] = lane + 64 * k < n ? __hip_atomic_load(g + 64 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)want << 32);
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                ok = ok && (unsigned)(q[k] >> 32) == want;
                v.c[k] = __uint_as_float((unsigned)q[k]);
            }
            if (__all(ok)) return true;
            if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spins > C.spin_cap) {
                if (lane == 0) {
                    __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (C.host_fail) __hip_atomic_store(C.host_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    static __device__ __forceinline__ float dots(const Rows& R, const Vec& v, int /*global_row_w0*/, int lane) {
        float rs[RW_];
#pragma unroll
        for (int j = 0; j < RW_; ++j) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < KC; ++k) a = fmaf(R.w[j][k], v.c[k], a);
            rs[j] = a;
        }
        return rp_rows_allreduce<RW_>(rs, lane);
    }
    // ---- the RLS sweep's operations in this layout ----
    static __device__ __forceinline__ float dots_plain(const Rows& R, const Vec& v, int lane) { return dots(R, v, 0, lane); }
    // rls_scalars_kernel's tree with leaf j = l + 64 k in lane l: levels 512 .. 64 add a lane's own strides (k + 8, + 4, + 2, + 1),
    // levels 32 .. 1 are the wave butterfly
    static __device__ __forceinline__ float tree_dot(const Vec& u, const Vec& x) {
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = k < KC ? fmaf(u.c[k < KC ? k : 0], x.c[k < KC ? k : 0], 0.f) : 0.f;
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int k = 0; k < w; ++k)
                if (k + w < KC) a[k] = a[k] + a[k + w];   // (absent strides are zeros: x + 0 == x)
        return rp_wave_allreduce(a[0]);
    }
    static __device__ __forceinline__ void add_scaled(Vec& w, float a, const Vec& z) {
#pragma unroll
        for (int k = 0; k < KC; ++k) w.c[k] = w.c[k] + a * z.c[k];
    }
    static __device__ __forceinline__ void row_sub_scaled(Rows& R, int j, float a, const Vec& z) {
#pragma unroll
        for (int k = 0; k < KC; ++k) R.w[j][k] = R.w[j][k] - a * z.c[k];
    }
    static __device__ __forceinline__ void zero_vec(Vec& v) {
#pragma unroll
        for (int k = 0; k < KC; ++k) v.c[k] = 0.f;
    }
    static __device__ __forceinline__ void store_vec(const Vec& v, float* dst, int lane, int n) {
#pragma unroll
        for (int k = 0; k < KC; ++k)
            if (lane + 64 * k < n) dst[lane + 64 * k] = v.c[k];
    }
    static __device__ __forceinline__ void store_rows(const Rows& R, float* P, int64_t ldp, int row_w0, int lane, int n) {
#pragma unroll
        for (int j = 0; j < RW_; ++j)
#pragma unroll
            for (int k = 0; k < KC; ++k)
                if (row_w0 + j < n && lane + 64 * k < n) P[(int64_t)(row_w0 + j) * ldp + lane + 64 * k] = R.w[j][k];
    }
};
constexpr int rp_resident_strides_s1(int n) {   // 64-column strides per lane, rounded up to a power of two (zero-padded)
    const int k = (n + 63) / 64;
    return k <= 1 ? 1 : (k <= 2 ? 2 : (k <= 4 ? 4 : (k <= 8 ? 8 : 16)));
}
constexpr int rp_resident_rows_per_wave_s1(int n) { return rp_resident_strides_s1(n) >= 16 ? 8 : 16; }

#define RP_UNPAREN(...) __VA_ARGS__
// the launch: LAUNCH((layout type)) for n columns on the 16-byte path (v4 = true) or the 4-byte path
#define RP_RESIDENT_DISPATCH_LAYOUT(n_, v4_, LAUNCH)                                       \
    if (v4_) {                                                                             \
        switch (rp_resident_chunks(n_) * 2 + ((n_) % 256 != 0 ? 1 : 0)) {                 \
            case 2: LAUNCH((rp_layout_v4<1, 16, false>)); break;                           \
            case 3: LAUNCH((rp_layout_v4<1, 16, true>)); break;                            \
            case 4: LAUNCH((rp_layout_v4<2, 16, false>)); break;                           \
            case 5: LAUNCH((rp_layout_v4<2, 16, true>)); break;                            \
            case 6: LAUNCH((rp_layout_v4<3, 8, false>)); break;                            \
            case 7: LAUNCH((rp_layout_v4<3, 8, true>)); break;                             \
            case 8: LAUNCH((rp_layout_v4<4, 8, false>)); break;                            \
            default: LAUNCH((rp_layout_v4<4, 8, true>)); break;                            \
        }                                                                                  \
    } else {                                                                               \
        switch (rp_resident_strides_s1(n_)) {                                              \
            case 1: LAUNCH((rp_layout_s1<1, 16>)); break;                                  \
            case 2: LAUNCH((rp_layout_s1<2, 16>)); break;                                  \
            case 4: LAUNCH((rp_layout_s1<4, 16>)); break;                                  \
            case 8: LAUNCH((rp_layout_s1<8, 16>)); break;                                  \
            default: LAUNCH((rp_layout_s1<16, 8>)); break;                                 \
        }                                                                                  \
    }

// ... the same with the wide layouts of the forward step and the adjoint sweeps (1024 < n <= 1536: 5 or 6 chunks, four rows per wave;
// the RLS sweep keeps P AND the weights in registers and stays at n <= 1024)
#define RP_RESIDENT_DISPATCH_LAYOUT_WIDE(n_, v4_, LAUNCH)                                  \
    if ((v4_) && rp_resident_chunks(n_) > 4) {                                             \
        switch (rp_resident_chunks(n_) * 2 + ((n_) % 256 != 0 ? 1 : 0)) {                 \
            case 10: LAUNCH((rp_layout_v4<5, 4, false>)); break;                           \
            case 11: LAUNCH((rp_layout_v4<5, 4, true>)); break;                            \
            case 12: LAUNCH((rp_layout_v4<6, 4, false>)); break;                           \
            default: LAUNCH((rp_layout_v4<6, 4, true>)); break;                            \
        }                                                                                  \
    } else {                                                                               \
        RP_RESIDENT_DISPATCH_LAYOUT(n_, v4_, LAUNCH)                                       \
    }

// ---- host side ------------------------------------------------------------------------------------------------------------
// The sticky failure word: pinned host memory the device can write (one int per process), looked at by the library's next calls
// (rp_resident_poll) and by rp_async_status -- no copy, no synchronisation.
inline int* rp_resident_host_word(int** device_view) {
    struct Word {
        int* host = nullptr;
        int* dev = nullptr;
        Word() {
            if (hipHostMalloc(reinterpret_cast<void**>(&host), 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
                host = nullptr;
                return;
            }
            *host = 0;
            if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0) != hipSuccess) dev = nullptr;
        }
    };
    static Word w;
    if (device_view) *device_view = w.dev;
    return w.host;
}
// The scratch of a resident launch (granules, failure word, roster; for a coupled circuit the members' argument blocks): ONE block
// per (device, stream), allocated on the stream's first resident launch and kept.  Launches on a stream follow each other, and each
// starts with a stream-ordered memset, so they can share it.  (r05, measured -- scripts/probes/async_alloc_probe.hip: with
// hipMallocAsync / hipFreeAsync around every launch the host waited for the PREVIOUS launch of the stream in every call, 208 us
// per iteration for a 200 us kernel; memsetAsync + launch on a kept block: 5 us.  The library's calls were not asynchronous.)
constexpr size_t kResidentScratchBytes = 192u << 10;
// Two host threads may launch on the SAME stream (the default stream is every thread's): the zeroing and the launch that uses the
// block must reach the stream as a pair, so a block is held (`hold`: its mutex taken) from rp_resident_begin* to rp_resident_end --
// or rp_resident_release on a path that ends without a launch.
struct rp_resident_block {
    void* p = nullptr;
    unsigned calls = 0;
    std::mutex seq;
};
struct rp_resident_table {
    std::mutex mu;
    std::map<std::pair<int, hipStream_t>, rp_resident_block> blocks;
};
inline rp_resident_table& rp_resident_blocks() {
    static rp_resident_table t;
    return t;
}
inline void rp_resident_release(hipStream_t st) {
    rp_resident_table& t = rp_resident_blocks();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    rp_resident_block* b = nullptr;
    {
        std::lock_guard<std::mutex> lock(t.mu);
        auto it = t.blocks.find({dev, st});
        if (it != t.blocks.end()) b = &it->second;
    }
    if (b) b->seq.unlock();
}
inline int rp_resident_scratch(hipStream_t st, size_t bytes, void** out, unsigned* serial = nullptr) {
    rp_resident_table& t = rp_resident_blocks();
    *out = nullptr;
    if (bytes > kResidentScratchBytes) return RP_ERR_HIP;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RP_ERR_HIP;
    rp_resident_block* b = nullptr;
    {
        std::lock_guard<std::mutex> lock(t.mu);
        auto it = t.blocks.find({dev, st});
        if (it == t.blocks.end()) {
            if (t.blocks.size() >= 256) return RP_ERR_HIP;   // (a process that keeps making streams: launches per step from here on)
            void* p = nullptr;
            if (hipMalloc(&p, kResidentScratchBytes) != hipSuccess) {
                (void)hipGetLastError();
                return RP_ERR_HIP;
            }
            it = t.blocks.try_emplace(std::make_pair(dev, st)).first;   // (map nodes do not move: the mutex stays where it is)
            it->second.p = p;
        }
        b = &it->second;
    }
    b->seq.lock();   // held until rp_resident_end / rp_resident_release
    *out = b->p;
    if (serial) *serial = b->calls++;   // (the stream's n-th launch with a copied argument block: rotating slots)
    return RP_OK;
}
// the stream's scratch (granules, failure word, roster), zeroed in stream order: a tag is never 0
inline int rp_resident_begin(int n, hipStream_t st, rp_resident_ctl* C, void** ws, bool v4 = true) {
    const size_t bytes = (size_t)2 * n * sizeof(unsigned long long) + 64;
    *ws = nullptr;
    if (const char* e = getenv("RECTIPY_AMD_RESIDENT_TEST_FAULT"); e && e[0] == '2') return RP_ERR_HIP;   // (test seam: no scratch)
    if (rp_resident_scratch(st, bytes, ws) != RP_OK) return RP_ERR_HIP;
    if (hipMemsetAsync(*ws, 0, bytes, st) != hipSuccess) {
        rp_resident_release(st);
        return RP_ERR_HIP;
    }
    C->gran[0] = static_cast<unsigned long long*>(*ws);
    C->gran[1] = C->gran[0] + n;
    C->fail = reinterpret_cast<int*>(C->gran[1] + n);
    C->ctl = reinterpret_cast<unsigned*>(C->fail) + 4;
    (void)rp_resident_host_word(&C->host_fail);
    const int rw = v4 ? rp_resident_rows_per_wave(n) : rp_resident_rows_per_wave_s1(n);
    const int waves = (n + rw - 1) / rw;
    C->need_blocks = (waves + kResidentWaves - 1) / kResidentWaves;
    C->one_xcd = C->need_blocks <= 8 ? 1 : 0;
    if (const char* e = getenv("RECTIPY_AMD_RESIDENT_XCD"); e && e[0] == '0') C->one_xcd = 0;
    else if (e && e[0] == '2' && C->one_xcd) C->one_xcd = 2;   // (tests: most of the roster has to come from the other XCDs)
    C->spin_cap = 1u << 22;   // polls of about a microsecond each: seconds
    C->test_fault = 0;
    if (const char* e = getenv("RECTIPY_AMD_RESIDENT_TEST_FAULT"); e && e[0] == '1') {   // (test seam: a peer that never delivers)
        C->test_fault = 1;
        C->spin_cap = 4000;
    }
    return RP_OK;
}
// the same for a coupled circuit: `streams` granule vectors of `cap` granules in two parities each, `extra` bytes behind them
// (the members' argument blocks); gran[0] = the first granule, gran[1] unused
inline int rp_resident_begin_streams(int streams, int cap, int need_blocks, size_t extra, hipStream_t st, rp_resident_ctl* C, void** ws,
                                     void** extra_ptr) {
    // the argument blocks are copied from host memory in stream order; consecutive launches still use different slots
    constexpr unsigned kSlots = 8;
    const size_t gran_bytes = (size_t)2 * streams * cap * sizeof(unsigned long long);
    const size_t slot_bytes = (extra + 63) & ~(size_t)63;
    const size_t bytes = gran_bytes + 64 + kSlots * slot_bytes;
    unsigned serial = 0;
    *ws = nullptr;
    if (const char* e = getenv("RECTIPY_AMD_RESIDENT_TEST_FAULT"); e && e[0] == '2') return RP_ERR_HIP;   // (test seam: no scratch)
    if (rp_resident_scratch(st, bytes, ws, &serial) != RP_OK) return RP_ERR_HIP;
    if (hipMemsetAsync(*ws, 0, gran_bytes + 64, st) != hipSuccess) {
        rp_resident_release(st);
        return RP_ERR_HIP;
    }
    C->gran[0] = static_cast<unsigned long long*>(*ws);
    C->gran[1] = nullptr;
    C->fail = reinterpret_cast<int*>(static_cast<char*>(*ws) + gran_bytes);
    C->ctl = reinterpret_cast<unsigned*>(C->fail) + 4;
    *extra_ptr = static_cast<char*>(*ws) + gran_bytes + 64 + (serial % kSlots) * slot_bytes;
    (void)rp_resident_host_word(&C->host_fail);
    C->need_blocks = need_blocks;
    C->one_xcd = need_blocks <= 8 ? 1 : 0;
    if (const char* e = getenv("RECTIPY_AMD_RESIDENT_XCD"); e && e[0] == '0') C->one_xcd = 0;
    else if (e && e[0] == '2' && C->one_xcd) C->one_xcd = 2;
    C->spin_cap = 1u << 22;
    C->test_fault = 0;
    if (const char* e = getenv("RECTIPY_AMD_RESIDENT_TEST_FAULT"); e && e[0] == '1') {
        C->test_fault = 1;
        C->spin_cap = 4000;
    }
    return RP_OK;
}
inline unsigned rp_resident_grid(const rp_resident_ctl& C) { return (unsigned)(C.one_xcd ? 8 * C.need_blocks + 8 : C.need_blocks); }

inline int rp_resident_poll() {   // a resident launch that timed out since the last look?
    volatile int* word = rp_resident_host_word(nullptr);
    if (word && *word != 0) {
        *word = 0;
        RP_FAIL(RP_ERR_HIP, "a resident multi-step launch timed out waiting for a peer wave (its workgroups were never resident together: is "
                            "another kernel holding the GPU?); its results are not valid.  RECTIPY_AMD_RESIDENT=0 selects one launch per step");
    }
    return RP_OK;
}

// after the launch -- no synchronisation (the scratch stays with the stream)
inline int rp_resident_end(const rp_resident_ctl& C, void* ws, hipStream_t st, const char* what) {
    (void)C;
    (void)ws;
    const hipError_t launched = hipGetLastError();
    rp_resident_release(st);
    if (launched != hipSuccess) RP_FAIL(RP_ERR_HIP, "%s: %s", what, hipGetErrorString(launched));
    return RP_OK;
}

// n <= 1024 on the 16-byte path
inline bool rp_resident_shape(int n, int64_t ld) { return n >= 4 && n <= 1024 && n % 4 == 0 && ld % 4 == 0; }
// ... the forward step and the adjoint sweeps on the 16-byte path: up to 1536 (four rows of up to six chunks per wave).  Every wave
// reads the whole vector every step, so the hand-off grows with n while the launch it replaces does not: measured break-even at
// n = 1800 for the forward (5.3 against 5.5 us per step at 1792; 4.7 / 5.5 at 1536), n = 2048 loses in fit_bptt (12.4 / 11.5)
inline bool rp_resident_shape_wide(int n, int64_t ld) { return n >= 4 && n <= 1536 && n % 4 == 0 && ld % 4 == 0; }
// ... or on the 4-byte path (any n, any leading dimension)
inline bool rp_resident_shape_any(int n) { return n >= 1 && n <= 1024; }

inline bool rp_resident_enabled() {
    const char* e = getenv("RECTIPY_AMD_RESIDENT");   // (read per call: tests compare both paths)
    return !(e && e[0] == '0');
}
