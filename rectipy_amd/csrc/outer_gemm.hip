// outer_gemm.hip -- weight-gradient contraction of one BPTT window on the matrix cores.
//
// Autograd of `weights @ r` (reference: the generated field called at rectipy/nodes.py:169,
// differentiated by Network._bptt_step, rectipy/network.py:1126-1130) accumulates one
// n x n outer product per step.  Here the adjoint sweep stores u_{t+1} = k*lambda_{t+1}
// and the forward pass stores r_t, and the whole window collapses into ONE GEMM
//     C[n x n] += alpha * A^T B,   A = U [T x n],  B = R [T x n]
// (2 n^2 T flop), computed with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: bit for bit a
// t-ordered fmaf chain, so the CPU oracle can mirror it).
//
// Tiling: 128 x 128 output tile per 256-thread block, 4 waves as 2 x 2, each wave a
// 64 x 64 sub-tile = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator registers).  The
// reduction dimension t is staged through LDS in slices of BK = 16 rows; both operands
// are "t-major" so a slice row is a contiguous 512-byte global read and the MFMA A/B
// fragments (lane l -> row l&31 of the tile at k = l>>5) are conflict-free ds_read_b32.

#include <atomic>

#include "rp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16;  // BK = 32 measured 2.5 % slower (scripts/bench_gemm.py)
constexpr int REPS = BK / 8;  // 256 threads move 8 rows x 128 columns per repetition
constexpr int kThreads = 256;

template <bool VEC4>
__device__ __forceinline__ void load_slice(const float* __restrict__ G, int64_t ld, int n, int T, int t0, int c0,
                                           int tid, rp_f4 (&reg)[REPS]) {
#pragma unroll
    for (int rep = 0; rep < REPS; ++rep) {
        const int row = (tid >> 5) + 8 * rep;  // 0..BK-1
        const int c = c0 + 4 * (tid & 31);
        const int t = t0 + row;
        rp_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < T) {
            const float* p = G + (int64_t)t * ld + c;
            if constexpr (VEC4) {
                if (c < n) v = rp_ld4(p);
            } else {
                if (c + 0 < n) v.x = p[0];
                if (c + 1 < n) v.y = p[1];
                if (c + 2 < n) v.z = p[2];
                if (c + 3 < n) v.w = p[3];
            }
        }
        reg[rep] = v;
    }
}

__device__ __forceinline__ void store_slice(float* S, int tid, const rp_f4 (&reg)[REPS]) {
#pragma unroll
    for (int rep = 0; rep < REPS; ++rep) {
        const int row = (tid >> 5) + 8 * rep;
        *reinterpret_cast<rp_f4*>(S + row * BM + 4 * (tid & 31)) = reg[rep];
    }
}

template <bool VEC4>
__global__ __launch_bounds__(kThreads) void outer_accum_kernel(int m, int n, int T, float alpha,
                                                               const float* __restrict__ A, int64_t lda,
                                                               const float* __restrict__ B, int64_t ldb,
                                                               float* __restrict__ C, int64_t ldc, int accumulate) {
    __shared__ __attribute__((aligned(16))) float As[BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (n + BN - 1) / BN;
    const int tiles_m = (m + BM - 1) / BM;
    int ti, tj;
    if (((tiles_m | tiles_n) & 7) == 0 && (((tiles_m >> 3) * (tiles_n >> 3)) & 7) == 0) {
        // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  Give
        // every XCD whole 8 x 8 super-tiles of output tiles in sequence, so that the ~64 tiles resident on an XCD at a
        // time share 8 A panels and 8 B panels and walk the reduction dimension roughly in step.
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        const int super = (l >> 6) * 8 + xcd, w = l & 63;
        const int super_n = tiles_n >> 3;
        ti = (super / super_n) * 8 + (w >> 3);
        tj = (super % super_n) * 8 + (w & 7);
    } else {
        ti = blockIdx.x / tiles_n;
        tj = blockIdx.x % tiles_n;
    }
    const int i0 = ti * BM;
    const int j0 = tj * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    rp_f4 ra[REPS], rb[REPS];
    load_slice<VEC4>(A, lda, m, T, 0, i0, tid, ra);
    load_slice<VEC4>(B, ldb, n, T, 0, j0, tid, rb);

    const int lk = lane >> 5;   // k offset of this lane inside an MFMA (0/1)
    const int lr = lane & 31;   // row (A) / column (B) inside the 32-wide tile
    for (int t0 = 0; t0 < T; t0 += BK) {
        __syncthreads();  // previous slice fully consumed
        store_slice(As, tid, ra);
        store_slice(Bs, tid, rb);
        __syncthreads();
        if (t0 + BK < T) {  // prefetch the next slice into registers behind the MFMAs
            load_slice<VEC4>(A, lda, m, T, t0 + BK, i0, tid, ra);
            load_slice<VEC4>(B, ldb, n, T, t0 + BK, j0, tid, rb);
        }
        // all operand fragments of the slice first (32 LDS reads, one exposed LDS latency per 32 MFMAs instead of one per
        // four: the compiler otherwise waits for each k-pair's reads right in front of its MFMAs), then the MFMAs back to back
        float a0[BK / 2], a1[BK / 2], b0[BK / 2], b1[BK / 2];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float* ap = As + (kk + lk) * BM + wm * 64 + lr;
            const float* bp = Bs + (kk + lk) * BN + wn * 64 + lr;
            a0[kk / 2] = ap[0];
            a1[kk / 2] = ap[32];
            b0[kk / 2] = bp[0];
            b1[kk / 2] = bp[32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BK / 2; ++q) {  // t ascending: the chain order of every element is unchanged
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b0[q], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b1[q], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b0[q], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b1[q], acc[1][1], 0, 0, 0);
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = j0 + wn * 64 + ni * 32 + lr;
            // the 16 elements of an MFMA tile: all loads first, then all stores (written as `*cp = *cp + ...` per element
            // the compiler must assume that a store aliases the next load and serialises 64 global round trips)
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                old[r] = (accumulate && row < m && col < n) ? C[(int64_t)row * ldc + col] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < m && col < n) C[(int64_t)row * ldc + col] = old[r] + alpha * acc[mi][ni][r];
            }
        }
}

// ---- the same GEMM with LDS-DMA staging (full 128 x 128 tiles only: m % 128 == 0, n % 128 == 0) -----------------------
// r03 PMC of the kernel above at n = 32768, T = 1000: 2.38 GHz, MFMA pipe 77 % busy, waves 16 % of their time at the two
// barriers / the vmcnt(0) of a slice -- the register-staged slice is requested only one slice (0.86 us) ahead and every
// slice costs two barriers.  Here both operands' slices go straight to the LDS (global_load_lds_dwordx4, 2 rows x 512 B
// per wave instruction, no staging registers, no ds_write) into a ring of three stages: slice s+2 is issued while slice s
// is multiplied, one counted s_waitcnt vmcnt + ONE raw barrier per slice.  The LDS image is the plain [t][column] slice
// (the MFMA fragment reads are conflict-free on it as they are), the k order of every output element is unchanged
// (t ascending): bit-identical to the kernel above.  A last slice with rows >= T goes through registers, zeros past T.
typedef const __attribute__((address_space(1))) void* rp_gptr;
typedef __attribute__((address_space(3))) void* rp_lptr;
constexpr int kStages = 3;

__global__ __launch_bounds__(kThreads) void outer_accum_dma_kernel(int m, int n, int T, float alpha,
                                                                   const float* __restrict__ A, int64_t lda,
                                                                   const float* __restrict__ B, int64_t ldb,
                                                                   float* __restrict__ C, int64_t ldc, int accumulate) {
    __shared__ __attribute__((aligned(16))) float S[kStages * 2 * BK * BM];  // stage: A slice, then B slice
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = n / BN, tiles_m = m / BM;
    int ti, tj;
    if (((tiles_m | tiles_n) & 7) == 0 && (((tiles_m >> 3) * (tiles_n >> 3)) & 7) == 0) {  // XCD-aware order, as above
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        const int super = (l >> 6) * 8 + xcd, w = l & 63;
        const int super_n = tiles_n >> 3;
        ti = (super / super_n) * 8 + (w >> 3);
        tj = (super % super_n) * 8 + (w & 7);
    } else {
        ti = blockIdx.x / tiles_n;
        tj = blockIdx.x % tiles_n;
    }
    const int i0 = ti * BM, j0 = tj * BN;
    // wave w brings rows 4 w .. 4 w + 3 of each operand's slice: two DMA instructions of 2 rows x 128 columns each
    const int prow = lane >> 5, pcol = 4 * (lane & 31);
    const float* asrc = A + (int64_t)(4 * wave + prow) * lda + i0 + pcol;
    const float* bsrc = B + (int64_t)(4 * wave + prow) * ldb + j0 + pcol;
    const int slices = (T + BK - 1) / BK;
    auto issue = [&](int s) {
        float* stage = S + (s % kStages) * (2 * BK * BM);
        const int t0 = s * BK;
        if (t0 + BK > T) {  // the ragged last slice: through registers, zeros past T
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = 4 * wave + 2 * q + prow;
                rp_f4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
                if (t0 + row < T) {
                    va = rp_ld4(asrc + (int64_t)(t0 + 2 * q) * lda);
                    vb = rp_ld4(bsrc + (int64_t)(t0 + 2 * q) * ldb);
                }
                *reinterpret_cast<rp_f4*>(stage + row * BM + pcol) = va;
                *reinterpret_cast<rp_f4*>(stage + BK * BM + row * BN + pcol) = vb;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            return;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_global_load_lds((rp_gptr)(asrc + (int64_t)(t0 + 2 * q) * lda),
                                             (rp_lptr)(stage + (4 * wave + 2 * q) * BM), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((rp_gptr)(bsrc + (int64_t)(t0 + 2 * q) * ldb),
                                             (rp_lptr)(stage + BK * BM + (4 * wave + 2 * q) * BN), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int lk = lane >> 5, lr = lane & 31;
    issue(0);
    if (slices > 1) issue(1);
    for (int s = 0; s < slices; ++s) {
        // slice s has landed; the four DMA loads of slice s+1 may stay in flight -- IF slice s+1 came by DMA: a ragged
        // last slice goes through registers (a wave whose rows lie past T issues no load for it at all), so the count of
        // four would then cover slice s's own loads
        if (s + 1 < slices && (s + 2) * BK <= T) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everyone's pieces of slice s are in; everyone is done with slice s-1's stage
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < slices) issue(s + 2);
        const float* As = S + (s % kStages) * (2 * BK * BM);
        const float* Bs = As + BK * BM;
        float a0[BK / 2], a1[BK / 2], b0[BK / 2], b1[BK / 2];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float* ap = As + (kk + lk) * BM + wm * 64 + lr;
            const float* bp = Bs + (kk + lk) * BN + wn * 64 + lr;
            a0[kk / 2] = ap[0];
            a1[kk / 2] = ap[32];
            b0[kk / 2] = bp[0];
            b1[kk / 2] = bp[32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BK / 2; ++q) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b0[q], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b1[q], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b0[q], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b1[q], acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = j0 + wn * 64 + ni * 32 + lr;
            float old[16];  // (all 64 loads first was measured too: 256 VGPRs, two waves per SIMD, no gain)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                old[r] = accumulate ? C[(int64_t)row * ldc + col] : 0.f;  // (store form: 0 + x, the bits of accumulating into zeros)
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                C[(int64_t)row * ldc + col] = old[r] + alpha * acc[mi][ni][r];
            }
        }
}

#ifdef RP_MEASUREMENT
std::atomic<int> g_gemm_dma{1};  // measurement build only (rp_set_gemm_dma): 0 = always the register-staged kernel
#else
constexpr int g_gemm_dma = 1;
#endif

}  // namespace

namespace {

int outer_launch(int32_t m, int32_t n, int32_t T, float alpha, const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc, int accumulate, void* stream) {
    if (!A || !B || !C) RP_FAIL(RP_ERR_ARG, "A, B, C must be non-null");
    if (m <= 0 || n <= 0 || T < 0) RP_FAIL(RP_ERR_ARG, "m, n must be > 0 and T >= 0");
    if (lda < m || ldb < n || ldc < n) RP_FAIL(RP_ERR_SHAPE, "leading dimensions too small");
    if (T == 0) {
        if (!accumulate) RP_HIP_CHECK(hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)n * sizeof(float), (size_t)m,
                                                       static_cast<hipStream_t>(stream)));
        return RP_OK;
    }
    const int tiles_m = (m + BM - 1) / BM, tiles_n = (n + BN - 1) / BN;
    const bool vec4 = (m % 4 == 0) && (n % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                      !(reinterpret_cast<uintptr_t>(A) & 15u) && !(reinterpret_cast<uintptr_t>(B) & 15u);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec4 && g_gemm_dma && m % BM == 0 && n % BN == 0 && T >= 1)
        hipLaunchKernelGGL(outer_accum_dma_kernel, dim3(tiles_m * tiles_n), dim3(kThreads), 0, st, m, n, T, alpha, A, lda, B,
                           ldb, C, ldc, accumulate);
    else if (vec4)
        hipLaunchKernelGGL(outer_accum_kernel<true>, dim3(tiles_m * tiles_n), dim3(kThreads), 0, st, m, n, T, alpha, A,
                           lda, B, ldb, C, ldc, accumulate);
    else
        hipLaunchKernelGGL(outer_accum_kernel<false>, dim3(tiles_m * tiles_n), dim3(kThreads), 0, st, m, n, T, alpha, A,
                           lda, B, ldb, C, ldc, accumulate);
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

}  // namespace

extern "C" int rp_outer_accum_rect(int32_t m, int32_t n, int32_t T, float alpha, const float* A, int64_t lda,
                                   const float* B, int64_t ldb, float* C, int64_t ldc, void* stream) {
    return outer_launch(m, n, T, alpha, A, lda, B, ldb, C, ldc, 1, stream);
}

extern "C" int rp_outer_store_rect(int32_t m, int32_t n, int32_t T, float alpha, const float* A, int64_t lda,
                                   const float* B, int64_t ldb, float* C, int64_t ldc, void* stream) {
    return outer_launch(m, n, T, alpha, A, lda, B, ldb, C, ldc, 0, stream);
}

#ifdef RP_MEASUREMENT
extern "C" int rp_set_gemm_dma(int on) {
    g_gemm_dma = on;
    return RP_OK;
}
#endif

extern "C" int rp_outer_accum(int32_t n, int32_t T, float alpha, const float* A, int64_t lda, const float* B,
                              int64_t ldb, float* C, int64_t ldc, void* stream) {
    return rp_outer_accum_rect(n, n, T, alpha, A, lda, B, ldb, C, ldc, stream);
}
