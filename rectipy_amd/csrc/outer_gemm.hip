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
                                                               float* __restrict__ C, int64_t ldc) {
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
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float* ap = As + (kk + lk) * BM + wm * 64 + lr;
            const float* bp = Bs + (kk + lk) * BN + wn * 64 + lr;
            const float a0 = ap[0], a1 = ap[32];
            const float b0 = bp[0], b1 = bp[32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = j0 + wn * 64 + ni * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < m && col < n) {
                    float* cp = C + (int64_t)row * ldc + col;
                    *cp = *cp + alpha * acc[mi][ni][r];
                }
            }
        }
}

}  // namespace

extern "C" int rp_outer_accum_rect(int32_t m, int32_t n, int32_t T, float alpha, const float* A, int64_t lda,
                                   const float* B, int64_t ldb, float* C, int64_t ldc, void* stream) {
    if (!A || !B || !C) RP_FAIL(RP_ERR_ARG, "A, B, C must be non-null");
    if (m <= 0 || n <= 0 || T < 0) RP_FAIL(RP_ERR_ARG, "m, n must be > 0 and T >= 0");
    if (lda < m || ldb < n || ldc < n) RP_FAIL(RP_ERR_SHAPE, "leading dimensions too small");
    if (T == 0) return RP_OK;
    const int tiles_m = (m + BM - 1) / BM, tiles_n = (n + BN - 1) / BN;
    const bool vec4 = (m % 4 == 0) && (n % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                      !(reinterpret_cast<uintptr_t>(A) & 15u) && !(reinterpret_cast<uintptr_t>(B) & 15u);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec4)
        hipLaunchKernelGGL(outer_accum_kernel<true>, dim3(tiles_m * tiles_n), dim3(kThreads), 0, st, m, n, T, alpha, A,
                           lda, B, ldb, C, ldc);
    else
        hipLaunchKernelGGL(outer_accum_kernel<false>, dim3(tiles_m * tiles_n), dim3(kThreads), 0, st, m, n, T, alpha, A,
                           lda, B, ldb, C, ldc);
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

extern "C" int rp_outer_accum(int32_t n, int32_t T, float alpha, const float* A, int64_t lda, const float* B,
                              int64_t ldb, float* C, int64_t ldc, void* stream) {
    return rp_outer_accum_rect(n, n, T, alpha, A, lda, B, ldb, C, ldc, stream);
}
