// rp_dot.h -- streaming row dot products shared by the forward step kernel and the transposed-copy adjoint.
// The summation tree per row is fixed (DESIGN.md "Summation order") and mirrored by oracle/rnn_oracle.c.
#pragma once

#include "rp_common.h"

// Start chunk of the 8-row group g = (global row) >> 3 on the 16-byte path: m0 = (kStagger * g) mod M.
// Without it all waves march through addresses that are equal modulo the row stride (a power of two at the
// benchmark sizes) in lock step; rotating the start is worth +3 % of HBM bandwidth.  Part of the summation spec.
constexpr unsigned kStagger = 3;

struct RowDotArgs {
    const float* W;    // [rows x ldw]
    int64_t ldw;
    int n, rows, row0;  // columns, local rows, global index of local row 0
    const float* src;  // [n]
};

template <bool NT>
__device__ __forceinline__ rp_f4 ld_w(const float* p) {
    if constexpr (NT)
        return rp_ld4_stream(p);
    else
        return rp_ld4(p);
}

// 16-byte path: requires n % 4 == 0, ldw % 4 == 0, 16-byte aligned W and src.
template <int R, int U, bool NT>
__device__ __forceinline__ void rp_dot_rows_v4(const RowDotArgs& P, int row_base, int lane, float (&sum)[R]) {
    const float* wrow[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        int rr = row_base + j;
        rr = rr < P.rows ? rr : P.rows - 1;  // clamp: tail rows re-read a valid row, result discarded
        wrow[j] = P.W + (int64_t)rr * P.ldw + 4 * lane;
    }
    const float* sp = P.src + 4 * lane;
    rp_f4 a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = rp_f4{0.f, 0.f, 0.f, 0.f};

    const int M = P.n >> 8;  // full 256-column chunks
    auto sweep = [&](int mA, int mB) {  // chunks [mA, mB) in ascending order
        int m = mA;
        for (; m + U <= mB; m += U) {
            rp_f4 rv[U];
            rp_f4 w[U][R];
#pragma unroll
            for (int u = 0; u < U; ++u) rv[u] = rp_ld4(sp + (int64_t)(m + u) * 256);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < R; ++j) w[u][j] = ld_w<NT>(wrow[j] + (int64_t)(m + u) * 256);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    a[j].x = fmaf(w[u][j].x, rv[u].x, a[j].x);
                    a[j].y = fmaf(w[u][j].y, rv[u].y, a[j].y);
                    a[j].z = fmaf(w[u][j].z, rv[u].z, a[j].z);
                    a[j].w = fmaf(w[u][j].w, rv[u].w, a[j].w);
                }
        }
        for (; m < mB; ++m) {
            rp_f4 rv = rp_ld4(sp + (int64_t)m * 256);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                rp_f4 w = ld_w<NT>(wrow[j] + (int64_t)m * 256);
                a[j].x = fmaf(w.x, rv.x, a[j].x);
                a[j].y = fmaf(w.y, rv.y, a[j].y);
                a[j].z = fmaf(w.z, rv.z, a[j].z);
                a[j].w = fmaf(w.w, rv.w, a[j].w);
            }
        }
    };
    const int m0 = M > 0 ? (int)(((unsigned)((P.row0 + row_base) >> 3) * kStagger) % (unsigned)M) : 0;
    sweep(m0, M);
    sweep(0, m0);
    if ((M << 8) + 4 * lane < P.n) {  // partial last chunk (n % 4 == 0, so all 4 columns are valid)
        rp_f4 rv = rp_ld4(sp + (int64_t)M * 256);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            rp_f4 w = ld_w<NT>(wrow[j] + (int64_t)M * 256);
            a[j].x = fmaf(w.x, rv.x, a[j].x);
            a[j].y = fmaf(w.y, rv.y, a[j].y);
            a[j].z = fmaf(w.z, rv.z, a[j].z);
            a[j].w = fmaf(w.w, rv.w, a[j].w);
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) sum[j] = rp_wave_allreduce(rp_lane_sum4(a[j]));
}

// The same dot products for SHORT rows (a few thousand columns: the launch-bound sizes of the shared-launch step of
// rp_forward_coupled).  With few chunks per row the two ascending sweeps of rp_dot_rows_v4 degenerate -- their
// remainders are loaded one chunk at a time, each a full memory round trip: 8 chunks from start chunk 3 = one batch of 4
// and four single loads.  Here the chunk sequence m0, m0 + 1, ..., wrapping, is walked as ONE sequence in batches of U
// (the last batch predicated), so a 2048-column row is one batch of 8 loads.  The accumulators see the same products in
// the same order: identical bits.
template <int R, int U, bool NT>
__device__ __forceinline__ void rp_dot_rows_v4_ring(const RowDotArgs& P, int row_base, int lane, float (&sum)[R]) {
    const float* wrow[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        int rr = row_base + j;
        rr = rr < P.rows ? rr : P.rows - 1;
        wrow[j] = P.W + (int64_t)rr * P.ldw + 4 * lane;
    }
    const float* sp = P.src + 4 * lane;
    rp_f4 a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = rp_f4{0.f, 0.f, 0.f, 0.f};
    const int M = P.n >> 8;
    const int m0 = M > 0 ? (int)(((unsigned)((P.row0 + row_base) >> 3) * kStagger) % (unsigned)M) : 0;
    for (int i = 0; i < M; i += U) {
        rp_f4 rv[U];
        rp_f4 w[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u < M) {  // (wave-uniform)
                int m = m0 + i + u;
                m = m >= M ? m - M : m;
                rv[u] = rp_ld4(sp + (int64_t)m * 256);
#pragma unroll
                for (int j = 0; j < R; ++j) w[u][j] = ld_w<NT>(wrow[j] + (int64_t)m * 256);
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u < M) {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    a[j].x = fmaf(w[u][j].x, rv[u].x, a[j].x);
                    a[j].y = fmaf(w[u][j].y, rv[u].y, a[j].y);
                    a[j].z = fmaf(w[u][j].z, rv[u].z, a[j].z);
                    a[j].w = fmaf(w[u][j].w, rv[u].w, a[j].w);
                }
            }
    }
    if ((M << 8) + 4 * lane < P.n) {  // partial last chunk (n % 4 == 0, so all 4 columns are valid)
        rp_f4 rv = rp_ld4(sp + (int64_t)M * 256);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            rp_f4 w = ld_w<NT>(wrow[j] + (int64_t)M * 256);
            a[j].x = fmaf(w.x, rv.x, a[j].x);
            a[j].y = fmaf(w.y, rv.y, a[j].y);
            a[j].z = fmaf(w.z, rv.z, a[j].z);
            a[j].w = fmaf(w.w, rv.w, a[j].w);
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) sum[j] = rp_wave_allreduce(rp_lane_sum4(a[j]));
}

// 4-byte path for sizes / alignments the 16-byte path cannot take (small test networks).
template <int R>
__device__ __forceinline__ void rp_dot_rows_s1(const RowDotArgs& P, int row_base, int lane, float (&sum)[R]) {
    float a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = 0.f;
    for (int col = lane; col < P.n; col += RP_WAVE) {
        float rv = P.src[col];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            int rr = row_base + j;
            rr = rr < P.rows ? rr : P.rows - 1;
            a[j] = fmaf(P.W[(int64_t)rr * P.ldw + col], rv, a[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) sum[j] = rp_wave_allreduce(a[j]);
}

// ---- packed sparse rows (rectipy_amd/sparse.py) ------------------------------------------------------------------
// Shared by the sparse step kernel and the sparse adjoint (packed W^T).  A lane walks its entry list (value + code =
// component << 14 | column chunk, four entries per lane per group) and feeds the accumulator the code names; the source
// vector is gathered per entry from LDS (component-major layout, conflict-free) or from L2.
constexpr int kSparseThreads = 1024;
constexpr int kSparseLdsMaxN = 32768;

// Cooperative copy of src[0..n) into LDS: inside every 64-column block the element at offset o goes to
// (o + ((column >> 6) & 3)) & 63, so that the 64 lanes of a gather (lane l asks for column 256 m_l + 4 l + c) hit 64
// different banks whatever their chunks m_l.
// Columns in [n_valid, n_total) are filled with zeros: padding entries (value 0, code 0) gather from chunk 0 of the staged
// range and must meet a finite number there.
template <int THREADS = kSparseThreads>
__device__ __forceinline__ void rp_sparse_stage_source(float* lds_src, const float* __restrict__ src, int n_valid,
                                                       int n_total) {
    for (int c4 = threadIdx.x * 4; c4 < n_total; c4 += THREADS * 4) {
        const rp_f4 v = c4 < n_valid ? rp_ld4(src + c4) : rp_f4{0.f, 0.f, 0.f, 0.f};
        // component-major inside every 256-column chunk: column 256 m + 4 l + c at 256 m + 64 c + l.  A gather of the 64
        // lanes -- each asking for its own (m, c) -- then hits bank l mod 32 in either half of the wave (ds_read_b32 is
        // served in two groups of 32 lanes over 32 banks): conflict-free whatever the entries ask for.  (r01-r04 rotated the
        // components inside 64-column blocks, which is conflict-free over 64 banks only: the counters showed half of the
        // LDS cycles of the sparse step as bank conflicts, profiles/r05_sparse_pmc.json.)
        const int pos = (c4 & ~255) + ((c4 & 255) >> 2);
        lds_src[pos] = v.x;
        lds_src[pos + 64] = v.y;
        lds_src[pos + 128] = v.z;
        lds_src[pos + 192] = v.w;
    }
    __syncthreads();
}

template <bool LDS_SRC>
__device__ __forceinline__ void sparse_entry(float v, unsigned code, const float* xs, int lane_base, int lane_rot,
                                             rp_f4& a) {
    float a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w;
    const int c = (int)(code >> 14), m = (int)(code & 0x3fffu);
    // LDS: position of column 256 m + 4 l + c in the component-major layout (lane_base = l); global: the column itself
    // (lane_base = 4 l)
    (void)lane_rot;
    const int idx = LDS_SRC ? (m << 8) + (c << 6) + lane_base : (m << 8) + lane_base + c;
    const float x = xs[idx];
    const float acc = c == 0 ? a0 : (c == 1 ? a1 : (c == 2 ? a2 : a3));
    const float r = fmaf(v, x, acc);
    a.x = c == 0 ? r : a0;
    a.y = c == 1 ? r : a1;
    a.z = c == 2 ? r : a2;
    a.w = c == 3 ? r : a3;
}

// ---- blocked form (n <= 32768; sparse.py: pack_rows_blocked) -----------------------------------------------------
// RB consecutive rows share one padded stream: a lane's list is the concatenation of its lists of the block's rows, the
// last entry of every row segment carries RP_SPARSE_END.  Lanes move from row to row at their own pace: at END a lane
// parks the lane sum (a0+a1)+(a2+a3) of the finished row in its own LDS slot stash[row][lane] and clears its
// accumulators; when the block's stream is exhausted every lane has parked RB sums and the rows' butterflies run.  Only
// the block is padded to its longest lane list (1.08x the ideal bytes at RB = 16, p = 0.1, n = 32768; per-row: 1.35x).
// A lane only ever reads back its own slots, so no barrier is needed.  `epi(row, sum)` is called by one lane per row.
// `stage()` brings the gathered vector into the LDS (it ends with the workgroup's barrier); it is called once per wave, AFTER
// the wave has requested the first groups of its first block, so that the stream's first round trip overlaps the staging.
template <int WAVES, int U, typename Stage, typename Epi>
__device__ __forceinline__ void rp_sparse_blocked_rows(const float* __restrict__ vals, const uint16_t* __restrict__ code,
                                                       const int32_t* __restrict__ blkoff, int rows, int RB,
                                                       const float* lds_src, float* lds_stash, Stage stage, Epi epi) {
    bool staged = false;
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int lane_base = lane;  // (position of the lane's columns inside a staged chunk: rp_sparse_stage_source)
    const int lane_rot = 0;
    float* const stash = lds_stash + (size_t)wave * RB * RP_WAVE + lane;
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    const int nblocks = (rows + RB - 1) / RB;
    for (int b = blockIdx.x * WAVES + wave; b < nblocks; b += gridDim.x * WAVES) {
        const int g0 = blkoff[b], len = blkoff[b + 1] - g0;
        const float* vp = vals + (int64_t)g0 * 256 + 4 * lane;
        const uint16_t* ip = code + (int64_t)g0 * 256 + 4 * lane;
        rp_f4 acc = {0.f, 0.f, 0.f, 0.f};
        float* st = stash;
        auto entry = [&](float v, unsigned cd) {
            sparse_entry<true>(v, cd & ~(unsigned)RP_SPARSE_END, lds_src, lane_base, lane_rot, acc);
            if (cd & RP_SPARSE_END) {
                *st = rp_lane_sum4(acc);
                st += RP_WAVE;
                acc = rp_f4{0.f, 0.f, 0.f, 0.f};
            }
        };
        // two register sets in turn: the next U groups are requested before the current ones are worked through (r05: the
        // counters showed the waves 64 % of their time at s_waitcnt with the vector ALUs 36 % and the LDS 20 % busy -- a wave
        // that requests, waits, then works leaves its stream idle while it works; same entries in the same order)
        rp_f4 va[U], vb[U];
        us4 ma[U], mb[U];
        auto fetch = [&](int k, rp_f4 (&v)[U], us4 (&m)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + u < len) {
                    v[u] = rp_ld4_stream(vp + (int64_t)(k + u) * 256);
                    m[u] = __builtin_nontemporal_load(reinterpret_cast<const us4*>(ip + (int64_t)(k + u) * 256));
                }
        };
        auto work = [&](int k, const rp_f4 (&v)[U], const us4 (&m)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + u < len) {
                    entry(v[u].x, m[u].x);
                    entry(v[u].y, m[u].y);
                    entry(v[u].z, m[u].z);
                    entry(v[u].w, m[u].w);
                }
        };
        fetch(0, va, ma);
        if (!staged) {
            stage();
            staged = true;
        }
        for (int k = 0; k < len; k += 2 * U) {
            fetch(k + U, vb, mb);
            work(k, va, ma);
            fetch(k + 2 * U, va, ma);
            work(k + U, vb, mb);
        }
        const int nr = min(RB, rows - b * RB);
        // the rows' butterflies at once (rp_rows_allreduce: the same pairings, i.e. the same additions per row; 8 rows: 10 cross-lane
        // steps instead of 48); lanes 0, 64 / RB, ... end up with the total of row rp_rows_allreduce_row<RB>(lane)
        if (RB == 16) {
            float part[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r] = r < nr ? stash[r * RP_WAVE] : 0.f;
            const float total = rp_rows_allreduce<16>(part, lane);
            const int f = rp_rows_allreduce_row<16>(lane);
            if ((lane & 3) == 0 && f < nr) epi(b * RB + f, total);
        } else if (RB == 8) {
            float part[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) part[r] = r < nr ? stash[r * RP_WAVE] : 0.f;
            const float total = rp_rows_allreduce<8>(part, lane);
            const int f = rp_rows_allreduce_row<8>(lane);
            if ((lane & 7) == 0 && f < nr) epi(b * RB + f, total);
        } else {
            float mine = 0.f;
            for (int r = 0; r < nr; ++r) {
                const float sum = rp_wave_allreduce(stash[r * RP_WAVE]);
                if (lane == r) mine = sum;
            }
            if (lane < nr) epi(b * RB + lane, mine);
        }
    }
    if (!staged) stage();  // (a wave without a block still takes part in the staging and its barrier)
}

// the same with the vector staged by the caller beforehand
template <int WAVES, int U, typename Epi>
__device__ __forceinline__ void rp_sparse_blocked_rows(const float* __restrict__ vals, const uint16_t* __restrict__ code,
                                                       const int32_t* __restrict__ blkoff, int rows, int RB,
                                                       const float* lds_src, float* lds_stash, Epi epi) {
    rp_sparse_blocked_rows<WAVES, U>(vals, code, blkoff, rows, RB, lds_src, lds_stash, [] {}, epi);
}

// ---- windowed form (n > 32768; sparse.py: pack_rows_windowed) ----------------------------------------------------
// The gathered vector is staged in LDS one 32768-column window at a time.  A workgroup takes a batch of 64 slots (rows
// whose sweeps start in the same window; 4 per wave) and walks the phases of that class: stage the window of the phase,
// let every wave run its rows' entries of that phase (codes hold window-relative chunks), next phase.  The accumulators
// stay in registers across the phases, so every one of them still sees its nonzeros in the dense order.  `epi(row, sum)`
// is called by one lane per finished row.
constexpr int kWindowChunks = 128;
constexpr int kBatchRows = 64;

struct SparseWindowed {
    const float* vals;
    const uint16_t* code;
    const int32_t* rowoff;        // [slots * phases + 1]
    const int32_t* row_ids;       // [batches * 64], -1 = padding
    const int32_t* batch_class;   // [batches]
    const int32_t* phase_window;  // [windows * phases], -1 = no such phase
    int phases, batches;
};

template <int U, typename Epi>
__device__ __forceinline__ void rp_sparse_windowed_rows(const SparseWindowed& S, const float* __restrict__ src, int n,
                                                        float* lds_src, Epi epi) {
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    constexpr int RB = kBatchRows / (kSparseThreads / RP_WAVE);  // rows per wave
    const int lane_base = lane;  // (position of the lane's columns inside a staged chunk: rp_sparse_stage_source)
    const int lane_rot = 0;
    const int NP = S.phases;
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    for (int batch = blockIdx.x; batch < S.batches; batch += gridDim.x) {
        const int cls = S.batch_class[batch];
        int slot[RB], row[RB];
        rp_f4 acc[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            slot[j] = batch * kBatchRows + wave * RB + j;
            row[j] = S.row_ids[slot[j]];
            acc[j] = rp_f4{0.f, 0.f, 0.f, 0.f};
        }
        int staged = -1;
        for (int ph = 0; ph < NP; ++ph) {
            const int w = S.phase_window[cls * NP + ph];
            if (w < 0) continue;  // block-uniform
            if (w != staged) {
                __syncthreads();  // every wave is done with the previous window
                const int col0 = w * kWindowChunks * 256;
                const int cols = min(kWindowChunks * 256, n - col0);
                rp_sparse_stage_source(lds_src, src + col0, cols, min(kWindowChunks * 256, (cols + 255) & ~255));
                staged = w;
            }
            const float* vp[RB];
            const uint16_t* ip[RB];
            int len[RB], nmax = 0;
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int g0 = S.rowoff[slot[j] * NP + ph];
                len[j] = S.rowoff[slot[j] * NP + ph + 1] - g0;
                vp[j] = S.vals + (int64_t)g0 * 256 + 4 * lane;
                ip[j] = S.code + (int64_t)g0 * 256 + 4 * lane;
                nmax = len[j] > nmax ? len[j] : nmax;
            }
            for (int k = 0; k < nmax; k += U) {
                rp_f4 v[RB][U];
                us4 m[RB][U];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < RB; ++j)
                        if (k + u < len[j]) {
                            v[j][u] = rp_ld4_stream(vp[j] + (int64_t)(k + u) * 256);
                            m[j][u] = __builtin_nontemporal_load(reinterpret_cast<const us4*>(ip[j] + (int64_t)(k + u) * 256));
                        }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < RB; ++j)
                        if (k + u < len[j]) {
                            sparse_entry<true>(v[j][u].x, m[j][u].x, lds_src, lane_base, lane_rot, acc[j]);
                            sparse_entry<true>(v[j][u].y, m[j][u].y, lds_src, lane_base, lane_rot, acc[j]);
                            sparse_entry<true>(v[j][u].z, m[j][u].z, lds_src, lane_base, lane_rot, acc[j]);
                            sparse_entry<true>(v[j][u].w, m[j][u].w, lds_src, lane_base, lane_rot, acc[j]);
                        }
            }
        }
        float mine = 0.f;
        int my_row = -1;
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const float sum = rp_wave_allreduce(rp_lane_sum4(acc[j]));
            if (lane == j) {
                mine = sum;
                my_row = row[j];
            }
        }
        if (lane < RB && my_row >= 0) epi(my_row, mine);
    }
}
