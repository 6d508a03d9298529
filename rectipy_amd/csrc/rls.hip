// rls.hip -- recursive-least-squares readout training on gfx950.
//
// Replaces RLS.update (reference rectipy/edges.py:227-234) driven step by step from
// Network._rls (rectipy/network.py:1093-1121):
//     z = beta^-1 P x;  k = 1 / (1 + x.z);  W += outer(k (y - W x), z);  P -= k z z^T;  loss = |y - y_hat|^2
// (the reference's weight expression `y - k x (W + y z^T)^T` equals k (y - W x) algebraically).
//
// The state-error correlation matrix P is n_in x n_in (4.3 GB at n_in = 32768): like the recurrent step this is
// an HBM-bound streaming problem.  The reference reads P once for z and reads + writes it once for the rank-1
// update: 12 n^2 bytes per update.  Because the readout does not feed back into the population, the inputs
// x_t of a whole window are known up front, so the rank-1 update of step t is FUSED with the mat-vec of step
// t+1: one pass reads P, applies P -= k z z^T, writes it back and accumulates z' = beta^-1 P_new x_{t+1}
// -- 8 n^2 bytes per update, the minimum for a read-modify-write of P.
//
// Row mapping and summation order as in the step kernel (one wave per R rows, lane l owns columns 256 m + 4 l
// .. +3, m ascending from 0, lane sum (a0+a1)+(a2+a3), fixed butterfly), mirrored by oracle/rls_oracle.c.

#include <atomic>

#include "rp_common.h"
#include "rp_resident.h"

namespace {

constexpr int kThreads = 256;

struct RlsParams {
    float* P;
    int64_t ldp;
    int n_in, n_out;
    float* W;
    int64_t ldw;
    float beta_inv;
    float* z;          // current z
    float* zn;         // next z (output of the fused pass)
    float* scal;       // [0] = k, [1] = loss
    const float* x;    // x of the current update step
    const float* y;    // target of the current update step
    const float* x_next;
    float* pred;       // prediction row of the current update step (pre-update weights)
    float* err;        // [n_out] workspace
    float* loss_out;   // loss of this update, replicated for the steps it stays valid for
    int loss_count;
    int no_dot;  // benchmark-only: skip the mat-vec half of the pass
    float* partials;  // tiled pass: [n_in x tiles_per_row] partial dot products
    float* colparts;  // symmetric pass: [ceil(n_in / 64) x ldc] column partials of the off-diagonal blocks
    int64_t ldc;
    // blocked symmetric sweep (rp_rls.block): the block before's rank-1 updates and the inputs of this block
    const float* zp;     // [npend x n_in] z vectors of the updates to apply
    const float* kp;     // [npend] their k
    int npend;
    const float* xb;     // first input row of the block; row j at xb + j * xb_stride
    int64_t xb_stride;
    int nrhs;
    int64_t part_stride, col_stride;  // floats between the partials / column partials of consecutive inputs
    float* yb;           // [K x n_in] Y_j = beta^-1 P x_j
    double* cm;          // [K x K] c_ij = z_i . x_j
    float* kk;           // [K] k of the block's updates
    int j, K;            // the update of the block the small kernels work on
    float* z0;           // where the z vector of the block's first update goes (= Y_0)
    // the block's scalars from its Gram matrix (rls_blk_gram_kernel / rls_blk_apply_kernel)
    double* gm;          // [K x K] G_lj = Y_l . x_j
    double* am;          // [K x K] A: z_j = sum_l A_jl Y_l
    float* hm;           // [n_out x K] H_oj = W_o . x_j with the weights the block starts from
    float* ke;           // [K x n_out] k_j err_j[o]
    int cnt;             // updates of this block
    const float* y0;     // targets of the block's first update; update j at y0 + j * y_stride
    int64_t y_stride;
    float* pred0;        // prediction row of the block's first update (or null); update j at pred0 + j * pred_stride
    int64_t pred_stride;
    float* loss0;        // loss of update j replicated into loss0[j * loss_stride ... + loss_count_j)
    int loss_stride, loss_tail;  // entries per update; entries of the block's LAST update
    float* zp_out;       // [K x n_in] receives the block's z vectors (the next pass applies them)
};


// eight operand pairs of a thread's chain, j = j0 + 1024 q: a full batch is loaded without predicates -- behind a bounds
// branch per element the compiler waits for every load before it issues the next one (16 round trips instead of one)
__device__ __forceinline__ void rls_load8(const float* u, const float* x, int j0, int n, float (&uv)[8], float (&xv)[8]) {
    if (j0 + 7 * 1024 < n) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            uv[q] = u[j0 + q * 1024];
            xv[q] = x[j0 + q * 1024];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = j0 + q * 1024;
            uv[q] = j < n ? u[j] : 0.f;
            xv[q] = j < n ? x[j] : 0.f;
        }
    }
}

// d = x.z, k = 1/(1+d), y_hat = W x, err = y - y_hat, loss = err.err.  One block per dot product (block 0: x.z, block
// o + 1: W_o.x), each in the fixed order "per-thread fmaf chain over j = tid + 1024 m, then a 1024-leaf tree"; the block that
// finishes last (a counter in scal[3]) turns the 1 + n_out sums into k, err, loss in output order.  A thread's operands are
// loaded eight chain links ahead of the chain itself (the chain is serial, its loads need not be).
__global__ __launch_bounds__(1024) void rls_scalars_kernel(const RlsParams P) {
    __shared__ float red[1024];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float* u = b == 0 ? P.z : P.W + (int64_t)(b - 1) * P.ldw;
    float a = 0.f;
    for (int j0 = tid; j0 < P.n_in; j0 += 8 * 1024) {
        float uv[8], xv[8];
        rls_load8(u, P.x, j0, P.n_in, uv, xv);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + q * 1024 < P.n_in) a = fmaf(uv[q], xv[q], a);
    }
    red[tid] = a;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if (tid < off) red[tid] = red[tid] + red[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        if (b == 0) P.scal[2] = red[0];
        else P.err[b - 1] = red[0];  // y_hat for now
        __threadfence();
        is_last = atomicAdd(reinterpret_cast<unsigned*>(P.scal + 3), 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    if (tid == 0) {
        __threadfence();
        const float d = __hip_atomic_load(P.scal + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float k = 1.0f / (1.0f + d);
        float loss = 0.f;
        for (int o = 0; o < P.n_out; ++o) {
            const float yhat = __hip_atomic_load(P.err + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float e = P.y[o] - yhat;
            loss = fmaf(e, e, loss);
            P.err[o] = e;
            if (P.pred) P.pred[o] = yhat;
        }
        P.scal[0] = k;
        P.scal[1] = loss;
        *reinterpret_cast<unsigned*>(P.scal + 3) = 0u;  // ready for the next update
        red[0] = loss;
    }
    __syncthreads();
    if (P.loss_out && tid < P.loss_count) P.loss_out[tid] = red[0];
}

// W[o][j] += (k * err[o]) * z[j]  (its own launch: folded into the last block of the kernel above it costs 30 us -- one
// block's worth of memory parallelism -- instead of 5)
__global__ void rls_weights_kernel(const RlsParams P) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.n_in) return;
    const float k = P.scal[0];
    const float zj = P.z[j];
    for (int o = 0; o < P.n_out; ++o) {
        float* w = P.W + (int64_t)o * P.ldw + j;
        *w = *w + (k * P.err[o]) * zj;
    }
}

// pred[t][o] = W[o] . X[t]  for `count` rows (one wave per (t, o) pair, butterfly reduce)
__global__ void rls_predict_kernel(const float* W, int64_t ldw, int n_in, int n_out, const float* X, int64_t ldx,
                                   float* pred, int64_t ldpred, int count) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (item >= count * n_out) return;
    const int t = item / n_out, o = item % n_out;
    const float* w = W + (int64_t)o * ldw;
    const float* x = X + (int64_t)t * ldx;
    float a = 0.f;
    for (int j = lane; j < n_in; j += 64) a = fmaf(w[j], x[j], a);
    a = rp_wave_allreduce(a);
    if (lane == 0) pred[(int64_t)t * ldpred + o] = a;
}

// Fused pass over P: (UPDATE) P[i][:] -= (k z_i) z[:], then zn_i = beta_inv * (P_new[i][:] . x_next)
template <int R, int U, bool NTS, bool VEC4, bool UPDATE, bool NTL = true>
__global__ __launch_bounds__(kThreads) void rls_pass_kernel(const RlsParams P) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int groups = (P.n_in + R - 1) / R;
    const float k = UPDATE ? P.scal[0] : 0.f;
    for (int g = blockIdx.x * (kThreads / 64) + wave; g < groups; g += gridDim.x * (kThreads / 64)) {
        const int row_base = g * R;
        float* prow[R];
        float kz[R];
        float sum[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            int rr = row_base + j;
            rr = rr < P.n_in ? rr : P.n_in - 1;
            prow[j] = P.P + (int64_t)rr * P.ldp;
            kz[j] = UPDATE ? k * P.z[rr] : 0.f;
        }
        const bool dot = P.x_next != nullptr && !P.no_dot;
        if constexpr (VEC4) {
            rp_f4 a[R];
#pragma unroll
            for (int j = 0; j < R; ++j) a[j] = rp_f4{0.f, 0.f, 0.f, 0.f};
            // U column chunks in flight per row: U * R 16-byte loads per lane
            const rp_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
            auto body = [&](int col, rp_f4 (&p)[R], rp_f4 zv, rp_f4 xv) {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if constexpr (UPDATE) {
                        p[j].x = p[j].x - kz[j] * zv.x;
                        p[j].y = p[j].y - kz[j] * zv.y;
                        p[j].z = p[j].z - kz[j] * zv.z;
                        p[j].w = p[j].w - kz[j] * zv.w;
                        if (row_base + j < P.n_in) {
                            if constexpr (NTS)
                                __builtin_nontemporal_store(p[j], reinterpret_cast<rp_f4*>(prow[j] + col));
                            else
                                *reinterpret_cast<rp_f4*>(prow[j] + col) = p[j];
                        }
                    }
                    a[j].x = fmaf(p[j].x, xv.x, a[j].x);
                    a[j].y = fmaf(p[j].y, xv.y, a[j].y);
                    a[j].z = fmaf(p[j].z, xv.z, a[j].z);
                    a[j].w = fmaf(p[j].w, xv.w, a[j].w);
                }
            };
            int col = 4 * lane;
            for (; col + 256 * (U - 1) < P.n_in; col += 256 * U) {
                rp_f4 p[U][R], zv[U], xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    zv[u] = UPDATE ? rp_ld4(P.z + col + 256 * u) : zero4;
                    xv[u] = dot ? rp_ld4(P.x_next + col + 256 * u) : zero4;
#pragma unroll
                    for (int j = 0; j < R; ++j)
                        p[u][j] = NTL ? rp_ld4_stream(prow[j] + col + 256 * u) : rp_ld4(prow[j] + col + 256 * u);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) body(col + 256 * u, p[u], zv[u], xv[u]);
            }
            for (; col < P.n_in; col += 256) {
                rp_f4 p[R];
                const rp_f4 zv = UPDATE ? rp_ld4(P.z + col) : zero4;
                const rp_f4 xv = dot ? rp_ld4(P.x_next + col) : zero4;
#pragma unroll
                for (int j = 0; j < R; ++j) p[j] = NTL ? rp_ld4_stream(prow[j] + col) : rp_ld4(prow[j] + col);
                body(col, p, zv, xv);
            }
#pragma unroll
            for (int j = 0; j < R; ++j) sum[j] = rp_wave_allreduce(rp_lane_sum4(a[j]));
        } else {
            float a[R];
#pragma unroll
            for (int j = 0; j < R; ++j) a[j] = 0.f;
            for (int col = lane; col < P.n_in; col += 64) {
                const float zv = UPDATE ? P.z[col] : 0.f;
                const float xv = dot ? P.x_next[col] : 0.f;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    float p = prow[j][col];
                    if constexpr (UPDATE) {
                        p = p - kz[j] * zv;
                        if (row_base + j < P.n_in) prow[j][col] = p;
                    }
                    a[j] = fmaf(p, xv, a[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < R; ++j) sum[j] = rp_wave_allreduce(a[j]);
        }
        if (dot) {
            float mine = 0.f;
#pragma unroll
            for (int j = 0; j < R; ++j)
                if (lane == j) mine = sum[j];
            if (lane < R && row_base + lane < P.n_in) P.zn[row_base + lane] = P.beta_inv * mine;
        }
    }
}

// ---- small n_in, an update every step: the whole sweep in ONE launch (r05; the machinery: rp_resident.h) ------------------------
// The dense sweep of rp_rls_sweep at launch-bound sizes costs three dependent launches per update (scalars, weights, the fused
// pass over P): 11 us per update whatever n_in.  Here every wave keeps its rows of P in registers for the whole sweep, and a full
// copy of W (n_out <= 4 rows) and of z; per update the waves exchange ONE vector, z = beta^-1 P x, as tagged granules.
// Everything else each wave forms for itself, in the orders of the launched kernels: d = x . z and y_hat_o = W_o . x as the
// scalar kernel's 1024-leaf tree (leaf j = fmaf(u_j, x_j, 0); levels 512 and 256 add a lane's own chunks, levels 128 .. 4 are the
// wave butterfly, levels 2 and 1 a lane's components (x + z) + (y + w)), k, the errors, the loss; W_o += (k err_o) z; then its
// rows of P: p -= (k z_i) z_j and the row dots with the next input (chunks ascending, lane sum, butterfly).  Bit-identical.
struct RlsResidentArgs {
    rp_resident_ctl C;
    int updates, n_in, n_out;
    const float* X;
    int64_t ldx;
    const float* Y;
    int64_t ldy;
    float* pred;
    int64_t ldpred;
    float* loss;
    float* P;
    int64_t ldp;
    float* W;
    int64_t ldw;
    float beta_inv;
};

// the lane that holds row j of a wave after rp_rows_allreduce<RW> (the inverse of rp_rows_allreduce_row)
template <int RW>
__host__ __device__ constexpr int rls_lane_of_row(int j) {
    int lane = 0, width = RW, off = 32;
    for (; width > 1; width >>= 1, off >>= 1)
        if (j & (width / 2)) lane |= off;
    return lane;
}

// L: the register layout of the dense pass's row-dot path (rp_resident.h: rp_layout_v4 on the 16-byte path, rp_layout_s1 else)
template <class L>
__global__ __launch_bounds__(kThreads) void resident_rls_kernel(const RlsResidentArgs A) {
    static_assert(kThreads == 64 * kResidentWaves, "rp_resident.h counts four waves per workgroup");
    constexpr int NO = 4;   // rows of W a wave carries (n_out <= 4)
    constexpr int RW = L::RW;
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int block = rp_resident_join(A.C);
    if (block < 0) return;
    const int gw = block * kResidentWaves + (threadIdx.x >> 6);
    const int row_w0 = gw * RW;
    const int n = A.n_in;
    typename L::Rows p;
    L::load_rows(p, A.P, A.ldp, row_w0, lane, n);
    typename L::Vec wv[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        if (o < A.n_out) L::load_vec(wv[o], A.W + (int64_t)o * A.ldw, lane, n);
        else L::zero_vec(wv[o]);
    }
    const int my_row = row_w0 + rp_rows_allreduce_row<RW>(lane);
    const bool epi_lane = (lane & (64 / RW - 1)) == 0 && my_row < n;
    typename L::Vec x, z;
    float zi[RW];   // z of the wave's own rows (uniform)
    // z = beta^-1 P x: the row dots of the pass kernel -- chunks ascending, no stagger --, lane sum, butterfly
    auto dots = [&](int q) -> bool {   // publishes exchange q (tag q + 1, parity q & 1) and gathers z; false: give up
        const float zt = A.beta_inv * L::dots_plain(p, x, lane);
#pragma unroll
        for (int j = 0; j < RW; ++j) zi[j] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(zt), rls_lane_of_row<RW>(j)));
        if (epi_lane) rp_resident_publish(A.C, q & 1, my_row, (unsigned)(q + 1), zt);
        return L::sweep(A.C, q & 1, (unsigned)(q + 1), z, lane, n);
    };
    L::load_vec(x, A.X, lane, n);
    if (!dots(0)) return;
    for (int t = 0; t < A.updates; ++t) {
        // scalars: d = x . z, k, y_hat = W x, err, loss
        const float d = L::tree_dot(z, x);
        const float k = 1.0f / (1.0f + d);
        float err[NO], loss = 0.f;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            err[o] = 0.f;
            if (o < A.n_out) {   // (uniform)
                const float yhat = L::tree_dot(wv[o], x);
                const float e = A.Y[(int64_t)t * A.ldy + o] - yhat;
                loss = fmaf(e, e, loss);
                err[o] = e;
                if (gw == 0 && lane == 0 && A.pred) A.pred[(int64_t)t * A.ldpred + o] = yhat;
            }
        }
        if (gw == 0 && lane == 0 && A.loss) A.loss[t] = loss;
        // weights: W_o += (k err_o) z
#pragma unroll
        for (int o = 0; o < NO; ++o)
            if (o < A.n_out) L::add_scaled(wv[o], k * err[o], z);
        // the pass: p -= (k z_i) z_j, then the row dots with the next input
#pragma unroll
        for (int j = 0; j < RW; ++j) L::row_sub_scaled(p, j, k * zi[j], z);
        if (t + 1 < A.updates) {
            L::load_vec(x, A.X + (int64_t)(t + 1) * A.ldx, lane, n);
            if (!dots(t + 1)) return;
        }
    }
    // the state back to memory: every wave its rows of P, the first wave W
    L::store_rows(p, A.P, A.ldp, row_w0, lane, n);
    if (gw == 0) {
#pragma unroll
        for (int o = 0; o < NO; ++o)
            if (o < A.n_out) L::store_vec(wv[o], A.W + (int64_t)o * A.ldw, lane, n);
    }
}

int launch_resident_rls(const rp_rls* r, int32_t steps, const float* X, int64_t ldx, const float* Y, int64_t ldy, float* pred,
                        int64_t ldpred, float* loss, bool v4, hipStream_t st) {
    RlsResidentArgs A = {};
    A.updates = steps;
    A.n_in = r->n_in;
    A.n_out = r->n_out;
    A.X = X;
    A.ldx = ldx;
    A.Y = Y;
    A.ldy = ldy;
    A.pred = pred;
    A.ldpred = ldpred;
    A.loss = loss;
    A.P = r->P;
    A.ldp = r->ldp;
    A.W = r->W;
    A.ldw = r->ldw;
    A.beta_inv = r->beta_inv;
    void* ws = nullptr;
    if (rp_resident_begin(r->n_in, st, &A.C, &ws, v4) != RP_OK) {
        (void)hipGetLastError();
        return kResidentUnavailable;
    }
    const dim3 grid(rp_resident_grid(A.C));
#define RP_LAUNCH_(L_) hipLaunchKernelGGL((resident_rls_kernel<RP_UNPAREN L_>), grid, dim3(kThreads), 0, st, A)
    RP_RESIDENT_DISPATCH_LAYOUT(r->n_in, v4, RP_LAUNCH_)
#undef RP_LAUNCH_
    return rp_resident_end(A.C, ws, st, "the resident RLS sweep");
}

// Tiled pass (rp_rls.partials given).  How the chip walks P decides the speed of an in-place update, not its arithmetic
// (scripts/probes/rmw_probe.hip, 4 GiB read-modify-write on one MI355X): one wave per row keeps 8192 far-apart row
// streams open -- 5.1 TB/s; ONE-SHOT workgroups that each take a contiguous piece in address order, all loads first,
// then all stores -- 6.2 TB/s with 16 KiB per workgroup, 6.7 TB/s with 4 KiB (one 16-byte access per lane: less in
// flight per CU, as for the step kernel's walk).  Here a workgroup (4 waves) takes the tile (row i, columns
// [1024 c, 1024 c + 1024)): thread (wave w, lane l) owns columns 1024 c + 256 w + 4 l .. +3.  The tile's share of the
// mat-vec: the four products of a lane, lane sum (a0+a1)+(a2+a3), butterfly; the four wave sums are added in wave order;
// rls_zn_kernel adds a row's tile partials in ascending order (mirrored by oracle/rls_oracle.c).  Measured at
// n_in = 32768 (scripts/archive/sweep_rls_tiles.py): 4096-column tiles 1.54 ms per update, 2048 1.56, 1024 1.41.
// (TU = 2, 4: wider tiles, thread owns columns 1024 u + 256 w + 4 l of the tile, accumulated over u ascending --
// benchmark-only, rp_set_rls_tile; a different summation order.)
constexpr int kTileCols = 1024;

template <bool NTS, bool UPDATE, int TU>
__global__ __launch_bounds__(kThreads) void rls_tile_kernel(const RlsParams P) {
    __shared__ float wsum[kThreads / 64];
    extern __shared__ float residency_pad[];  // (never touched: a dynamic allocation caps the workgroups per CU)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = P.n_in;
    constexpr int kCols = 1024 * TU;
    const int tpr = (n + kCols - 1) / kCols;
    const int row = blockIdx.x / tpr;
    const int ct = blockIdx.x - row * tpr;
    const int col0 = ct * kCols + 256 * wave + 4 * lane;
    float* prow = P.P + (int64_t)row * P.ldp;
    const bool dot = P.x_next != nullptr && !P.no_dot;
    const float kz = UPDATE ? P.scal[0] * P.z[row] : 0.f;
    const rp_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    rp_f4 p[TU], zv[TU], xv[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
        const int col = col0 + 1024 * u;
        if (col < n) {
            p[u] = rp_ld4_stream(prow + col);
            zv[u] = UPDATE ? rp_ld4(P.z + col) : zero4;
            xv[u] = dot ? rp_ld4(P.x_next + col) : zero4;
        }
    }
    rp_f4 a = zero4;
#pragma unroll
    for (int u = 0; u < TU; ++u) {
        const int col = col0 + 1024 * u;
        if (col < n) {
            if constexpr (UPDATE) {
                p[u].x = p[u].x - kz * zv[u].x;
                p[u].y = p[u].y - kz * zv[u].y;
                p[u].z = p[u].z - kz * zv[u].z;
                p[u].w = p[u].w - kz * zv[u].w;
                if constexpr (NTS)
                    __builtin_nontemporal_store(p[u], reinterpret_cast<rp_f4*>(prow + col));
                else
                    *reinterpret_cast<rp_f4*>(prow + col) = p[u];
            }
            a.x = fmaf(p[u].x, xv[u].x, a.x);
            a.y = fmaf(p[u].y, xv[u].y, a.y);
            a.z = fmaf(p[u].z, xv[u].z, a.z);
            a.w = fmaf(p[u].w, xv[u].w, a.w);
        }
    }
    if (dot) {
        const float s = rp_wave_allreduce(rp_lane_sum4(a));
        if (lane == 0) wsum[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) P.partials[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
    }
}

// z'_i = beta_inv * (part[i][0] + part[i][1] + ...), ascending
__global__ void rls_zn_kernel(const RlsParams P, int tile_cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_in) return;
    const int tpr = (P.n_in + tile_cols - 1) / tile_cols;
    const float* part = P.partials + (int64_t)i * tpr;
    float s = part[0];
    for (int c = 1; c < tpr; ++c) s = s + part[c];
    P.zn[i] = P.beta_inv * s;
}

// ---- symmetric pass (rp_rls.colparts given) -----------------------------------------------------------------------------
// P -= k z z^T keeps P symmetric, so only the blocks on and above the diagonal of a 1024 x 1024 block grid are streamed:
// 4 n^2 instead of 8 n^2 bytes per update (plus the diagonal blocks, 1 / nb of P, which are kept whole).  A ONE-SHOT
// workgroup takes the tile (rows [R c, R c + R), columns [1024 ct, 1024 ct + 1024)) with ct >= I = (R c) >> 10 -- all its
// loads first, as the tiled pass (the walk that reaches the chip's read-modify-write rate); thread (wave w, lane l) owns
// columns 1024 ct + 256 w + 4 l .. +3 of all R rows.  Per row: P_ij -= (k z_i) z_j, store, and the tile's share of the
// mat-vec z' = beta^-1 P x in BOTH roles of the element:
//   row role     rowpart[i][ct] = tile partial of sum_j P_ij x_j: products, lane sum (a0+a1)+(a2+a3), the butterfly over
//                the 64 lanes (done for the R rows together by a transposed butterfly: R - 1 + log2(64 / R) shuffles
//                instead of 6 R, the same additions), the four wave sums added in wave order -- the tiled pass's partial
//   column role  (ct > I only) colpart[c][j] = fmaf chain over the tile's rows i ascending of P_ij x_i, from 0
// rls_sym_colsum_kernel: colsum[I][j] = colpart chain over the chunks c of block row I, ascending.
// rls_sym_zn_kernel: z'_i = beta^-1 (colsum[0][i] + ... + colsum[I_i - 1][i], then rowpart[i][I_i] + ... ascending).
// After the last update of a sweep rls_mirror_kernel copies the off-diagonal blocks to their transposed positions, so the
// caller always sees a full P.  Mirrored by oracle/rls_oracle.c.
// C = rows per tile (the chunk of the column partials: part of the summation order), R = rows in flight per wave (how the
// tile is walked: groups of R rows, the next group's loads issued before the current one is processed; never the order).
// FULL: n_in is a multiple of 1024, so every tile is complete and no access is predicated -- the loads of the next group
// are then plain straight-line code and the compiler waits with a counted vmcnt for the current group only (behind the
// per-access bounds branches it waits for vmcnt(0), i.e. for the prefetch it has just issued).
// PROBE (measurements only, wrong results): 1 = no column chains, 2 = no row sums, 3 = neither (the bare walk)
template <bool UPDATE, int C, int R, bool FULL, int PROBE = 0>
__global__ __launch_bounds__(kThreads) void rls_sym_kernel(const RlsParams P) {
    __shared__ float wsum[C][4];
    __shared__ float zrow[C], xrow[C];
    static_assert(C % R == 0 && (R & (R - 1)) == 0 && R <= 32, "groups of R rows");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = P.n_in;
    const int nb = (n + 1023) >> 10;
    // blockIdx.x enumerates the block pairs (I, ct >= I) row by row -- only tiles that exist are launched --, blockIdx.y
    // the chunk inside block row I
    int b = blockIdx.x, I = 0;
    while (b >= nb - I) {
        b -= nb - I;
        ++I;
    }
    const int ct = I + b;
    const int chunk = I * (1024 / C) + blockIdx.y;
    const int row0 = chunk * C;
    if (!FULL && row0 >= n) return;
    const bool upper = ct > I;
    const int col = ct * 1024 + 256 * wave + 4 * lane;
    const bool valid = FULL || col < n;
    const bool dot = P.x_next != nullptr;
    const rp_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float* base = P.P + (int64_t)row0 * P.ldp + col;
    rp_f4 bufA[R], bufB[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        bufA[r] = (FULL || (valid && row0 + r < n)) ? rp_ld4_stream(base + (int64_t)r * P.ldp) : zero4;
    const rp_f4 zv = (UPDATE && valid) ? rp_ld4(P.z + col) : zero4;
    const rp_f4 xv = (dot && valid) ? rp_ld4(P.x_next + col) : zero4;
    if (threadIdx.x < C) {
        const int i = row0 + threadIdx.x;
        zrow[threadIdx.x] = (UPDATE && (FULL || i < n)) ? P.z[i] : 0.f;
        xrow[threadIdx.x] = (dot && (FULL || i < n)) ? P.x_next[i] : 0.f;
    }
    const float k = UPDATE ? P.scal[0] : 0.f;
    __syncthreads();
    rp_f4 cp = zero4;
    constexpr int kGroup = 64 / R;  // lanes that hold the same row's total after the transposed butterfly
    // one group of R rows: update + store, the lane sums of the row partials, the column chains
    auto group = [&](const int g, const rp_f4 (&cur)[R]) {
        float rs[R], zr[R], xr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {  // (adjacent LDS words: the compiler merges them into wide reads, one wait per group)
            zr[r] = UPDATE ? zrow[g + r] : 0.f;
            xr[r] = xrow[g + r];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = row0 + g + r;
            rp_f4 p = cur[r];
            if constexpr (UPDATE) {
                const float kz = k * zr[r];
                p.x = p.x - kz * zv.x;
                p.y = p.y - kz * zv.y;
                p.z = p.z - kz * zv.z;
                p.w = p.w - kz * zv.w;
                if (FULL || (valid && i < n))
                    __builtin_nontemporal_store(p, reinterpret_cast<rp_f4*>(base + (int64_t)(g + r) * P.ldp));
            }
            const rp_f4 a = rp_fma4(p, xv, zero4);
            rs[r] = rp_lane_sum4(a);
            // (rows past the end must not touch the chain: the oracle stops at n.  On a diagonal block the chain is
            // computed and dropped -- 32 of 528 blocks -- rather than branched around: no divergent path in the loop)
            if (!(PROBE & 1) && (FULL || (upper && i < n))) {
                const float xi = xr[r];
                cp = rp_fma4s(p, xi, cp);
            }
        }
        if (PROBE & 2) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) t += rs[r];
            if (t == 12345.f) wsum[g][wave] = t;  // (keeps the loads alive)
        } else if (dot) {
            const float total = rp_rows_allreduce<R>(rs, lane);
            if ((lane & (kGroup - 1)) == 0) {  // the row index of a lane's total is spelled by its lane bits, high bit first
                int r = 0;
                int width = R, off = 32;
                for (; width > 1; width >>= 1, off >>= 1) r += (lane & off) ? width / 2 : 0;
                wsum[g + r][wave] = total;
            }
        }
    };
    // Two register buffers A / B used in turn, the loop unrolled by two groups so that neither is ever copied: the loads
    // of group g + 1 are issued, unconditionally, before group g is processed, and the wait in front of a group is a
    // counted one (the R loads just issued -- and the stores of the group before -- may stay in flight).  The last pair
    // is peeled (nothing to prefetch behind it).
    auto fetch = [&](const int g, rp_f4 (&buf)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            buf[r] = (FULL || (valid && row0 + g + r < n)) ? rp_ld4_stream(base + (int64_t)(g + r) * P.ldp) : zero4;
    };
    static_assert(C % (2 * R) == 0, "pairs of groups");
#pragma unroll 1
    for (int g = 0; g + 2 * R < C; g += 2 * R) {
        fetch(g + R, bufB);
        group(g, bufA);
        fetch(g + 2 * R, bufA);
        group(g + R, bufB);
    }
    fetch(C - R, bufB);
    group(C - 2 * R, bufA);
    group(C - R, bufB);
    if (!dot) return;
    __syncthreads();
    if (threadIdx.x < C && (FULL || row0 + (int)threadIdx.x < n)) {
        const int r = threadIdx.x;
        P.partials[(int64_t)(row0 + r) * nb + ct] = ((wsum[r][0] + wsum[r][1]) + wsum[r][2]) + wsum[r][3];
    }
    if (upper && valid) *reinterpret_cast<rp_f4*>(P.colparts + (int64_t)chunk * P.ldc + col) = cp;
}

// z'_i = beta^-1 (colsum[0][i] + ... + colsum[I_i - 1][i], then rowpart[i][I_i] + ... ascending), colsum[b][i] = the
// column partials colpart[c][i] of the chunks c of block row b added in ascending order from the first one.  One
// workgroup per 32 columns: thread (c, s) adds up the block rows b = s, s + 32, ... of column c (their sums are
// independent), thread (c, 0) chains them and the row partials in the order above.
template <int COLS>  // columns per workgroup: a wave reads COLS * 4 contiguous bytes of a row of column partials
__global__ __launch_bounds__(1024) void rls_sym_zn_kernel(const RlsParams P0, int rows_per_chunk) {
    extern __shared__ float colsum[];  // [nb][COLS]
    constexpr int SLOTS = 1024 / COLS;
    RlsParams P = P0;  // blocked sweep: blockIdx.y = input j of the block (its slices of the partials, its row of Y)
    P.partials += (int64_t)blockIdx.y * P0.part_stride;
    P.colparts += (int64_t)blockIdx.y * P0.col_stride;
    P.zn += (int64_t)blockIdx.y * P0.n_in;
    const int c = threadIdx.x % COLS, slot = threadIdx.x / COLS;
    const int i = blockIdx.x * COLS + c;
    const int nb = (P.n_in + 1023) >> 10;
    const int I = (blockIdx.x * COLS) >> 10;  // (COLS divides 1024: one block row per workgroup)
    const int per_block = 1024 / rows_per_chunk;
    if (i < P.n_in) {
        for (int b = slot; b < I; b += SLOTS) {
            const float* col = P.colparts + (int64_t)b * per_block * P.ldc + i;
            // eight independent loads in flight, added in ascending order (r05: the first batch starts at chunk 0 -- with 8 chunks per
            // block row, 128-row tiles, the old "first, then batches of eight" took the one-load-at-a-time tail for all of them)
            float s = 0.f;
            int q = 0;
            for (; q + 8 <= per_block; q += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(q + u) * P.ldc];
                s = q == 0 ? v[0] : s + v[0];
#pragma unroll
                for (int u = 1; u < 8; ++u) s = s + v[u];
            }
            for (; q < per_block; ++q) s = q == 0 ? col[0] : s + col[(int64_t)q * P.ldc];
            colsum[b * COLS + c] = s;
        }
    }
    __syncthreads();
    if (slot != 0 || i >= P.n_in) return;
    float s = 0.f;
    bool first = true;
    for (int b = 0; b < I; ++b) {
        const float v = colsum[b * COLS + c];
        s = first ? v : s + v;
        first = false;
    }
    const float* part = P.partials + (int64_t)i * nb;
    for (int ct = I; ct < nb; ++ct) {
        s = first ? part[ct] : s + part[ct];
        first = false;
    }
    P.zn[i] = P.beta_inv * s;
    if (P0.z0 && blockIdx.y == 0) P0.z0[i] = P.beta_inv * s;  // blocked sweep: z_0 = Y_0
}

// P[j][i] = P[i][j] for one 64 x 64 tile of an off-diagonal block (I < J), through LDS: 256-byte row pieces on both
// sides (16 lanes x 16 bytes), four of them per thread in flight.  blockIdx.x = the block pair (only pairs that exist
// are launched, enumerated row by row), blockIdx.y = the tile inside the 1024 x 1024 block.
__global__ __launch_bounds__(kThreads) void rls_mirror_kernel(float* Pm, int64_t ldp, int n) {
    __shared__ float tile[64][65];
    const int nb = (n + 1023) >> 10;
    int b = blockIdx.x, I = 0;
    while (b >= nb - 1 - I) {
        b -= nb - 1 - I;
        ++I;
    }
    const int J = I + 1 + b;
    const int i0 = I * 1024 + (blockIdx.y >> 4) * 64, j0 = J * 1024 + (blockIdx.y & 15) * 64;
    if (j0 >= n) return;  // (i0 < j0: a source row always exists where its column does)
    const int r = threadIdx.x >> 4, q = threadIdx.x & 15;  // 16 rows x 16 quads per pass, four passes
    rp_f4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int j = j0 + 4 * q;
        v[p] = j < n ? rp_ld4_stream(Pm + (int64_t)(i0 + 16 * p + r) * ldp + j) : rp_f4{0.f, 0.f, 0.f, 0.f};  // (n % 4 == 0)
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        tile[16 * p + r][4 * q + 0] = v[p].x;
        tile[16 * p + r][4 * q + 1] = v[p].y;
        tile[16 * p + r][4 * q + 2] = v[p].z;
        tile[16 * p + r][4 * q + 3] = v[p].w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int jj = j0 + 16 * p + r;  // row jj of the mirrored tile, columns i0 + 4 q .. + 3
        if (jj < n) {
            rp_f4 w;
            w.x = tile[4 * q + 0][16 * p + r];
            w.y = tile[4 * q + 1][16 * p + r];
            w.z = tile[4 * q + 2][16 * p + r];
            w.w = tile[4 * q + 3][16 * p + r];
            __builtin_nontemporal_store(w, reinterpret_cast<rp_f4*>(Pm + (int64_t)jj * ldp + i0 + 4 * q));
        }
    }
}


// ---- blocked symmetric sweep (rp_rls.block = K) ------------------------------------------------------------------------
// The symmetric pass with K rank-1 updates and K inputs at once: per element p -= (k_0 z_0i) z_0j, then -= (k_1 z_1i) z_1j,
// ... (update order: the bits of K separate passes), store, and for every input x_j of the block the row partial and
// the column chain exactly as rls_sym_kernel forms them for its one input -- into slice j of the partials / column
// partials.  Updates / inputs beyond npend / nrhs are zeros (p - 0 * 0 = p; their partials are never read).
// The K x R lane sums of a group of R rows go through ONE transposed butterfly (K R values: the same additions per row).
template <int K, bool UPDATE, bool DOT, int C, int R, bool FULL>
__global__ __launch_bounds__(kThreads) void rls_symk_kernel(const RlsParams P) {
    __shared__ float wsum[DOT ? K : 1][C][4];
    __shared__ float kzrow[C][UPDATE ? K : 1], xrow[C][DOT ? K : 1];  // (a row's K values side by side: wide LDS reads)
    static_assert(C % (2 * R) == 0 && (R & (R - 1)) == 0 && K * R <= 64, "groups of R rows, K R values per butterfly");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = P.n_in;
    const int nb = (n + 1023) >> 10;
    int b = blockIdx.x, I = 0;
    while (b >= nb - I) {
        b -= nb - I;
        ++I;
    }
    const int ct = I + b;
    const int chunk = I * (1024 / C) + blockIdx.y;
    const int row0 = chunk * C;
    if (!FULL && row0 >= n) return;
    const bool upper = ct > I;
    const int col = ct * 1024 + 256 * wave + 4 * lane;
    const bool valid = FULL || col < n;
    const rp_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float* base = P.P + (int64_t)row0 * P.ldp + col;
    rp_f4 bufA[R], bufB[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        bufA[r] = (FULL || (valid && row0 + r < n)) ? rp_ld4_stream(base + (int64_t)r * P.ldp) : zero4;
    rp_f4 zv[UPDATE ? K : 1], xv[DOT ? K : 1];
    if constexpr (UPDATE) {
#pragma unroll
        for (int k = 0; k < K; ++k) zv[k] = (valid && k < P.npend) ? rp_ld4(P.zp + (int64_t)k * n + col) : zero4;
    }
    if constexpr (DOT) {
#pragma unroll
        for (int k = 0; k < K; ++k) xv[k] = (valid && k < P.nrhs) ? rp_ld4(P.xb + (int64_t)k * P.xb_stride + col) : zero4;
    }
    for (int e = threadIdx.x; e < K * C; e += kThreads) {
        const int k = e / C, r = e - k * C;
        const int i = row0 + r;
        const bool in = FULL || i < n;
        if constexpr (UPDATE) kzrow[r][k] = (in && k < P.npend) ? P.kp[k] * P.zp[(int64_t)k * n + i] : 0.f;
        if constexpr (DOT) xrow[r][k] = (in && k < P.nrhs) ? P.xb[(int64_t)k * P.xb_stride + i] : 0.f;
    }
    __syncthreads();
    rp_f4 cp[DOT ? K : 1];
    if constexpr (DOT) {
#pragma unroll
        for (int k = 0; k < K; ++k) cp[k] = zero4;
    }
    constexpr int kGroup = 64 / (K * R);  // lanes that hold the same (input, row) total after the transposed butterfly
    auto group = [&](const int g, const rp_f4 (&cur)[R]) {
        float rs[DOT ? K * R : 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = row0 + g + r;
            rp_f4 p = cur[r];
            if constexpr (UPDATE) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float kz = kzrow[g + r][k];
                    p.x = p.x - kz * zv[k].x;
                    p.y = p.y - kz * zv[k].y;
                    p.z = p.z - kz * zv[k].z;
                    p.w = p.w - kz * zv[k].w;
                }
                if (FULL || (valid && i < n))
                    __builtin_nontemporal_store(p, reinterpret_cast<rp_f4*>(base + (int64_t)(g + r) * P.ldp));
            }
            if constexpr (DOT) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const rp_f4 a = rp_fma4(p, xv[k], zero4);
                    rs[k * R + r] = rp_lane_sum4(a);
                    if (FULL || (upper && i < n)) {  // (on a diagonal block the chain is computed and dropped, as in rls_sym_kernel)
                        cp[k] = rp_fma4s(p, xrow[g + r][k], cp[k]);
                    }
                }
            }
        }
        if constexpr (DOT) {
            const float total = rp_rows_allreduce<K * R>(rs, lane);
            if ((lane & (kGroup - 1)) == 0) {
                const int f = rp_rows_allreduce_row<K * R>(lane);
                wsum[f / R][g + f % R][wave] = total;
            }
        }
    };
    auto fetch = [&](const int g, rp_f4 (&buf)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            buf[r] = (FULL || (valid && row0 + g + r < n)) ? rp_ld4_stream(base + (int64_t)(g + r) * P.ldp) : zero4;
    };
    // (r05: a ring of four register buffers -- three groups of rows in flight -- was measured: 0.150 against 0.127 ms per update;
    // the extra registers cost more than the latency they cover)
#pragma unroll 1
    for (int g = 0; g + 2 * R < C; g += 2 * R) {
        fetch(g + R, bufB);
        group(g, bufA);
        fetch(g + 2 * R, bufA);
        group(g + R, bufB);
    }
    fetch(C - R, bufB);
    group(C - 2 * R, bufA);
    group(C - R, bufB);
    if constexpr (DOT) {
        __syncthreads();
        for (int e = threadIdx.x; e < K * C; e += kThreads) {
            const int k = e / C, r = e - k * C;
            if (k < P.nrhs && (FULL || row0 + r < n))
                P.partials[(int64_t)k * P.part_stride + (int64_t)(row0 + r) * nb + ct] =
                    ((wsum[k][r][0] + wsum[k][r][1]) + wsum[k][r][2]) + wsum[k][r][3];
        }
        if (upper && valid) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (k < P.nrhs)
                    *reinterpret_cast<rp_f4*>(P.colparts + (int64_t)k * P.col_stride + (int64_t)chunk * P.ldc + col) = cp[k];
        }
    }
}

// W[o][q] += (k err[o]) z_j[q] as rls_weights_kernel, and -- `next` -- the z vector of the block's NEXT update, whose
// coefficients are complete now (k_j from this update's scalar kernel, c_{i, j+1} from the updates before):
// z_{j+1}[q] = (float)(Y_{j+1}[q] - sum_{i<=j} g_i z_i[q]) in double: acc = Y[q]; acc = fma(-g_i, z_i[q], acc), i ascending;
// g_i = beta^-1 (k_i c_{i, j+1})
__global__ __launch_bounds__(256) void rls_blk_weights_kernel(const RlsParams P, int next) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= P.n_in) return;
    const float k = P.scal[0];
    const float zq = P.z[q];
    for (int o = 0; o < P.n_out; ++o) {
        float* w = P.W + (int64_t)o * P.ldw + q;
        *w = *w + (k * P.err[o]) * zq;
    }
    if (!next) return;
    const int jn = P.j + 1;
    double acc = (double)P.yb[(int64_t)jn * P.n_in + q];
    for (int i = 0; i < jn; ++i) {
        const double g = (double)P.beta_inv * ((double)(i == P.j ? k : P.kk[i]) * P.cm[i * P.K + jn]);
        acc = fma(-g, (double)P.z[(int64_t)(i - P.j) * P.n_in + q], acc);
    }
    P.z[(int64_t)P.n_in + q] = (float)acc;
}

// The scalar kernel of the blocked sweep.  Block 0: d = z_j . x_j, blocks 1 + jn: c_{j, j + 1 + jn} = z_j . x_{j+1+jn} -- both
// with DOUBLE accumulators in the order of rls_scalars_kernel -- , the last n_out blocks: y_hat_o = W_o . x_j as there.  The
// block that finishes last forms k_j = 1 / (1 + d) (rounded once), err, loss.
__global__ __launch_bounds__(1024) void rls_blk_scalars_kernel(const RlsParams P, int extra) {
    __shared__ double red64[1024];
    __shared__ float red[1024];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    if (b <= extra) {
        const float* x = b == 0 ? P.x : P.x + (int64_t)b * P.xb_stride;  // (x_j ... and the later inputs of the block)
        double a = 0.0;
        for (int j0 = tid; j0 < P.n_in; j0 += 8 * 1024) {
            float uv[8], xv[8];
            rls_load8(P.z, x, j0, P.n_in, uv, xv);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j0 + q * 1024 < P.n_in) a = fma((double)uv[q], (double)xv[q], a);
        }
        red64[tid] = a;
        __syncthreads();
        for (int off = 512; off >= 1; off >>= 1) {
            if (tid < off) red64[tid] = red64[tid] + red64[tid + off];
            __syncthreads();
        }
        if (tid == 0) P.cm[P.j * P.K + P.j + b] = red64[0];  // (the diagonal entry holds d)
    } else {
        const int o = b - extra - 1;
        const float* u = P.W + (int64_t)o * P.ldw;
        float a = 0.f;
        for (int j0 = tid; j0 < P.n_in; j0 += 8 * 1024) {
            float uv[8], xv[8];
            rls_load8(u, P.x, j0, P.n_in, uv, xv);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j0 + q * 1024 < P.n_in) a = fmaf(uv[q], xv[q], a);
        }
        red[tid] = a;
        __syncthreads();
        for (int off = 512; off >= 1; off >>= 1) {
            if (tid < off) red[tid] = red[tid] + red[tid + off];
            __syncthreads();
        }
        if (tid == 0) P.err[o] = red[0];  // y_hat for now
    }
    if (tid == 0) {
        __threadfence();
        is_last = atomicAdd(reinterpret_cast<unsigned*>(P.scal + 3), 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    if (tid == 0) {
        __threadfence();
        const double d = __hip_atomic_load(P.cm + P.j * P.K + P.j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float k = (float)(1.0 / (1.0 + d));
        float loss = 0.f;
        for (int o = 0; o < P.n_out; ++o) {
            const float yhat = __hip_atomic_load(P.err + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float e = P.y[o] - yhat;
            loss = fmaf(e, e, loss);
            P.err[o] = e;
            if (P.pred) P.pred[o] = yhat;
        }
        P.scal[0] = k;
        P.scal[1] = loss;
        P.kk[P.j] = k;
        *reinterpret_cast<unsigned*>(P.scal + 3) = 0u;
        red[0] = loss;
    }
    __syncthreads();
    if (P.loss_out && tid < P.loss_count) P.loss_out[tid] = red[0];
}

// ---- r05: the scalars of a whole block from ONE launch ---------------------------------------------------------------------
// Inside a block every z_j is a combination of the block's Y vectors, z_j = sum_{l <= j} A_jl Y_l (A_jj = 1), so every dot
// product the recurrence needs follows from the Gram matrix G_lj = Y_l . x_j:  c_ij = z_i . x_j = sum_l A_il G_lj.  One launch
// forms G (cnt^2 dot products in double, each in the order of rls_blk_scalars_kernel: per-thread chain over q = tid + 1024 m,
// then a 1024-leaf tree) and H_oj = W_o . x_j (the predictions' part that comes from the weights the block starts from, float
// as there); the block that finishes last runs the recurrence on K x K doubles:
//   for j: A_j = e_j; for i < j: c_ij = sum_{l<=i} A_il G_lj (l ascending from 0, fma); g = beta^-1 (k_i c_ij); A_j -= g A_i (fma)
//          d_j = c_jj likewise; k_j = (float)(1 / (1 + d_j))
//          yhat_j[o] = (float)(H_oj + sum_{i<j} (k_i err_i[o]) c_ij) (double, i ascending); err, loss as always
// and rls_blk_apply_kernel then forms all z_j (double: acc = Y_j[q]; acc = fma(A_jl, Y_l[q], acc), l ascending < j) and applies
// the cnt weight updates one after the other per element.  Two launches per BLOCK instead of two per update; the products are
// the mathematically identical ones, with every cancellation (z_j is a small difference of large vectors) carried in double.
__global__ __launch_bounds__(1024) void rls_blk_gram_kernel(const RlsParams P) {
    __shared__ double red64[1024];
    __shared__ float red[1024];
    __shared__ int is_last;
    __shared__ double G[8][8], A[8][8], c[8];  // (run-time indexed: LDS, not registers)
    const int tid = threadIdx.x, b = blockIdx.x, cnt = P.cnt, K = P.K;
    if (b < cnt * cnt) {
        const int l = b / cnt, j = b - l * cnt;
        const float* yl = P.yb + (int64_t)l * P.n_in;
        const float* x = P.xb + (int64_t)j * P.xb_stride;
        double a = 0.0;
        for (int j0 = tid; j0 < P.n_in; j0 += 8 * 1024) {
            float uv[8], xv[8];
            rls_load8(yl, x, j0, P.n_in, uv, xv);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j0 + q * 1024 < P.n_in) a = fma((double)uv[q], (double)xv[q], a);
        }
        red64[tid] = a;
        __syncthreads();
        for (int off = 512; off >= 1; off >>= 1) {
            if (tid < off) red64[tid] = red64[tid] + red64[tid + off];
            __syncthreads();
        }
        if (tid == 0) P.gm[l * K + j] = red64[0];
    } else {
        const int e = b - cnt * cnt, o = e / cnt, j = e - o * cnt;
        const float* u = P.W + (int64_t)o * P.ldw;
        const float* x = P.xb + (int64_t)j * P.xb_stride;
        float a = 0.f;
        for (int j0 = tid; j0 < P.n_in; j0 += 8 * 1024) {
            float uv[8], xv[8];
            rls_load8(u, x, j0, P.n_in, uv, xv);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j0 + q * 1024 < P.n_in) a = fmaf(uv[q], xv[q], a);
        }
        red[tid] = a;
        __syncthreads();
        for (int off = 512; off >= 1; off >>= 1) {
            if (tid < off) red[tid] = red[tid] + red[tid + off];
            __syncthreads();
        }
        if (tid == 0) P.hm[o * K + j] = red[0];
    }
    if (tid == 0) {
        __threadfence();
        is_last = atomicAdd(reinterpret_cast<unsigned*>(P.scal + 3), 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    // The recurrence, on LDS operands, by a handful of threads in lock step (every chain in the order stated above: the work
    // is split over the independent chains -- c_ij over i, A_jl over l, yhat_j[o] over o --, never inside one).  One thread
    // walking it through global memory took 50 us: ~300 dependent round trips.
    __shared__ float Hs[64][8], Ys[8][64], KE[8][64], ks[8], ls[8], yh_s[8][64];
    const int m = P.n_out;
    __threadfence();
    if (tid < cnt * cnt) G[tid / cnt][tid % cnt] = __hip_atomic_load(P.gm + (tid / cnt) * K + tid % cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int e = tid; e < m * cnt; e += 1024) {
        const int o = e / cnt, j = e - o * cnt;
        if (o < 64) {
            Hs[o][j] = __hip_atomic_load(P.hm + o * K + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            Ys[j][o] = P.y0[(int64_t)j * P.y_stride + o];
        }
    }
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
        if (tid < j) {  // c_ij, i = tid
            double cij = 0.0;
            for (int l = 0; l <= tid; ++l) cij = fma(A[tid][l], G[l][j], cij);
            c[tid] = cij;
        }
        __syncthreads();
        if (tid < cnt) {  // A_jl, l = tid
            double a = tid == j ? 1.0 : 0.0;
            for (int i = 0; i < j; ++i)
                if (tid <= i) a = fma(-((double)P.beta_inv * ((double)ks[i] * c[i])), A[i][tid], a);
            A[j][tid] = a;
        }
        __syncthreads();
        if (tid == 0) {
            double d = 0.0;
            for (int l = 0; l <= j; ++l) d = fma(A[j][l], G[l][j], d);
            ks[j] = (float)(1.0 / (1.0 + d));
        }
        __syncthreads();
        if (tid < m && tid < 64) {  // yhat_j[o], o = tid
            double yh = (double)Hs[tid][j];
            for (int i = 0; i < j; ++i) yh = fma((double)KE[i][tid], c[i], yh);
            const float yhat = (float)yh;
            yh_s[j][tid] = yhat;
            KE[j][tid] = ks[j] * (Ys[j][tid] - yhat);
        }
        __syncthreads();
        if (tid == 0) {
            float loss = 0.f;
            for (int o = 0; o < m; ++o) {
                const float e = Ys[j][o] - yh_s[j][o];
                loss = fmaf(e, e, loss);
            }
            ls[j] = loss;
        }
        __syncthreads();
    }
    // results: A, k, k err, predictions, losses
    if (tid < cnt * cnt) P.am[(tid / cnt) * K + tid % cnt] = A[tid / cnt][tid % cnt];
    if (tid < cnt) P.kk[tid] = ks[tid];
    for (int e = tid; e < m * cnt; e += 1024) {
        const int j = e / m, o = e - j * m;
        P.ke[j * m + o] = KE[j][o];
        if (P.pred0) P.pred0[(int64_t)j * P.pred_stride + o] = yh_s[j][o];
    }
    if (P.loss0)
        for (int e = tid; e < cnt * P.loss_stride; e += 1024) {
            const int j = e / P.loss_stride, q = e - j * P.loss_stride;
            if (q < (j + 1 < cnt ? P.loss_stride : P.loss_tail)) P.loss0[(int64_t)j * P.loss_stride + q] = ls[j];
        }
    if (tid == 0) {
        P.scal[0] = ks[cnt - 1];
        P.scal[1] = ls[cnt - 1];
        *reinterpret_cast<unsigned*>(P.scal + 3) = 0u;
    }
}

// z_j[q] for every update of the block, then the block's weight updates element by element in update order
__global__ __launch_bounds__(256) void rls_blk_apply_kernel(const RlsParams P) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= P.n_in) return;
    const int cnt = P.cnt, K = P.K;
    float z[8];
    float yv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) yv[j] = j < cnt ? P.yb[(int64_t)j * P.n_in + q] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < cnt) {
            double acc = (double)yv[j];
#pragma unroll
            for (int l = 0; l < 8; ++l)
                if (l < j) acc = fma(P.am[j * K + l], (double)yv[l], acc);
            z[j] = (float)acc;
            P.zp_out[(int64_t)j * P.n_in + q] = z[j];
        }
    }
    for (int o = 0; o < P.n_out; ++o) {
        float* wp = P.W + (int64_t)o * P.ldw + q;
        float w = *wp;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < cnt) w = w + P.ke[j * P.n_out + o] * z[j];
        *wp = w;
    }
}

constexpr int kSymRowsDefault = 64;
#ifdef RP_MEASUREMENT
std::atomic<int> g_sym_inflight{0};  // measurement build only (rp_set_rls_sym): rows in flight per wave; 0 = built-in
std::atomic<int> g_sym_per_cu{0}, g_sym_probe{0};  // measurement build only: residency cap (through an LDS allocation), ablation
#endif

template <bool UPDATE, int C, int R>
void launch_sym_rows(const RlsParams& P, hipStream_t st) {
    const int nb = (P.n_in + 1023) >> 10;
    const dim3 grid(nb * (nb + 1) / 2, 1024 / C);
    // An in-place read-modify-write walk is fastest with LITTLE in flight: ONE resident workgroup per CU (four waves, two
    // groups of rows each) -- 0.816 ms per update at n_in = 32768 against 0.858-0.866 with the chip filled (3-4 workgroups
    // per CU), scripts/archive/probes/rls_sym_cap.py; presumably because a tile's stores then reach DRAM pages its loads have
    // just opened.  The cap is an LDS allocation nobody uses.  Small matrices (fewer than four tiles per CU) are not
    // HBM-bound and run uncapped.
#ifdef RP_MEASUREMENT
    const int req = g_sym_per_cu, probe = g_sym_probe;
#else
    constexpr int req = 0;
#endif
    // (the read-only first pass of a sweep, UPDATE = false, is a plain stream and likes the chip full: 0.39 against 0.47 ms)
    const int per_cu = req == 15 ? 0 : req > 0 ? req
                       : (UPDATE && (int64_t)grid.x * grid.y >= 4 * (int64_t)rp_cu_count() ? 1 : 0);
    const size_t lds = per_cu > 0 ? ((size_t)rp_lds_limit() - 4096) / per_cu - 4096 : 0;  // (static: up to 1.6 KiB)
#ifdef RP_MEASUREMENT
    if constexpr (C == 32 && R == 8) {
        if (P.n_in % 1024 == 0 && probe) {
#define RP_SYM_PROBE(Q_)                                                                                              \
    if (probe == Q_) {                                                                                                \
        if (lds) rp_allow_dynamic_lds<rls_sym_kernel<UPDATE, C, R, true, Q_>>(rp_lds_limit() - 8192);                 \
        hipLaunchKernelGGL((rls_sym_kernel<UPDATE, C, R, true, Q_>), grid, dim3(kThreads), lds, st, P);               \
    }
            RP_SYM_PROBE(0) RP_SYM_PROBE(1) RP_SYM_PROBE(2) RP_SYM_PROBE(3)
#undef RP_SYM_PROBE
            return;  // (measurements only: no column sums, no z')
        }
    }
#endif
    if (P.n_in % 1024 == 0) {
        if (lds) rp_allow_dynamic_lds<rls_sym_kernel<UPDATE, C, R, true>>(rp_lds_limit() - 8192);
        hipLaunchKernelGGL((rls_sym_kernel<UPDATE, C, R, true>), grid, dim3(kThreads), lds, st, P);
    } else {
        if (lds) rp_allow_dynamic_lds<rls_sym_kernel<UPDATE, C, R, false>>(rp_lds_limit() - 8192);
        hipLaunchKernelGGL((rls_sym_kernel<UPDATE, C, R, false>), grid, dim3(kThreads), lds, st, P);
    }
    if (P.x_next)
        hipLaunchKernelGGL(rls_sym_zn_kernel<32>, dim3((P.n_in + 31) / 32), dim3(1024), (size_t)nb * 32 * sizeof(float), st, P, C);
}

template <bool UPDATE>
int launch_sym(const RlsParams& P, int rows, hipStream_t st) {
#ifdef RP_MEASUREMENT
    const int inflight = g_sym_inflight > 0 ? (int)g_sym_inflight : 4;
#else
    constexpr int inflight = 4;
#endif
    switch (rows * 100 + inflight) {
#define RP_SYM(C_, R_) case C_ * 100 + R_: launch_sym_rows<UPDATE, C_, R_>(P, st); return RP_OK;
        RP_SYM(16, 4) RP_SYM(32, 4) RP_SYM(64, 4)
#ifdef RP_MEASUREMENT
        RP_SYM(16, 2) RP_SYM(16, 8) RP_SYM(32, 2) RP_SYM(32, 8) RP_SYM(64, 1) RP_SYM(64, 2) RP_SYM(64, 8)
#endif
#undef RP_SYM
        default: return RP_ERR_ARG;
    }
}


#ifdef RP_MEASUREMENT
std::atomic<int> g_blk_R{0}, g_blk_per_cu{-1};  // measurement build only (rp_set_rls_blk): rows in flight, residency cap
#endif

template <int K, bool UPDATE, bool DOT, int C, int R>
void launch_symk_r(const RlsParams& P, hipStream_t st) {
    const int nb = (P.n_in + 1023) >> 10;
    const dim3 grid(nb * (nb + 1) / 2, 1024 / C);
    // the residency cap of launch_sym_rows -- except for K = 8: with eight updates and eight inputs per element the pass
    // is no longer a bare read-modify-write walk and wants the chip full
    int per_cu = (K < 8 && UPDATE && (int64_t)grid.x * grid.y >= 4 * (int64_t)rp_cu_count()) ? 1 : 0;
#ifdef RP_MEASUREMENT
    if (g_blk_per_cu >= 0) per_cu = g_blk_per_cu;
#endif
    const size_t lds = per_cu > 0 ? ((size_t)rp_lds_limit() - 4096) / per_cu - 28672 : 0;
    if (P.n_in % 1024 == 0) {
        if (lds) rp_allow_dynamic_lds<rls_symk_kernel<K, UPDATE, DOT, C, R, true>>(rp_lds_limit() - 32768);  // (up to 24 KiB static)
        hipLaunchKernelGGL((rls_symk_kernel<K, UPDATE, DOT, C, R, true>), grid, dim3(kThreads), lds, st, P);
    } else {
        if (lds) rp_allow_dynamic_lds<rls_symk_kernel<K, UPDATE, DOT, C, R, false>>(rp_lds_limit() - 32768);
        hipLaunchKernelGGL((rls_symk_kernel<K, UPDATE, DOT, C, R, false>), grid, dim3(kThreads), lds, st, P);
    }
    if (DOT)
        hipLaunchKernelGGL(rls_sym_zn_kernel<64>, dim3((P.n_in + 63) / 64, P.nrhs), dim3(1024), (size_t)nb * 64 * sizeof(float), st, P, C);
}

template <int K, bool UPDATE, bool DOT, int C>
void launch_symk(const RlsParams& P, hipStream_t st) {
    // rows in flight per wave (registers: K inputs and K updates per lane).  Measured at n_in = 32768, ms per update
    // (scripts/probes/rls_blocked_sweep.py): K = 8, 64-row tiles: R = 2 0.196 with one workgroup per CU, 0.160 / 0.159 with two /
    // no cap (R = 4: 0.170 / 0.167 / 0.169); K = 4: R = 4 0.237 with one per CU (0.248 / 0.249), R = 2 0.267 / 0.245 / 0.253
    // r05 (register cross-lane reductions, packed fma): K = 8, 128-row tiles: R = 4 0.1247 / 0.1248 (two per CU / no cap), R = 2 0.137 / 0.127
    constexpr int R = 4;
#ifdef RP_MEASUREMENT
    if (g_blk_R == 4 && R != 4) return launch_symk_r<K, UPDATE, DOT, C, 4>(P, st);
    if (g_blk_R == 2 && R != 2) return launch_symk_r<K, UPDATE, DOT, C, 2>(P, st);
#endif
    launch_symk_r<K, UPDATE, DOT, C, R>(P, st);
}

template <int K>
int launch_symk_rows(const RlsParams& P, int rows, bool update, bool dot, hipStream_t st) {
#define RP_SYMK(C_)                                                  \
    if (rows == C_) {                                                \
        if (update && dot) launch_symk<K, true, true, C_>(P, st);    \
        else if (update) launch_symk<K, true, false, C_>(P, st);     \
        else launch_symk<K, false, true, C_>(P, st);                 \
        return RP_OK;                                                \
    }
    RP_SYMK(64) RP_SYMK(128) RP_SYMK(256)
#undef RP_SYMK
    return RP_ERR_ARG;
}

// rp_rls.block > 1: P is streamed once per K updates (include/rectipy_hip.h; mirrored by oracle/rls_oracle.c: sweep_blocked)
int sweep_blocked(const rp_rls* r, RlsParams P, int sym_rows, int32_t steps, int32_t stride, const float* X, int64_t ldx,
                  const float* Y, int64_t ldy, float* pred, int64_t ldpred, float* loss, hipStream_t st) {
    const int K = r->block, n = r->n_in;
    if (K != 2 && K != 4 && K != 8) RP_FAIL(RP_ERR_ARG, "rp_rls.block must be 0, 1, 2, 4 or 8, got %d", K);
    if (!r->blk || (reinterpret_cast<uintptr_t>(r->blk) & 15u)) RP_FAIL(RP_ERR_ARG, "the blocked sweep needs the 16-byte aligned workspace blk");
    const int nb = (n + 1023) >> 10, chunks = (n + sym_rows - 1) / sym_rows;
    float* Yb = r->blk;
    float* Zb = Yb + (int64_t)K * n;
    P.cm = reinterpret_cast<double*>(Zb + (int64_t)K * n);
    P.gm = P.cm;                 // (the Gram form keeps G where the update-by-update form keeps c)
    P.am = P.cm + K * K;
    P.kk = reinterpret_cast<float*>(P.am + K * K);
    P.hm = P.kk + K;             // [n_out x K]
    P.ke = P.hm + (int64_t)r->n_out * K;  // [K x n_out]
    // predictions BETWEEN updates (stride > 1 with pred) need the weights after every single update: the update-by-update form
    const bool gram = (stride == 1 || !pred) && r->n_out <= 64;   // (the recurrence keeps its operands in LDS: <= 64 read-out channels)
    P.yb = Yb;
    P.K = K;
    P.zp = Zb;
    P.kp = P.kk;
    P.part_stride = (int64_t)n * nb;
    P.col_stride = (int64_t)chunks * r->ldc;
    P.xb_stride = (int64_t)stride * ldx;
    P.zn = Yb;
    P.z0 = Zb;
    P.x_next = nullptr;
    RP_HIP_CHECK(hipMemsetAsync(r->scratch + 3, 0, sizeof(float), st));
    const int updates = (steps + stride - 1) / stride;
    int pending = 0;
    auto pass = [&](int npend, int nrhs, const float* xb) -> int {
        P.npend = npend;
        P.nrhs = nrhs;
        P.xb = xb;
        const bool update = npend > 0, dot = nrhs > 0;
        if (!update && !dot) return RP_OK;
        switch (K) {
            case 2: return launch_symk_rows<2>(P, sym_rows, update, dot, st);
            case 4: return launch_symk_rows<4>(P, sym_rows, update, dot, st);
            default: return launch_symk_rows<8>(P, sym_rows, update, dot, st);
        }
    };
    for (int u0 = 0; u0 < updates; u0 += K) {
        const int cnt = updates - u0 < K ? updates - u0 : K;
        if (pass(pending, cnt, X + (int64_t)u0 * stride * ldx) != RP_OK) RP_FAIL(RP_ERR_ARG, "no blocked RLS kernel for %d rows per tile", sym_rows);
        if (gram) {  // the block's scalars from its Gram matrix: two launches per block
            const int t0 = u0 * stride;
            P.cnt = cnt;
            P.y0 = Y + (int64_t)t0 * ldy;
            P.y_stride = (int64_t)stride * ldy;
            P.pred0 = pred ? pred + (int64_t)t0 * ldpred : nullptr;
            P.pred_stride = (int64_t)stride * ldpred;
            P.loss0 = loss ? loss + t0 : nullptr;
            P.loss_stride = stride;
            const int t_last = (u0 + cnt - 1) * stride;
            P.loss_tail = t_last + stride <= steps ? stride : steps - t_last;
            P.zp_out = Zb;
            hipLaunchKernelGGL(rls_blk_gram_kernel, dim3(cnt * cnt + r->n_out * cnt), dim3(1024), 0, st, P);
            hipLaunchKernelGGL(rls_blk_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, st, P);
            pending = cnt;
            continue;
        }
        for (int j = 0; j < cnt; ++j) {
            const int t = (u0 + j) * stride;
            P.j = j;
            P.z = Zb + (int64_t)j * n;
            P.x = X + (int64_t)t * ldx;
            P.y = Y + (int64_t)t * ldy;
            P.pred = pred ? pred + (int64_t)t * ldpred : nullptr;
            P.loss_out = loss ? loss + t : nullptr;
            P.loss_count = t + stride <= steps ? stride : steps - t;
            // two launches per update: the dot products (z_j is there: Y_0 for the block's first update, else left by the
            // weights kernel of the update before), then the weights and the next z
            const int extra = cnt - 1 - j;
            hipLaunchKernelGGL(rls_blk_scalars_kernel, dim3(1 + extra + r->n_out), dim3(1024), 0, st, P, extra);
            hipLaunchKernelGGL(rls_blk_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, st, P, j + 1 < cnt ? 1 : 0);
            if (pred && stride > 1) {
                const int count = (t + stride <= steps ? stride : steps - t) - 1;
                if (count > 0) {
                    const int items = count * r->n_out;
                    hipLaunchKernelGGL(rls_predict_kernel, dim3((items + 3) / 4), dim3(256), 0, st, r->W, r->ldw, n, r->n_out,
                                       X + (int64_t)(t + 1) * ldx, ldx, pred + (int64_t)(t + 1) * ldpred, ldpred, count);
                }
            }
        }
        pending = cnt;
    }
    (void)pass(pending, 0, nullptr);
    if (!r->defer_mirror && nb > 1)
        hipLaunchKernelGGL(rls_mirror_kernel, dim3(nb * (nb - 1) / 2, 256), dim3(kThreads), 0, st, r->P, r->ldp, n);
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

#ifdef RP_MEASUREMENT
std::atomic<int> g_rls_R{0}, g_rls_U{0}, g_rls_nts{-1};  // measurement build only (rp_set_rls_tuning)
std::atomic<int> g_rls_blocks{0};  // measurement build only: grid cap override
std::atomic<int> g_rls_tile_u{0}, g_rls_tile_per_cu{0};  // measurement build only (rp_set_rls_tile): KiB-columns per tile, residency cap
#else
constexpr int g_rls_R = 0, g_rls_U = 0, g_rls_nts = -1, g_rls_blocks = 0, g_rls_tile_u = 0, g_rls_tile_per_cu = 0;
#endif

template <int R, int U, bool NTS, bool UPDATE>
void launch_cfg(const RlsParams& P, bool vec4, hipStream_t st) {
    int groups = (P.n_in + R - 1) / R;
    int blocks = (groups + 3) / 4;
    const int cap_req = g_rls_blocks;
    const int cap = cap_req > 0 ? cap_req : 2048;
    if (blocks > cap) blocks = cap;
    const int policy = g_rls_nts;
#ifdef RP_MEASUREMENT
    if (vec4 && policy >= 0 && (policy & 2)) {
        hipLaunchKernelGGL((rls_pass_kernel<R, U, NTS, true, UPDATE, false>), dim3(blocks), dim3(kThreads), 0, st, P);
        return;
    }
#endif
    (void)policy;
    if (vec4)
        hipLaunchKernelGGL((rls_pass_kernel<R, U, NTS, true, UPDATE>), dim3(blocks), dim3(kThreads), 0, st, P);
    else
        hipLaunchKernelGGL((rls_pass_kernel<R, 1, NTS, false, UPDATE>), dim3(blocks), dim3(kThreads), 0, st, P);
}

template <bool UPDATE>
void launch_pass(const RlsParams& P, bool vec4, hipStream_t st) {
    const int tr = g_rls_R, tu = g_rls_U, tn = g_rls_nts;
    const int R = tr ? tr : 2, U = tu ? tu : 4;
    const bool nts = tn < 0 ? true : (tn & 1) != 0;
    if (vec4 && P.partials) {  // one-shot tiles in address order + the per-row sum of the tile partials
        const int TU = g_rls_tile_u > 0 ? (int)g_rls_tile_u : kTileCols / 1024;
        const int per_cu = g_rls_tile_per_cu;
        const int tile_cols = 1024 * TU;
        const int tpr = (P.n_in + tile_cols - 1) / tile_cols;
        const unsigned blocks = (unsigned)P.n_in * (unsigned)tpr;
        const size_t lds = per_cu > 0 ? ((size_t)rp_lds_limit() - 1024) / (per_cu > 0 ? per_cu : 1) - 64 : 0;  // (the kernel has 16 static bytes)
#ifdef RP_MEASUREMENT
#define RP_TILE(TU_)                                                                                                  \
    if (TU == TU_) {                                                                                                  \
        if (nts) {                                                                                                    \
            if (lds) rp_allow_dynamic_lds<rls_tile_kernel<true, UPDATE, TU_>>(rp_lds_limit() - 1024);                        \
            hipLaunchKernelGGL((rls_tile_kernel<true, UPDATE, TU_>), dim3(blocks), dim3(kThreads), lds, st, P);      \
        } else {                                                                                                      \
            if (lds) rp_allow_dynamic_lds<rls_tile_kernel<false, UPDATE, TU_>>(rp_lds_limit() - 1024);                       \
            hipLaunchKernelGGL((rls_tile_kernel<false, UPDATE, TU_>), dim3(blocks), dim3(kThreads), lds, st, P);     \
        }                                                                                                             \
    }
#else
#define RP_TILE(TU_)                                                                                                  \
    if (TU == TU_) {                                                                                                  \
        if (lds) rp_allow_dynamic_lds<rls_tile_kernel<true, UPDATE, TU_>>(rp_lds_limit() - 1024);                     \
        hipLaunchKernelGGL((rls_tile_kernel<true, UPDATE, TU_>), dim3(blocks), dim3(kThreads), lds, st, P);           \
    }
#endif
        RP_TILE(kTileCols / 1024)
#ifdef RP_MEASUREMENT
        RP_TILE(2) RP_TILE(4)
#endif
#undef RP_TILE
        if (P.x_next && !P.no_dot)
            hipLaunchKernelGGL(rls_zn_kernel, dim3((P.n_in + 255) / 256), dim3(256), 0, st, P, tile_cols);
        return;
    }
#define RP_RLS(R_, U_)                                            \
    if (R == R_ && U == U_) {                                     \
        if (nts)                                                  \
            launch_cfg<R_, U_, true, UPDATE>(P, vec4, st);        \
        else                                                      \
            launch_cfg<R_, U_, false, UPDATE>(P, vec4, st);       \
        return;                                                   \
    }
#ifdef RP_MEASUREMENT
    RP_RLS(2, 4) RP_RLS(1, 4) RP_RLS(1, 8) RP_RLS(2, 2) RP_RLS(4, 2) RP_RLS(4, 1) RP_RLS(2, 8) RP_RLS(1, 16) RP_RLS(4, 4)
#endif
#undef RP_RLS
    (void)R;
    (void)U;
    (void)nts;
    launch_cfg<2, 4, true, UPDATE>(P, vec4, st);
}

}  // namespace

extern "C" int rp_sizeof_rls(void) { return (int)sizeof(rp_rls); }

#ifdef RP_MEASUREMENT
extern "C" int rp_set_rls_sym(int rows_in_flight) {
    g_sym_inflight = rows_in_flight & 0xff;
    g_sym_per_cu = (rows_in_flight >> 8) & 0xf;   // measurements only: workgroups per CU (0 = no cap)
    g_sym_probe = (rows_in_flight >> 12) & 0xf;   // measurements only: ablation of the (32, 8) kernel, see PROBE
    return RP_OK;
}

extern "C" int rp_set_rls_blk(int rows_in_flight, int workgroups_per_cu) {
    g_blk_R = rows_in_flight;          // 0: built-in; 2 / 4
    g_blk_per_cu = workgroups_per_cu;  // -1: built-in rule; 0: no cap; n: n workgroups per CU
    return RP_OK;
}

extern "C" int rp_set_rls_tile(int kcols_per_tile, int workgroups_per_cu) {
    if (kcols_per_tile != 0 && kcols_per_tile != 1 && kcols_per_tile != 2 && kcols_per_tile != 4)
        RP_FAIL(RP_ERR_ARG, "rp_set_rls_tile: tile of %d x 1024 columns is not built", kcols_per_tile);
    g_rls_tile_u = kcols_per_tile;
    g_rls_tile_per_cu = workgroups_per_cu;
    return RP_OK;
}

extern "C" int rp_set_rls_tuning(int rows_per_wave, int unroll, int nontemporal_store) {
    g_rls_R = rows_per_wave;
    g_rls_U = unroll & 0xff;          // bits 8..: grid cap in blocks (benchmark only)
    g_rls_blocks = unroll >> 8;
    g_rls_nts = nontemporal_store;    // -1 default; bit 0: non-temporal stores, bit 1: cached (not streaming) loads
    return RP_OK;
}
#endif

extern "C" int rp_rls_sweep(const rp_rls* r, int32_t steps, int32_t stride, const float* X, int64_t ldx,
                            const float* Y, int64_t ldy, float* pred, int64_t ldpred, float* loss, void* stream) {
    if (!r) RP_FAIL(RP_ERR_ARG, "rls is null");
    if (r->n_in <= 0 || r->n_out <= 0) RP_FAIL(RP_ERR_SHAPE, "n_in/n_out must be > 0");
    if (!r->P || !r->W || !r->z || !r->zn || !r->scratch) RP_FAIL(RP_ERR_ARG, "P, W, z, zn, scratch must be non-null");
    if (r->ldp < r->n_in || r->ldw < r->n_in) RP_FAIL(RP_ERR_SHAPE, "ldp/ldw < n_in");
    if (steps < 0 || stride < 1 || stride > 1024) RP_FAIL(RP_ERR_ARG, "steps >= 0 and 1 <= stride <= 1024 required");
    if (steps == 0) return RP_OK;
    if (!X || !Y || ldx < r->n_in || ldy < r->n_out) RP_FAIL(RP_ERR_SHAPE, "X/Y rows too short");
    if (pred && ldpred < r->n_out) RP_FAIL(RP_ERR_SHAPE, "ldpred < n_out");
    const bool vec4 = (r->n_in % 4 == 0) && (r->ldp % 4 == 0) && (ldx % 4 == 0) &&
                      !(reinterpret_cast<uintptr_t>(r->P) & 15u) && !(reinterpret_cast<uintptr_t>(X) & 15u) &&
                      !(reinterpret_cast<uintptr_t>(r->z) & 15u) && !(reinterpret_cast<uintptr_t>(r->zn) & 15u);
    hipStream_t st = static_cast<hipStream_t>(stream);

    RlsParams P = {};
    P.no_dot = (g_rls_nts >= 0 && (g_rls_nts & 4)) ? 1 : 0;
    P.P = r->P;
    P.ldp = r->ldp;
    P.n_in = r->n_in;
    P.n_out = r->n_out;
    P.W = r->W;
    P.ldw = r->ldw;
    P.beta_inv = r->beta_inv;
    P.scal = r->scratch;
    P.err = r->scratch + 4;
    P.partials = vec4 ? r->partials : nullptr;
    const bool sym = vec4 && r->partials && r->colparts;
    if (r->colparts && !sym) RP_FAIL(RP_ERR_ARG, "the symmetric pass (colparts) needs partials and the 16-byte path");
    if (sym && (r->ldc < r->n_in || r->ldc % 4 != 0 || (reinterpret_cast<uintptr_t>(r->colparts) & 15u)))
        RP_FAIL(RP_ERR_SHAPE, "colparts: ldc >= n_in, ldc %% 4 == 0 and 16-byte alignment required");
    P.colparts = sym ? r->colparts : nullptr;
    P.ldc = r->ldc;
    const int sym_rows = r->sym_rows > 0 ? r->sym_rows : kSymRowsDefault;
    if (sym && r->block > 1 ? (sym_rows != 64 && sym_rows != 128 && sym_rows != 256) : (sym_rows != 16 && sym_rows != 32 && sym_rows != 64))
        RP_FAIL(RP_ERR_ARG, "sym_rows must be 16, 32 or 64 (blocked sweep: 64, 128 or 256), got %d", sym_rows);
    if (r->block > 1) {
        if (!sym) RP_FAIL(RP_ERR_ARG, "rp_rls.block > 1 needs the symmetric pass (partials and colparts)");
        return sweep_blocked(r, P, sym_rows, steps, stride, X, ldx, Y, ldy, pred, ldpred, loss, st);
    }
    // launch-bound sizes, an update every step: the whole sweep in one launch
    // (the row-per-wave pass only: the tiled pass -- rp_rls.partials -- sums a row in its own order)
    // (either row-dot path of that pass: the 16-byte one, or -- n_in % 4 != 0, up to 512 inputs: its registers -- the 4-byte one)
    if (!sym && !r->partials && stride == 1 && steps >= 4 && r->n_out <= 4 && rp_resident_enabled() && g_rls_nts < 0 &&
        (vec4 ? (rp_resident_shape(r->n_in, r->ldp) && r->ldw % 4 == 0 && !(reinterpret_cast<uintptr_t>(r->W) & 15u)) : r->n_in <= 512)) {
        if (const int prc = rp_resident_poll(); prc != RP_OK) return prc;
        const int rrc = launch_resident_rls(r, steps, X, ldx, Y, ldy, pred, ldpred, loss, vec4, st);
        if (rrc != kResidentUnavailable) return rrc;
    }
    float* za = r->z;
    float* zb = r->zn;

    // z_0 = beta^-1 P x_0 (no update)
    P.z = za;
    P.zn = za;
    P.x_next = X;
    if (sym && launch_sym<false>(P, sym_rows, st) != RP_OK) RP_FAIL(RP_ERR_ARG, "no symmetric RLS kernel for %d rows per tile with this tuning", sym_rows);
    if (!sym) launch_pass<false>(P, vec4, st);

    RP_HIP_CHECK(hipMemsetAsync(r->scratch + 3, 0, sizeof(float), st));  // the scalar kernel's arrival counter
    for (int t = 0; t < steps; t += stride) {
        P.z = za;
        P.zn = zb;
        P.x = X + (int64_t)t * ldx;
        P.y = Y + (int64_t)t * ldy;
        P.pred = pred ? pred + (int64_t)t * ldpred : nullptr;
        P.loss_out = loss ? loss + t : nullptr;  // the loss of this update holds until the next one
        P.loss_count = t + stride <= steps ? stride : steps - t;
        hipLaunchKernelGGL(rls_scalars_kernel, dim3(1 + r->n_out), dim3(1024), 0, st, P);
        hipLaunchKernelGGL(rls_weights_kernel, dim3((r->n_in + 255) / 256), dim3(256), 0, st, P);
        const int next = t + stride;
        P.x_next = next < steps ? X + (int64_t)next * ldx : nullptr;
        if (sym) (void)launch_sym<true>(P, sym_rows, st);
        else launch_pass<true>(P, vec4, st);
        if (pred && stride > 1) {  // steps between updates are predicted with the updated weights
            const int count = (next <= steps ? stride : steps - t) - 1;
            if (count > 0) {
                const int items = count * r->n_out;
                hipLaunchKernelGGL(rls_predict_kernel, dim3((items + 3) / 4), dim3(256), 0, st, r->W, r->ldw, r->n_in,
                                   r->n_out, X + (int64_t)(t + 1) * ldx, ldx, pred + (int64_t)(t + 1) * ldpred, ldpred,
                                   count);
            }
        }
        float* tmp = za;
        za = zb;
        zb = tmp;
    }
    if (sym && !r->defer_mirror) {  // the caller sees a full P again: off-diagonal blocks copied to their transposed positions
        const int nbm = (r->n_in + 1023) / 1024;
        if (nbm > 1)
            hipLaunchKernelGGL(rls_mirror_kernel, dim3(nbm * (nbm - 1) / 2, 256), dim3(kThreads), 0, st, r->P, r->ldp, r->n_in);
    }
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

extern "C" int rp_rls_mirror(const rp_rls* r, void* stream) {
    if (!r || !r->P) RP_FAIL(RP_ERR_ARG, "rls / P is null");
    if (r->n_in <= 0 || r->ldp < r->n_in) RP_FAIL(RP_ERR_SHAPE, "n_in > 0 and ldp >= n_in required");
    if (r->n_in % 4 != 0 || r->ldp % 4 != 0 || (reinterpret_cast<uintptr_t>(r->P) & 15u))
        RP_FAIL(RP_ERR_ARG, "rp_rls_mirror needs n_in %% 4 == 0, ldp %% 4 == 0 and a 16-byte aligned P");
    const int nbm = (r->n_in + 1023) / 1024;
    if (nbm > 1)
        hipLaunchKernelGGL(rls_mirror_kernel, dim3(nbm * (nbm - 1) / 2, 256), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                           r->P, r->ldp, r->n_in);
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}
