// rnn_backward.hip -- hand-written adjoint (BPTT) sweep of the rate-neuron recurrence.
//
// Replaces torch.autograd over the unrolled loop of Network._bptt / _bptt_step
// (reference rectipy/network.py:1028-1040, 1123-1130) for the leaky-integrator models.
// With y_{t+1} = y_t + dt f(y_t), out_t = v_t or r_t = act(v_t), G_t = dL/d out_t:
//
//   u_{t+1} = k * lambda_{t+1}
//   z       = W^T u_{t+1}                                (4 n^2 bytes of W, HBM-bound)
//   lambda_t = lambda_{t+1} + dt (-lambda_{t+1}/tau + act'(v_t) * z) + G_t [* act'(v_t)]
//   grad_eta += dt lambda_{t+1};  grad_k += dt lambda_{t+1} (W r_t);
//   grad_tau += dt lambda_{t+1} v_t / tau^2;  grad_x_t = dt lambda_{t+1}
//   grad_W  += dt u_{t+1} r_t^T   -> deferred to ONE MFMA GEMM per window (outer_gemm.hip)
//
// W is stored row-major and is NOT copied or transposed: the transposed product is a
// column-strip accumulation.  A wave owns one 256-column strip (16 bytes per lane,
// 1 KiB contiguous per row) and a chunk of RC consecutive rows, streams them with
// non-temporal loads, scales each by the wave-uniform u_i and keeps 4 column sums per
// lane -- no cross-lane reduction at all.  The Q = ceil(n/RC) chunk partials per
// column are then added in chunk order by the element-wise finalize kernel, which
// also applies the adjoint update and prepares u for the next (earlier) step.  The
// summation order is a function of n only (mirrored in oracle/rnn_oracle.c).

#include <atomic>

#include "rp_common.h"
#include "rp_dot.h"
#include "rp_resident.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunks = 64;
#ifdef RP_MEASUREMENT
std::atomic<int> g_bwd_chunks{0};  // measurement build only (rp_set_bwd_chunks): overrides the chunk count, i.e. the summation order
#endif

// RC(n, rows): rows per chunk (multiple of 8) for `rows` rows of an n-column matrix.  The chunk count is part of the
// summation-order contract (oracle/rnn_oracle.c mirrors it): Q = clamp(3072 / ceil(n / 256), 8, 64), i.e. about 3072
// (chunk, 256-column strip) items -- 12 waves per CU, each with 2 rows x 1 KiB in flight, which is where the chip's
// HBM streams fastest (scripts/archive/tune_chunks.py, us per adjoint step, r02): n = 32768: Q = 16 634 (r01 rule), 20 616,
// 24 602, 28 638, 40 626 (599 with one row in flight), 64 655;  n = 65536: Q = 8 2508 (r01 rule), 12 2405, 24 2408;
// n = 16384: Q = 32 167 (r01 rule), 40 156, 48 168.  Best chunk counts of a later sweep over other sizes (Q, rows in
// flight x strip KiB, us): 8192: 40, 4x2, 47 (this rule: 64 -> 52);  12288: 40, 4x2, 94 (64 -> 96);  20480: 20, 4x2, 248
// (38);  24576: 28, 2x1, 342 (32 -> 345);  40960: 10, 4x2, 948 (19; 16 -> 964);  49152: 16, 2x1, 1349 (16).  Either
// ~3072 items of 2 rows x 1 KiB or ~768 items of 4 rows x 2 KiB -- 24 KiB in flight per CU both ways; the rule keeps
// the first form, which is the best one at the benchmark sizes (32768, 65536) and within 7 % elsewhere.
inline int chunk_rows(int n, int rows) {
    int q = 3072 / ((n + 255) / 256);
    q = q < 8 ? 8 : (q > kMaxChunks ? kMaxChunks : q);
#ifdef RP_MEASUREMENT
    if (g_bwd_chunks > 0) q = g_bwd_chunks;
#endif
    int rc = (rows + q - 1) / q;
    rc = (rc + 7) & ~7;
    return rc < 8 ? 8 : rc;
}

struct BwdParams {
    const float* W;
    int64_t ldw;
    int n;     // columns of W (global neuron count)
    int rows;  // rows of W held here (== n unless row-sharded)
    const float* z_direct;  // sharded: this rank's slice of W^T u after the reduce-scatter (else chunk partials are summed)
    int model;
    const float* p[RP_MAX_PARAMS];
    float dt;
    int rc, q;  // rows per chunk, number of chunks
    const float* u;  // [n] u_{t+1}
    float* partials;  // [q x n]
    // finalize
    int t;
    const float* v_t;
    const float* r_t;
    int grad_var;
    const float* g_t;
    float* lam;
    // prep for step t-1 (null u_next => last step)
    float* u_next;
    const float* v_prev;
    const float* acc_prev;
    float* grad_tau;
    float* grad_k;
    float* grad_eta;
    float* gx_prev;
    float* grad_act[3];  // LI_SIGMOID: gradients of r_max, s, v0 (rp_bwd.grad_p[RP_P_RMAX / SLOPE / V0]) or null
    // spiking (QIF) sweep
    float* lam_s;
    float* grad_tau_s;
    float slope, theta;
    int x_mode;
    const float* x_prev;   // input row of step t-1 (prep) ...
    const float* s_prev;   // ... and its s_{t-1}
    const float* s_t;
    float* lam_x;          // adjoint of the second state (QIF_SFA x, IK u)
    const float* aux_prev;
    const float* acc_t;
    float inv_dt;
    int x_target;
    // population-mean models (IKU, IK_BIEXP)
    float* lam_x2;           // adjoint of the synaptic rise variable (IK_BIEXP)
    const float* aux2_prev;  // x of step t-1
    float* means;            // [4]: mean(lx b/tau_u), mean(kappa lx) of the current adjoint; mean(v), mean(spike/dt) of step t-1
    // packed W^T (rp_bwd.spt_*)
    const float* spt_vals;
    const uint16_t* spt_cidx;
    const int32_t* spt_rowoff;
    const int32_t* spt_row_ids;
    const int32_t* spt_batch_class;
    const int32_t* spt_phase_window;
    int spt_phases, spt_batches, spt_block_rows;
};

// partials[q][j] = sum_{i in chunk q, ascending} W[i][j] * u[i]
// SW = 16-byte column groups per lane: a wave covers a strip of 256*SW columns (SW KiB contiguous per row).
template <bool VEC4, int U, int SW>
__device__ __forceinline__ void wt_partial_body(const BwdParams& P) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row indices and u[i] go scalar
    constexpr int CW = VEC4 ? 256 * SW : 64;  // columns per strip
    const int strips = (P.n + CW - 1) / CW;
    const int items = strips * P.q;
    for (int item = blockIdx.x * (kThreads / 64) + wave; item < items; item += gridDim.x * (kThreads / 64)) {
        const int strip = item % strips;
        const int chunk = item / strips;
        const int i0 = chunk * P.rc;
        const int i1 = min(i0 + P.rc, P.rows);
        const float* __restrict__ u = P.u;
        if constexpr (VEC4) {
            const int col = strip * CW + 4 * lane;
            bool ok[SW];
            rp_f4 a[SW];
#pragma unroll
            for (int g = 0; g < SW; ++g) {
                ok[g] = col + 256 * g < P.n;
                a[g] = rp_f4{0.f, 0.f, 0.f, 0.f};
            }
            // out-of-range column groups re-read the first group's address; their sums are discarded
            const float* wp = P.W + (int64_t)i0 * P.ldw + col;
            int goff[SW];
#pragma unroll
            for (int g = 0; g < SW; ++g) goff[g] = ok[g] ? 256 * g : 0;
            if (ok[0]) {
                int i = i0;
                for (; i + U <= i1; i += U) {
                    rp_f4 w[U][SW];
                    float uu[U];
#pragma unroll
                    for (int k = 0; k < U; ++k)
#pragma unroll
                        for (int g = 0; g < SW; ++g) w[k][g] = rp_ld4_stream(wp + (int64_t)k * P.ldw + goff[g]);
#pragma unroll
                    for (int k = 0; k < U; ++k) uu[k] = u[i + k];
#pragma unroll
                    for (int k = 0; k < U; ++k)
#pragma unroll
                        for (int g = 0; g < SW; ++g) {
                            a[g].x = fmaf(w[k][g].x, uu[k], a[g].x);
                            a[g].y = fmaf(w[k][g].y, uu[k], a[g].y);
                            a[g].z = fmaf(w[k][g].z, uu[k], a[g].z);
                            a[g].w = fmaf(w[k][g].w, uu[k], a[g].w);
                        }
                    wp += (int64_t)U * P.ldw;
                }
                for (; i < i1; ++i) {
                    float uu = u[i];
#pragma unroll
                    for (int g = 0; g < SW; ++g) {
                        rp_f4 w = rp_ld4_stream(wp + goff[g]);
                        a[g].x = fmaf(w.x, uu, a[g].x);
                        a[g].y = fmaf(w.y, uu, a[g].y);
                        a[g].z = fmaf(w.z, uu, a[g].z);
                        a[g].w = fmaf(w.w, uu, a[g].w);
                    }
                    wp += P.ldw;
                }
#pragma unroll
                for (int g = 0; g < SW; ++g)
                    if (ok[g]) *reinterpret_cast<rp_f4*>(P.partials + (int64_t)chunk * P.n + col + 256 * g) = a[g];
            }
        } else {
            const int col = strip * 64 + lane;
            if (col < P.n) {
                float a = 0.f;
                for (int i = i0; i < i1; ++i) a = fmaf(P.W[(int64_t)i * P.ldw + col], u[i], a);
                P.partials[(int64_t)chunk * P.n + col] = a;
            }
        }
    }
}

template <bool VEC4, int U, int SW>
__global__ __launch_bounds__(kThreads) void wt_partial_kernel(const BwdParams P) {
    wt_partial_body<VEC4, U, SW>(P);
}

#ifdef RP_MEASUREMENT
// measurement build only: override of (U, SW) and of the grid cap; 0 = built-in choice (rp_set_bwd_tuning, rp_set_bwd_grid)
std::atomic<int> g_bwd_U{0}, g_bwd_SW{0}, g_bwd_blocks{0};
#endif

void launch_wt_partial(const BwdParams& P, bool vec4, hipStream_t st) {
    if (!vec4) {
        const int strips = (P.n + 63) / 64;
        int blocks = (strips * P.q + 3) / 4;
        hipLaunchKernelGGL((wt_partial_kernel<false, 1, 1>), dim3(blocks > 4096 ? 4096 : blocks), dim3(kThreads), 0, st, P);
        return;
    }
    // measured (scripts/archive/size_sweep.py, scripts/archive/tune_backward.py): 2-KiB strips with 4 rows in flight are the fastest
    // shape from n = 8192 up (n = 32768: 6.7-6.8 TB/s vs 6.3 TB/s for 1-KiB strips); below that the narrow strips give
    // more waves.  2-KiB strips need >= 1024 waves.
    // r02: with >= 2560 items of 1-KiB strips, 2 rows in flight per wave on at most 768 workgroups (12 waves per CU,
    // 24 KiB in flight per CU) is the fastest shape: n = 32768 602 us vs 634 for 2-KiB strips x 4 rows.
    const bool many = ((P.n + 255) / 256) * P.q >= 2560;
    const bool wide = ((P.n + 511) / 512) * P.q >= 1024;
#ifdef RP_MEASUREMENT
    const int tu = g_bwd_U, tsw = g_bwd_SW;
#else
    constexpr int tu = 0, tsw = 0;
#endif
    // r05 (scripts/probes/adj_strips.py): at n = 16384 -- a member or an edge of two coupled populations -- 2-KiB strips x 2
    // rows in flight beat the 1-KiB strips (157.8 vs 165.2 us per 1 GiB matrix); at n = 32768 the 1-KiB strips stay ahead
    // (599 vs 620)
    const bool mid = many && P.n <= 16384 && ((P.n + 511) / 512) * P.q >= 1024;
    const int U = tu ? tu : (many ? 2 : wide ? 4 : 8), SW = tsw ? tsw : (mid ? 2 : many ? 1 : wide ? 2 : 1);
    const int strips = (P.n + 256 * SW - 1) / (256 * SW);
    int blocks = (strips * P.q + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (many && !tu && !tsw && blocks > 768) blocks = 768;
#ifdef RP_MEASUREMENT
    if (const int cap = g_bwd_blocks; cap > 0 && blocks > cap) blocks = cap;
#endif
#define RP_BW(U_, SW_)                                                                                         \
    if (U == U_ && SW == SW_) {                                                                                \
        hipLaunchKernelGGL((wt_partial_kernel<true, U_, SW_>), dim3(blocks), dim3(kThreads), 0, st, P);        \
        return;                                                                                                \
    }
    RP_BW(2, 1) RP_BW(4, 2) RP_BW(2, 2)  // (and (8, 1) below): the shapes the rule above picks
#ifdef RP_MEASUREMENT
    RP_BW(16, 1) RP_BW(4, 1) RP_BW(8, 2) RP_BW(6, 2) RP_BW(32, 1) RP_BW(12, 1) RP_BW(3, 2) RP_BW(5, 2)
    RP_BW(1, 1) RP_BW(3, 1) RP_BW(1, 2)
#endif
#undef RP_BW
    hipLaunchKernelGGL((wt_partial_kernel<true, 8, 1>), dim3(blocks), dim3(kThreads), 0, st, P);
}

__device__ __forceinline__ float act_deriv(const BwdParams& P, int j, float r) {
    if (P.model == RP_MODEL_LI_TANH) return 1.0f - r * r;
    // d/dv [r_max / (1 + exp(s (v0 - v)))] = s r (1 - r / r_max)
    return P.p[RP_P_SLOPE][j] * r * (1.0f - r / P.p[RP_P_RMAX][j]);
}

// Sum of the chunk partials of column j in chunk order; loads are issued 8 at a time so that their latencies overlap
// (the additions stay sequential: the order is part of the summation spec).
__device__ __forceinline__ float chunk_sum(const BwdParams& P, int j) {
    float z = 0.f;
    int c = 0;
    for (; c + 8 <= P.q; c += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = P.partials[(int64_t)(c + k) * P.n + j];
#pragma unroll
        for (int k = 0; k < 8; ++k) z = z + t[k];
    }
    for (; c < P.q; ++c) z = z + P.partials[(int64_t)c * P.n + j];
    return z;
}

// The per-neuron gradient sums of the rate sweep (tau, k, eta; the sigmoid's r_max, s, v0): the per-step kernels add into
// memory step by step, the resident sweep keeps neuron j's sums in registers from the sweep's first step to its last -- the same
// additions in the same order.
struct RateAcc {
    float eta, k, tau, act[3];
};
__device__ __forceinline__ RateAcc rate_acc_load(const BwdParams& P, int j) {
    RateAcc G;
    G.eta = P.grad_eta ? P.grad_eta[j] : 0.f;
    G.k = P.grad_k ? P.grad_k[j] : 0.f;
    G.tau = P.grad_tau ? P.grad_tau[j] : 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) G.act[q] = P.grad_act[q] ? P.grad_act[q][j] : 0.f;
    return G;
}
__device__ __forceinline__ void rate_acc_store(const BwdParams& P, int j, const RateAcc& G) {
    if (P.grad_eta) P.grad_eta[j] = G.eta;
    if (P.grad_k) P.grad_k[j] = G.k;
    if (P.grad_tau) P.grad_tau[j] = G.tau;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (P.grad_act[q]) P.grad_act[q][j] = G.act[q];
}

// u_next = k * lam and the per-neuron parameter gradients that pair lambda_{t+1} with step t; returns u_next[j]
__device__ __forceinline__ float prep_core(const BwdParams& P, int j, float lam, RateAcc& G) {
    const float u = P.p[RP_P_K][j] * lam;
    P.u_next[j] = u;
    const float dl = P.dt * lam;
    if (P.grad_eta) G.eta += dl;
    if (P.grad_k) G.k += dl * P.acc_prev[j];
    if (P.grad_tau) {
        const float tau = P.p[RP_P_TAU][j];
        G.tau += dl * (P.v_prev[j] / (tau * tau));
    }
    if (P.gx_prev) P.gx_prev[j] = dl;
    return u;
}
__device__ __forceinline__ void prep_step(const BwdParams& P, int j, float lam) {
    RateAcc G = rate_acc_load(P, j);
    (void)prep_core(P, j, lam, G);
    rate_acc_store(P, j, G);
}

__global__ void prep_kernel(const BwdParams P) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < P.rows) prep_step(P, j, P.lam[j]);
}

// z_full[c] = sum of the chunk partials of column c, in chunk order (sharded sweep: input of the reduce-scatter)
__global__ void chunk_sum_kernel(const BwdParams P, float* z_full) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.n) return;
    z_full[c] = chunk_sum(P, c);
}

// `lam`: the adjoint state of neuron j before the step (the per-step kernels read it from memory, the resident sweep keeps it in a
// register); returns the new one, *u_out: u_next[j] (0 at the sweep's last step)
__device__ __forceinline__ float finalize_core(const BwdParams& P, int j, float z, float lam, float* u_out, RateAcc& G) {
    const float d = act_deriv(P, j, P.r_t[j]);
    const float g = P.g_t ? P.g_t[j] : 0.f;
    const float gv = (P.grad_var == RP_OUT_STATE0) ? g : g * d;
    const float lam_new = (lam + P.dt * (-lam / P.p[RP_P_TAU][j] + d * z)) + gv;
    P.lam[j] = lam_new;
    if (P.model == RP_MODEL_LI_SIGMOID && (P.grad_act[0] || P.grad_act[1] || P.grad_act[2])) {
        // r_t = r_max / (1 + exp(s (v0 - v_t))) enters step t through k (W r_t) and, for output r, through the loss:
        // dL/dr_t = dt z (+ G_t);  dr/dr_max = r / r_max,  dr/ds = -(v0 - v) r (1 - r/r_max),  dr/dv0 = -act'(v)
        const float r = P.r_t[j], rmax = P.p[RP_P_RMAX][j];
        const float dr = P.dt * z + ((P.grad_var == RP_OUT_STATE0) ? 0.f : g);
        if (P.grad_act[0]) G.act[0] += dr * (r / rmax);
        if (P.grad_act[1]) G.act[1] += -(dr * ((P.p[RP_P_V0][j] - P.v_t[j]) * (r * (1.0f - r / rmax))));
        if (P.grad_act[2]) G.act[2] += -(dr * d);
    }
    *u_out = P.u_next ? prep_core(P, j, lam_new, G) : 0.f;
    return lam_new;
}

__device__ __forceinline__ void finalize_elem(const BwdParams& P, int j, float z) {
    float u;
    RateAcc G = rate_acc_load(P, j);
    (void)finalize_core(P, j, z, P.lam[j], &u, G);
    rate_acc_store(P, j, G);
}

__global__ void finalize_kernel(const BwdParams P) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.rows) return;
    finalize_elem(P, j, P.z_direct ? P.z_direct[j] : chunk_sum(P, j));
}

// ---- spiking sweep (QIF, QIF_SFA, LIF, IK) -------------------------------------------------------------
// forward step (nodes.py:382-392):  sp = H(v - theta), reset = sp.detach(), spike_in = sp/dt,
//   v' = (v + dt dv)(1 - reset) + reset v_reset,  s' = s + dt (-s/tau_s + spike_in [+ s_ext]),  aux' = aux + dt daux
// adjoint with lv, ls, lx = dL/d(v, s, aux)_{t+1}, a = lv (1 - reset_t), sg = 1/(1 + slope |v_t - theta|)^2
// (Spike.backward, nodes.py:479-481), z = W^T u:
//   QIF      u = k a            lv' = a + dt a (2v/tau)            + ls sg
//   LIF      u = k a            lv' = a + dt (-a/tau)              + ls sg
//   QIF_SFA  u = k a            lv' = a + dt a (2v/tau)            + (ls + alpha lx) sg     lx' = lx + dt (-lx/tau_x - a/tau)
//   IK       u = a g (E_r-v)/C  lv' = a + dt (a dfdv + lx b/tau_u) + (ls + kappa lx) sg     lx' = lx + dt (-lx/tau_u - a/C)
//            dfdv = (k ((v-v_r) + (v-v_theta)) - g (W s)) / C
//   IKU      as IK with the recovery driven by population means (ik.yaml:34-40): the local terms lx b/tau_u and
//            kappa lx become the population means  A = mean_i(lx_i b_i/tau_u_i),  B = mean_i(kappa_i lx_i):
//            lv' = a + dt (a dfdv + A) + (ls + B) sg
//   IK_BIEXP as IKU with s' = -s/tau_d + x, x' = -x/tau_r + spike (ik.yaml:42-68): the spike feeds x,
//            lv' = a + dt (a dfdv + A) + (lx2 + B) sg,   lx2' = lx2 + dt (-lx2/tau_r + ls)
//   all      ls' = ls + dt (-ls/tau_s + z)            (+ G_t on the output variable)

struct SpkGrad {  // per-neuron parameter gradient targets, resolved on the host
    float* p[RP_MAX_PARAMS];
};

__device__ __forceinline__ float spk_input(const BwdParams& P, const float* row, int j) {
    if (P.x_mode == RP_X_FULL) return row[j];
    if (P.x_mode == RP_X_BCAST) return row[0];
    return 0.f;
}

__device__ __forceinline__ void spk_prep_step(const BwdParams& P, const SpkGrad& G, int j, float lv, float ls, float lx,
                                              float lx2) {
    const float v = P.v_prev[j];
    const float sp = (v - P.theta >= 0.0f) ? 1.0f : 0.0f;
    const float a = lv * (1.0f - sp);
    const float da = P.dt * a;
    const float I = spk_input(P, P.x_prev, j);
    const float Im = P.x_target == RP_XT_MEMBRANE ? I : 0.f;
    const float spike_in = sp * P.inv_dt;
    float gx = 0.f;
    if (P.model >= RP_MODEL_IK) {  // IK, IKU, IK_BIEXP
        const float C = P.p[RP_P_IK_C][j], vr = P.p[RP_P_V_R][j], vt = P.p[RP_P_V_THETA][j];
        const float g = P.p[RP_P_G][j], Er = P.p[RP_P_E_R][j], b = P.p[RP_P_B][j], tu = P.p[RP_P_TAU_U][j];
        const float acc = P.acc_prev[j], uu = P.aux_prev[j];
        const float dvr = v - vr, dvt = v - vt;
        P.u_next[j] = a * ((g * (Er - v)) / C);
        const float dlx = P.dt * lx;
        if (G.p[RP_P_ETA]) G.p[RP_P_ETA][j] += da / C;
        if (G.p[RP_P_K]) G.p[RP_P_K][j] += da * ((dvr * dvt) / C);
        if (G.p[RP_P_G]) G.p[RP_P_G][j] += da * ((acc * (Er - v)) / C);
        if (G.p[RP_P_IK_C]) {
            const float num = ((((P.p[RP_P_K][j] * dvr) * dvt - uu) + Im) + P.p[RP_P_ETA][j]) + (g * acc) * (Er - v);
            G.p[RP_P_IK_C][j] += -(da * (num / (C * C)));
        }
        // recovery equation: driven by the neuron's own v and spike (IK) or by the population means of step t-1
        const float dvu = P.model == RP_MODEL_IK ? dvr : P.means[2] - vr;
        const float spu = P.model == RP_MODEL_IK ? spike_in : P.means[3];
        if (G.p[RP_P_B]) G.p[RP_P_B][j] += dlx * (dvu / tu);
        if (G.p[RP_P_TAU_U]) G.p[RP_P_TAU_U][j] += -(dlx * ((b * dvu - uu) / (tu * tu)));
        if (G.p[RP_P_KAPPA]) G.p[RP_P_KAPPA][j] += dlx * spu;
        if (P.model == RP_MODEL_IK_BIEXP && G.p[RP_P_TAU_R]) {
            const float tr = P.p[RP_P_TAU_R][j];
            G.p[RP_P_TAU_R][j] += (P.dt * lx2) * (P.aux2_prev[j] / (tr * tr));
        }
        if (G.p[RP_P_V_R]) G.p[RP_P_V_R][j] += -(da * ((P.p[RP_P_K][j] * dvt) / C)) - dlx * (b / tu);
        if (G.p[RP_P_V_THETA]) G.p[RP_P_V_THETA][j] += -(da * ((P.p[RP_P_K][j] * dvr) / C));
        if (G.p[RP_P_E_R]) G.p[RP_P_E_R][j] += da * ((g * acc) / C);
        gx = da / C;
    } else {
        const float tau = P.p[RP_P_TAU][j];
        P.u_next[j] = P.p[RP_P_K][j] * a;
        if (G.p[RP_P_K]) G.p[RP_P_K][j] += da * P.acc_prev[j];
        if (P.model == RP_MODEL_LIF) {
            if (G.p[RP_P_ETA]) G.p[RP_P_ETA][j] += da;
            if (G.p[RP_P_TAU]) G.p[RP_P_TAU][j] += da * (v / (tau * tau));
            gx = P.x_target == RP_XT_MEMBRANE ? da : P.dt * ls;
        } else {  // QIF, QIF_SFA
            if (G.p[RP_P_ETA]) G.p[RP_P_ETA][j] += da / tau;
            if (G.p[RP_P_TAU]) {
                const float eta_eff = P.model == RP_MODEL_QIF_SFA ? P.p[RP_P_ETA][j] - P.aux_prev[j] : P.p[RP_P_ETA][j];
                const float num = (v * v + eta_eff) + Im;
                G.p[RP_P_TAU][j] += -(da * (num / (tau * tau)));
            }
            gx = da / tau;
            if (P.model == RP_MODEL_QIF_SFA) {
                const float dlx = P.dt * lx;
                const float tx = P.p[RP_P_TAU_X][j];
                if (G.p[RP_P_ALPHA]) G.p[RP_P_ALPHA][j] += dlx * spike_in;
                if (G.p[RP_P_TAU_X]) G.p[RP_P_TAU_X][j] += dlx * (P.aux_prev[j] / (tx * tx));
            }
        }
    }
    if (G.p[RP_P_TAU_S]) {
        const float ts = P.p[RP_P_TAU_S][j];
        G.p[RP_P_TAU_S][j] += (P.dt * ls) * (P.s_prev[j] / (ts * ts));
    }
    if (P.gx_prev) P.gx_prev[j] = gx;
}

// Population sums of one adjoint step of the mean-coupled models, one block: the adjoint means A, B over the CURRENT
// lx (they enter lv of step t) and the forward means of v and spike/dt of step t-1 (they enter the parameter gradients
// the prep step pairs with that step).  1024 strided partial sums + halving tree, as mean_kernel of the forward pass.
__global__ __launch_bounds__(1024) void adj_mean_kernel(const BwdParams P) {
    __shared__ float red[4][1024];
    const int tid = threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int j = tid; j < P.n; j += 1024) {
        const float lx = P.lam_x[j];
        s0 = s0 + lx * (P.p[RP_P_B][j] / P.p[RP_P_TAU_U][j]);
        s1 = s1 + P.p[RP_P_KAPPA][j] * lx;
        if (P.u_next) {  // a step t-1 exists
            const float v = P.v_prev[j];
            s2 = s2 + v;
            s3 = s3 + ((v - P.theta >= 0.0f) ? 1.0f : 0.0f) * P.inv_dt;
        }
    }
    red[0][tid] = s0;
    red[1][tid] = s1;
    red[2][tid] = s2;
    red[3][tid] = s3;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if (tid < off) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][tid] = red[q][tid] + red[q][tid + off];
        }
        __syncthreads();
    }
    if (tid < 4) P.means[tid] = red[tid][0] / (float)P.n;
}

__global__ void spk_prep_kernel(const BwdParams P, const SpkGrad G) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < P.rows)
        spk_prep_step(P, G, j, P.lam[j], P.lam_s[j], P.lam_x ? P.lam_x[j] : 0.f, P.lam_x2 ? P.lam_x2[j] : 0.f);
}

__device__ __forceinline__ void spk_finalize_elem(const BwdParams& P, const SpkGrad& G, int j, float z) {
    const float lv = P.lam[j];
    const float ls = P.lam_s[j];
    const float lx = P.lam_x ? P.lam_x[j] : 0.f;
    const float lx2 = P.lam_x2 ? P.lam_x2[j] : 0.f;
    const float v = P.v_t[j];
    const float x = v - P.theta;
    const float reset = (x >= 0.0f) ? 1.0f : 0.0f;
    const float a = lv * (1.0f - reset);
    const float den = 1.0f + P.slope * fabsf(x);
    const float sg = 1.0f / (den * den);
    const float g = P.g_t ? P.g_t[j] : 0.f;
    const float gv = (P.grad_var == RP_OUT_STATE0) ? g : 0.f;
    const float gs = (P.grad_var == RP_OUT_STATE0) ? 0.f : g;
    float lv_new, lx_new = 0.f, lx2_new = 0.f;
    if (P.model == RP_MODEL_QIF) {
        lv_new = ((a + P.dt * (a * ((2.0f * v) / P.p[RP_P_TAU][j]))) + ls * sg) + gv;
    } else if (P.model == RP_MODEL_LIF) {
        lv_new = ((a + P.dt * (-a / P.p[RP_P_TAU][j])) + ls * sg) + gv;
    } else if (P.model == RP_MODEL_QIF_SFA) {
        const float tau = P.p[RP_P_TAU][j];
        lv_new = ((a + P.dt * (a * ((2.0f * v) / tau))) + (ls + P.p[RP_P_ALPHA][j] * lx) * sg) + gv;
        lx_new = lx + P.dt * (-lx / P.p[RP_P_TAU_X][j] - a / tau);
    } else {  // IK family
        const float C = P.p[RP_P_IK_C][j], tu = P.p[RP_P_TAU_U][j];
        const float dfdv = (P.p[RP_P_K][j] * ((v - P.p[RP_P_V_R][j]) + (v - P.p[RP_P_V_THETA][j])) -
                            P.p[RP_P_G][j] * P.acc_t[j]) / C;
        if (P.model == RP_MODEL_IK) {
            lv_new = ((a + P.dt * (a * dfdv + lx * (P.p[RP_P_B][j] / tu))) + (ls + P.p[RP_P_KAPPA][j] * lx) * sg) + gv;
        } else {  // the recovery variables of ALL neurons see this neuron's v and spike through the means
            const float into_spike = P.model == RP_MODEL_IK_BIEXP ? lx2 : ls;
            lv_new = ((a + P.dt * (a * dfdv + P.means[0])) + (into_spike + P.means[1]) * sg) + gv;
            if (P.model == RP_MODEL_IK_BIEXP) lx2_new = lx2 + P.dt * (-lx2 / P.p[RP_P_TAU_R][j] + ls);
        }
        lx_new = lx + P.dt * (-lx / tu - a / C);
    }
    const float ls_new = (ls + P.dt * (-ls / P.p[RP_P_TAU_S][j] + z)) + gs;
    P.lam[j] = lv_new;
    P.lam_s[j] = ls_new;
    if (P.lam_x) P.lam_x[j] = lx_new;
    if (P.lam_x2) P.lam_x2[j] = lx2_new;
    if (P.u_next) spk_prep_step(P, G, j, lv_new, ls_new, lx_new, lx2_new);
}

__global__ void spk_finalize_kernel(const BwdParams P, const SpkGrad G) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.rows) return;
    spk_finalize_elem(P, G, j, P.z_direct ? P.z_direct[j] : chunk_sum(P, j));
}

// ---- transposed-copy sweep for small n (rp_bwd.wt) -------------------------------------------------------------
// WT[j][i] = W[i][j]: 32 x 32 tiles through LDS (padded), both sides coalesced.  Once per sweep.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ W, int64_t ldw, int n,
                                                        float* __restrict__ WT, int64_t ldwt) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int tiles = (n + 31) / 32;
    const int bi = (blockIdx.x / tiles) * 32, bj = (blockIdx.x % tiles) * 32;
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        if (bi + r < n && bj + tx < n) tile[r][tx] = W[(int64_t)(bi + r) * ldw + bj + tx];
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        if (bj + r < n && bi + tx < n) WT[(int64_t)(bj + r) * ldwt + bi + tx] = tile[tx][r];
}

// One adjoint step in ONE launch: z_j = <row j of W^T, u> with the forward's row-dot (same summation tree as the step
// kernel), then the element-wise update of neuron j by the lane that owns the row.  P.W / P.ldw point at W^T.
template <bool SPIKING, int R, int U, bool VEC4>
__global__ __launch_bounds__(kThreads) void adj_rowdot_kernel(const BwdParams P, const SpkGrad G) {
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int groups = (P.n + R - 1) / R;
    const RowDotArgs D{P.W, P.ldw, P.n, P.n, 0, P.u};
    for (int g = blockIdx.x * (kThreads / 64) + wave; g < groups; g += gridDim.x * (kThreads / 64)) {
        const int row_base = g * R;
        float sum[R];
        if constexpr (VEC4)
            rp_dot_rows_v4<R, U, false>(D, row_base, lane, sum);
        else
            rp_dot_rows_s1<R>(D, row_base, lane, sum);
        float mine = 0.f;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (lane == j) mine = sum[j];
        if (lane < R && row_base + lane < P.n) {
            if constexpr (SPIKING)
                spk_finalize_elem(P, G, row_base + lane, mine);
            else
                finalize_elem(P, row_base + lane, mine);
        }
    }
}

// ---- the rate models' adjoint sweep in ONE launch at launch-bound sizes (r05; the machinery: rp_resident.h) --------------------
// The transposed-copy sweep (adj_rowdot_kernel: one launch per adjoint step) with the rows of W^T in registers for the whole
// sweep and u_{t+1} = k lam handed from wave to wave as tagged granules; the element-wise update is finalize_core, its lane keeps
// lam in a register.  Bit-identical to the launches.
struct AdjResidentArgs {
    rp_resident_ctl C;
    int steps;
    float* u_traj;
    const float* traj_v;
    const float* traj_r;
    const float* traj_acc;
    int64_t ld;
    const float* grad_out;
    int64_t ldg;
    float* grad_x;
    int64_t ldgx;
};

template <class L>
__global__ __launch_bounds__(kThreads) void resident_adj_kernel(const BwdParams P0, const AdjResidentArgs A) {
    static_assert(kThreads == 64 * kResidentWaves, "rp_resident.h counts four waves per workgroup");
    constexpr int RW = L::RW;
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int block = rp_resident_join(A.C);
    if (block < 0) return;
    const int row_w0 = (block * kResidentWaves + (threadIdx.x >> 6)) * RW;
    BwdParams P = P0;
    typename L::Rows rows;
    L::load_rows(rows, P.W, P.ldw, row_w0, lane, P.n);   // (P.W: the transposed copy)
    const int my_row = row_w0 + rp_rows_allreduce_row<RW>(lane);
    const bool epi_lane = (lane & (64 / RW - 1)) == 0 && my_row < P.n;
    float lam_reg = epi_lane ? P0.lam[my_row] : 0.f;
    RateAcc G = {};
    if (epi_lane) G = rate_acc_load(P0, my_row);   // (what the prep launch in front left there)
    const int T = A.steps;
    for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s;
        P.t = t;
        P.u = A.u_traj + (int64_t)t * A.ld;
        P.v_t = A.traj_v + (int64_t)t * A.ld;
        P.r_t = A.traj_r + (int64_t)t * A.ld;
        P.g_t = A.grad_out ? A.grad_out + (int64_t)t * A.ldg : nullptr;
        if (t > 0) {
            P.u_next = A.u_traj + (int64_t)(t - 1) * A.ld;
            P.v_prev = A.traj_v + (int64_t)(t - 1) * A.ld;
            P.acc_prev = A.traj_acc ? A.traj_acc + (int64_t)(t - 1) * A.ld : nullptr;
            P.gx_prev = A.grad_x ? A.grad_x + (int64_t)(t - 1) * A.ldgx : nullptr;
        } else {
            P.u_next = nullptr;
        }
        typename L::Vec r;
        if (s == 0) {   // u_T: written by the prep launch in front of this one
            L::load_vec(r, P.u, lane, P.n);
        } else if (!L::sweep(A.C, (s - 1) & 1, (unsigned)s, r, lane, P.n)) {
            return;
        }
        const float z = L::dots(rows, r, row_w0, lane);
        if (epi_lane) {
            float u;
            lam_reg = finalize_core(P, my_row, z, lam_reg, &u, G);
            if (t > 0) rp_resident_publish(A.C, s & 1, my_row, (unsigned)(s + 1), u);
        }
    }
    if (epi_lane) rate_acc_store(P0, my_row, G);
}

int launch_resident_adj(const BwdParams& P, const rp_bwd* b, bool v4, hipStream_t st) {
    AdjResidentArgs A = {};
    A.steps = b->steps;
    A.u_traj = b->u_traj;
    A.traj_v = b->traj_v;
    A.traj_r = b->traj_r;
    A.traj_acc = b->traj_acc;
    A.ld = b->ld;
    A.grad_out = b->grad_out;
    A.ldg = b->ldg;
    A.grad_x = b->grad_x;
    A.ldgx = b->ldgx;
    void* ws = nullptr;
    if (rp_resident_begin(P.n, st, &A.C, &ws, v4) != RP_OK) {
        (void)hipGetLastError();
        return kResidentUnavailable;
    }
    const dim3 grid(rp_resident_grid(A.C));
#define RP_LAUNCH_(L_) hipLaunchKernelGGL((resident_adj_kernel<RP_UNPAREN L_>), grid, dim3(kThreads), 0, st, P, A)
    RP_RESIDENT_DISPATCH_LAYOUT_WIDE(P.n, v4, RP_LAUNCH_)
#undef RP_LAUNCH_
    return rp_resident_end(A.C, ws, st, "the resident adjoint sweep");
}

// the spiking sweep (QIF, QIF_SFA, LIF, IK; not the population-mean models) the same way: spk_finalize_elem as it stands --
// adjoint states and gradient sums in memory, all a neuron's own --, then the u it left in row t - 1 of u_traj goes on as a granule
struct SpkResidentArgs {
    rp_resident_ctl C;
    int steps;
    float* u_traj;
    const float* traj_v;
    const float* traj_r;
    const float* traj_aux;
    const float* traj_aux2;
    const float* traj_acc;
    int64_t ld;
    const float* grad_out;
    int64_t ldg;
    float* grad_x;
    int64_t ldgx;
    const float* x;
    int64_t ldx;
};

template <class L>
__global__ __launch_bounds__(kThreads) void resident_spk_adj_kernel(const BwdParams P0, const SpkGrad G, const SpkResidentArgs A) {
    constexpr int RW = L::RW;
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int block = rp_resident_join(A.C);
    if (block < 0) return;
    const int row_w0 = (block * kResidentWaves + (threadIdx.x >> 6)) * RW;
    BwdParams P = P0;
    typename L::Rows rows;
    L::load_rows(rows, P.W, P.ldw, row_w0, lane, P.n);   // (P.W: the transposed copy)
    const int my_row = row_w0 + rp_rows_allreduce_row<RW>(lane);
    const bool epi_lane = (lane & (64 / RW - 1)) == 0 && my_row < P.n;
    const int T = A.steps;
    for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s;
        P.t = t;
        P.u = A.u_traj + (int64_t)t * A.ld;
        P.v_t = A.traj_v + (int64_t)t * A.ld;
        P.acc_t = A.traj_acc ? A.traj_acc + (int64_t)t * A.ld : nullptr;
        P.g_t = A.grad_out ? A.grad_out + (int64_t)t * A.ldg : nullptr;
        if (t > 0) {   // (spk_set_prep)
            const int64_t o = (int64_t)(t - 1) * A.ld;
            P.u_next = A.u_traj + o;
            P.v_prev = A.traj_v + o;
            P.s_prev = A.traj_r + o;
            P.aux_prev = A.traj_aux ? A.traj_aux + o : nullptr;
            P.aux2_prev = A.traj_aux2 ? A.traj_aux2 + o : nullptr;
            P.acc_prev = A.traj_acc ? A.traj_acc + o : nullptr;
            P.gx_prev = A.grad_x ? A.grad_x + (int64_t)(t - 1) * A.ldgx : nullptr;
            P.x_prev = A.x ? A.x + (int64_t)(t - 1) * A.ldx : nullptr;
        } else {
            P.u_next = nullptr;
        }
        typename L::Vec r;
        if (s == 0) {
            L::load_vec(r, P.u, lane, P.n);
        } else if (!L::sweep(A.C, (s - 1) & 1, (unsigned)s, r, lane, P.n)) {
            return;
        }
        const float z = L::dots(rows, r, row_w0, lane);
        if (epi_lane) {
            spk_finalize_elem(P, G, my_row, z);
            if (t > 0) rp_resident_publish(A.C, s & 1, my_row, (unsigned)(s + 1), P.u_next[my_row]);   // (this lane's own store)
        }
    }
}

int launch_resident_spk_adj(const BwdParams& P, const SpkGrad& G, const rp_bwd* b, bool v4, hipStream_t st) {
    SpkResidentArgs A = {};
    A.steps = b->steps;
    A.u_traj = b->u_traj;
    A.traj_v = b->traj_v;
    A.traj_r = b->traj_r;
    A.traj_aux = b->traj_aux;
    A.traj_aux2 = b->traj_aux2;
    A.traj_acc = b->traj_acc;
    A.ld = b->ld;
    A.grad_out = b->grad_out;
    A.ldg = b->ldg;
    A.grad_x = b->grad_x;
    A.ldgx = b->ldgx;
    A.x = b->x;
    A.ldx = b->ldx;
    void* ws = nullptr;
    if (rp_resident_begin(P.n, st, &A.C, &ws, v4) != RP_OK) {
        (void)hipGetLastError();
        return kResidentUnavailable;
    }
    const dim3 grid(rp_resident_grid(A.C));
#define RP_LAUNCH_(L_) hipLaunchKernelGGL((resident_spk_adj_kernel<RP_UNPAREN L_>), grid, dim3(kThreads), 0, st, P, G, A)
    RP_RESIDENT_DISPATCH_LAYOUT_WIDE(P.n, v4, RP_LAUNCH_)
#undef RP_LAUNCH_
    return rp_resident_end(A.C, ws, st, "the resident adjoint sweep");
}

// Launch shapes follow the forward's small-n choices (rnn_forward.hip launch_step): 1 row x 4 chunks per wave up to
// n = 2048, 2 rows x 2 chunks above.
template <bool SPIKING>
void launch_adj_rowdot(const BwdParams& P, const SpkGrad& G, bool vec4, hipStream_t st) {
    const int wpb = kThreads / 64;
    if (!vec4) {
        hipLaunchKernelGGL((adj_rowdot_kernel<SPIKING, 1, 1, false>), dim3((P.n + wpb - 1) / wpb), dim3(kThreads), 0, st,
                           P, G);
    } else if (P.n <= 2048) {
        hipLaunchKernelGGL((adj_rowdot_kernel<SPIKING, 1, 4, true>), dim3((P.n + wpb - 1) / wpb), dim3(kThreads), 0, st,
                           P, G);
    } else {
        const int groups = (P.n + 1) / 2;
        hipLaunchKernelGGL((adj_rowdot_kernel<SPIKING, 2, 2, true>), dim3((groups + wpb - 1) / wpb), dim3(kThreads), 0,
                           st, P, G);
    }
}

// One adjoint step on the packed rows of W^T (fixed sparse W): z_j = <packed row j of W^T, u> in the row-dot order of
// the transposed-copy sweep, then the element-wise update of neuron j -- the sparse step kernel with u as the gathered
// vector and the adjoint update as its epilogue.
template <bool SPIKING, bool LDS_SRC, int U>
__global__ __launch_bounds__(kSparseThreads) void adj_sparse_kernel(const BwdParams P, const SpkGrad G) {
    extern __shared__ float lds_u[];
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    constexpr int kWaves = kSparseThreads / RP_WAVE;
    if constexpr (LDS_SRC) rp_sparse_stage_source(lds_u, P.u, P.n, (P.n + 255) & ~255);
    const int lane_base = LDS_SRC ? lane : 4 * lane;  // (LDS: the lane's position inside a staged chunk, rp_sparse_stage_source)
    const int lane_rot = 0;
    const float* xs = LDS_SRC ? lds_u : P.u;
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    for (int j = blockIdx.x * kWaves + wave; j < P.n; j += gridDim.x * kWaves) {
        const int g0 = P.spt_rowoff[j], len = P.spt_rowoff[j + 1] - g0;
        const float* vp = P.spt_vals + (int64_t)g0 * 256 + 4 * lane;
        const uint16_t* ip = P.spt_cidx + (int64_t)g0 * 256 + 4 * lane;
        rp_f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < len; k += U) {
            rp_f4 v[U];
            us4 m[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + u < len) {
                    v[u] = rp_ld4_stream(vp + (int64_t)(k + u) * 256);
                    m[u] = __builtin_nontemporal_load(reinterpret_cast<const us4*>(ip + (int64_t)(k + u) * 256));
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + u < len) {
                    sparse_entry<LDS_SRC>(v[u].x, m[u].x, xs, lane_base, lane_rot, acc);
                    sparse_entry<LDS_SRC>(v[u].y, m[u].y, xs, lane_base, lane_rot, acc);
                    sparse_entry<LDS_SRC>(v[u].z, m[u].z, xs, lane_base, lane_rot, acc);
                    sparse_entry<LDS_SRC>(v[u].w, m[u].w, xs, lane_base, lane_rot, acc);
                }
        }
        const float z = rp_wave_allreduce(rp_lane_sum4(acc));
        if (lane == 0) {
            if constexpr (SPIKING)
                spk_finalize_elem(P, G, j, z);
            else
                finalize_elem(P, j, z);
        }
    }
}

// windowed form of the packed W^T (n > 32768): rp_sparse_windowed_rows with the adjoint update as the epilogue
template <bool SPIKING, int U>
__global__ __launch_bounds__(kSparseThreads) void adj_sparse_windowed_kernel(const BwdParams P, const SpkGrad G) {
    extern __shared__ float lds_u[];
    const SparseWindowed S{P.spt_vals, P.spt_cidx, P.spt_rowoff, P.spt_row_ids, P.spt_batch_class, P.spt_phase_window,
                           P.spt_phases, P.spt_batches};
    rp_sparse_windowed_rows<U>(S, P.u, P.n, lds_u, [&](int j, float z) {
        if constexpr (SPIKING)
            spk_finalize_elem(P, G, j, z);
        else
            finalize_elem(P, j, z);
    });
}

// blocked form of the packed W^T (n <= 32768): rp_sparse_blocked_rows with the adjoint update as the epilogue
template <bool SPIKING, int WAVES, int U>
__global__ __launch_bounds__(WAVES * RP_WAVE) void adj_sparse_blocked_kernel(const BwdParams P, const SpkGrad G) {
    extern __shared__ float lds_u[];
    const int n_pad = (P.n + 255) & ~255;
    rp_sparse_stage_source<WAVES * RP_WAVE>(lds_u, P.u, P.n, n_pad);
    rp_sparse_blocked_rows<WAVES, U>(P.spt_vals, P.spt_cidx, P.spt_rowoff, P.n, P.spt_block_rows, lds_u, lds_u + n_pad,
                                     [&](int j, float z) {
                                         if constexpr (SPIKING)
                                             spk_finalize_elem(P, G, j, z);
                                         else
                                             finalize_elem(P, j, z);
                                     });
}

template <bool SPIKING, int WAVES, int U>
int launch_adj_sparse_blocked_cfg(const BwdParams& P, const SpkGrad& G, hipStream_t st, int n_cu, size_t lds) {
    if (rp_allow_dynamic_lds<&adj_sparse_blocked_kernel<SPIKING, WAVES, U>>(rp_lds_limit()) != RP_OK) return RP_ERR_HIP;
    const int nblocks = (P.n + P.spt_block_rows - 1) / P.spt_block_rows;
    int blocks = (nblocks + WAVES - 1) / WAVES;
    if (blocks > n_cu) blocks = n_cu;
    hipLaunchKernelGGL((adj_sparse_blocked_kernel<SPIKING, WAVES, U>), dim3(blocks), dim3(WAVES * RP_WAVE), lds, st, P, G);
    return RP_OK;
}

template <bool SPIKING>
int launch_adj_sparse(const BwdParams& P, const SpkGrad& G, hipStream_t st) {
    const int n_cu = rp_cu_count();
    if (n_cu == 0) return RP_ERR_HIP;
    if (P.spt_block_rows > 0) {
        const size_t src_bytes = sizeof(float) * (size_t)((P.n + 255) & ~255);
        const size_t per_wave = sizeof(float) * RP_WAVE * (size_t)P.spt_block_rows;
        const size_t limit = rp_lds_limit();
        if (src_bytes + 16 * per_wave <= limit)
            return launch_adj_sparse_blocked_cfg<SPIKING, 16, 1>(P, G, st, n_cu, src_bytes + 16 * per_wave);  // (two register sets of one group: as the step kernel)
        if (src_bytes + 8 * per_wave <= limit)
            return launch_adj_sparse_blocked_cfg<SPIKING, 8, 8>(P, G, st, n_cu, src_bytes + 8 * per_wave);
        return RP_ERR_HIP;
    }
    if (P.spt_row_ids) {
        if (rp_allow_dynamic_lds<&adj_sparse_windowed_kernel<SPIKING, 2>>(sizeof(float) * kWindowChunks * 256) != RP_OK)
            return RP_ERR_HIP;
        const int wblocks = P.spt_batches < n_cu ? P.spt_batches : n_cu;
        if (wblocks > 0)
            hipLaunchKernelGGL((adj_sparse_windowed_kernel<SPIKING, 2>), dim3(wblocks), dim3(kSparseThreads),
                               sizeof(float) * kWindowChunks * 256, st, P, G);
        return RP_OK;
    }
    constexpr int kWaves = kSparseThreads / RP_WAVE;
    int blocks = (P.n + kWaves - 1) / kWaves;
    if (P.n <= kSparseLdsMaxN) {
        if (blocks > n_cu) blocks = n_cu;
        if (rp_allow_dynamic_lds<&adj_sparse_kernel<SPIKING, true, 4>>(sizeof(float) * kSparseLdsMaxN) != RP_OK)
            return RP_ERR_HIP;
        hipLaunchKernelGGL((adj_sparse_kernel<SPIKING, true, 4>), dim3(blocks), dim3(kSparseThreads),
                           sizeof(float) * (size_t)((P.n + 255) & ~255), st, P, G);
    } else {
        if (blocks > 2 * n_cu) blocks = 2 * n_cu;
        hipLaunchKernelGGL((adj_sparse_kernel<SPIKING, false, 4>), dim3(blocks), dim3(kSparseThreads), 0, st, P, G);
    }
    return RP_OK;
}

// packed W^T given: validates it and fills the pointers
int sparse_t_mode(const rp_net* net, const rp_bwd* b, BwdParams& P, bool& use) {
    use = b->spt_vals != nullptr;
    if (!use) return RP_OK;
    if (!b->spt_cidx || !b->spt_rowoff) RP_FAIL(RP_ERR_ARG, "spt_vals given without spt_cidx / spt_rowoff");
    if (net->n % 4 != 0) RP_FAIL(RP_ERR_SHAPE, "the packed form of W^T needs n %% 4 == 0 (n=%d)", net->n);
    if (net->rows != net->n) RP_FAIL(RP_ERR_ARG, "the packed W^T sweep needs the unsharded population");
    if ((reinterpret_cast<uintptr_t>(b->spt_vals) & 15u) || (reinterpret_cast<uintptr_t>(b->spt_cidx) & 7u) ||
        (reinterpret_cast<uintptr_t>(b->u_traj) & 15u) || (b->ld % 4 != 0))
        RP_FAIL(RP_ERR_ARG, "spt_vals / u_traj must be 16-byte (ld a multiple of 4) and spt_cidx 8-byte aligned");
    P.spt_vals = b->spt_vals;
    P.spt_cidx = b->spt_cidx;
    P.spt_rowoff = b->spt_rowoff;
    if (b->spt_row_ids && (!b->spt_batch_class || !b->spt_phase_window || b->spt_phases < 3 || b->spt_batches < 0))
        RP_FAIL(RP_ERR_ARG, "windowed packed W^T needs spt_batch_class, spt_phase_window, spt_phases >= 3, spt_batches");
    P.spt_row_ids = b->spt_row_ids;
    P.spt_batch_class = b->spt_batch_class;
    P.spt_phase_window = b->spt_phase_window;
    P.spt_phases = b->spt_phases;
    P.spt_batches = b->spt_batches;
    P.spt_block_rows = b->spt_block_rows;
    if (b->spt_block_rows < 0 || b->spt_block_rows > 64 || (b->spt_block_rows > 0 && (b->spt_row_ids || net->n > kSparseLdsMaxN)))
        RP_FAIL(RP_ERR_ARG, "spt_block_rows must be in [0, 64], without the windowed form and with n <= %d", kSparseLdsMaxN);
    return RP_OK;
}

// true when the sweep uses the transposed copy; validates the workspace
int transposed_mode(const rp_net* net, const rp_bwd* b, bool& use, bool& vec4_t) {
    use = b->wt != nullptr && net->n <= RP_BWD_TRANSPOSE_MAX_N;
    vec4_t = false;
    if (!use) return RP_OK;
    if (b->ldwt < net->n) RP_FAIL(RP_ERR_SHAPE, "ldwt < n");
    vec4_t = (net->n % 4 == 0) && (b->ldwt % 4 == 0);
    if (vec4_t && ((reinterpret_cast<uintptr_t>(b->wt) & 15u) || (reinterpret_cast<uintptr_t>(b->u_traj) & 15u) ||
                   (b->ld % 4 != 0)))
        RP_FAIL(RP_ERR_ARG, "wt and u_traj must be 16-byte aligned (ld a multiple of 4) for the transposed sweep");
    return RP_OK;
}

void launch_transpose(const rp_net* net, const rp_bwd* b, hipStream_t st) {
    const int tiles = (net->n + 31) / 32;
    hipLaunchKernelGGL(transpose_kernel, dim3(tiles * tiles), dim3(256), 0, st, net->W, net->ldw, net->n, b->wt,
                       b->ldwt);
}

}  // namespace

extern "C" {

int32_t rp_bwd_partial_rows(int32_t n) {
    if (n <= 0) return 0;
    int rc = chunk_rows(n, n);
    return (n + rc - 1) / rc;
}

int rp_rate_backward(const rp_net* net, const rp_bwd* b, void* stream) {
    if (const int prc = rp_resident_poll(); prc != RP_OK) return prc;
    if (!net || !b) RP_FAIL(RP_ERR_ARG, "net/bwd is null");
    if (net->model != RP_MODEL_LI_TANH && net->model != RP_MODEL_LI_SIGMOID)
        RP_FAIL(RP_ERR_ARG, "rp_rate_backward supports the leaky-integrator models only (model=%d)", net->model);
    if (net->rows != net->n || net->row0 != 0)
        RP_FAIL(RP_ERR_ARG, "rp_rate_backward needs the unsharded matrix (rows=%d n=%d)", net->rows, net->n);
    if (net->n <= 0 || net->ldw < net->n) RP_FAIL(RP_ERR_SHAPE, "bad n/ldw");
    if (b->steps < 0) RP_FAIL(RP_ERR_ARG, "steps < 0");
    if (!net->W || !b->traj_v || !b->traj_r || !b->lam || !b->u_traj || !b->partials)
        RP_FAIL(RP_ERR_ARG, "W, traj_v, traj_r, lam, u_traj, partials must be non-null");
    if (!net->p[RP_P_TAU] || !net->p[RP_P_K]) RP_FAIL(RP_ERR_ARG, "tau/k null");
    if (net->model == RP_MODEL_LI_SIGMOID && (!net->p[RP_P_RMAX] || !net->p[RP_P_SLOPE]))
        RP_FAIL(RP_ERR_ARG, "sigmoid parameters null");
    if (b->ld < net->n) RP_FAIL(RP_ERR_SHAPE, "ld < n");
    if (b->grad_out && b->ldg < net->n) RP_FAIL(RP_ERR_SHAPE, "ldg < n");
    if (b->grad_x && b->ldgx < net->n) RP_FAIL(RP_ERR_SHAPE, "ldgx < n");
    if (b->grad_k && !b->traj_acc) RP_FAIL(RP_ERR_ARG, "grad_k needs traj_acc");
    if (b->grad_var != RP_OUT_STATE0 && b->grad_var != RP_OUT_SOURCE) RP_FAIL(RP_ERR_ARG, "bad grad_var");
    const bool vec4 = (net->n % 4 == 0) && (net->ldw % 4 == 0);
    if (vec4 && ((reinterpret_cast<uintptr_t>(net->W) & 15u) || (reinterpret_cast<uintptr_t>(b->partials) & 15u)))
        RP_FAIL(RP_ERR_ARG, "W and partials must be 16-byte aligned");
    bool use_wt, vec4_t;
    {
        int rc = transposed_mode(net, b, use_wt, vec4_t);
        if (rc != RP_OK) return rc;
    }
    if (b->steps == 0) return RP_OK;

    hipStream_t st = static_cast<hipStream_t>(stream);
    const int n = net->n;
    const int T = b->steps;
    BwdParams P = {};
    P.W = use_wt ? b->wt : net->W;
    P.ldw = use_wt ? b->ldwt : net->ldw;
    P.n = n;
    P.rows = n;
    P.model = net->model;
    for (int q = 0; q < RP_MAX_PARAMS; ++q) P.p[q] = net->p[q];
    P.dt = net->dt;
    P.rc = chunk_rows(n, n);
    P.q = (n + P.rc - 1) / P.rc;
    P.partials = b->partials;
    P.grad_var = b->grad_var;
    P.lam = b->lam;
    P.grad_tau = b->grad_tau;
    P.grad_k = b->grad_k;
    P.grad_eta = b->grad_eta;
    if (net->model == RP_MODEL_LI_SIGMOID) {
        P.grad_act[0] = b->grad_p[RP_P_RMAX];
        P.grad_act[1] = b->grad_p[RP_P_SLOPE];
        P.grad_act[2] = b->grad_p[RP_P_V0];
        if ((P.grad_act[1] || P.grad_act[2]) && !net->p[RP_P_V0]) RP_FAIL(RP_ERR_ARG, "sigmoid v0 null");
    }

    const int eblocks = (n + 255) / 256;
    const int strips = vec4 ? (n + 255) / 256 : (n + 63) / 64;
    int mblocks = (strips * P.q + 3) / 4;
    if (mblocks > 4096) mblocks = 4096;

    auto set_prep = [&](int tp) {  // prepare u_{tp+1} (row tp of u_traj) and the grads paired with step tp
        P.u_next = b->u_traj + (int64_t)tp * b->ld;
        P.v_prev = b->traj_v + (int64_t)tp * b->ld;
        P.acc_prev = b->traj_acc ? b->traj_acc + (int64_t)tp * b->ld : nullptr;
        P.gx_prev = b->grad_x ? b->grad_x + (int64_t)tp * b->ldgx : nullptr;
    };

    bool use_spt;
    {
        int rc = sparse_t_mode(net, b, P, use_spt);
        if (rc != RP_OK) return rc;
    }
    if (use_spt) use_wt = false;
    if (use_wt) launch_transpose(net, b, st);
    set_prep(T - 1);
    hipLaunchKernelGGL(prep_kernel, dim3(eblocks), dim3(256), 0, st, P);
    if (use_wt && rp_resident_enabled() && T >= 4 &&
        (vec4_t ? (rp_resident_shape_wide(n, b->ldwt) && b->ld % 4 == 0 && !(reinterpret_cast<uintptr_t>(b->u_traj) & 15u)) : rp_resident_shape_any(n)))
    {
        const int rrc = launch_resident_adj(P, b, vec4_t, st);   // the whole sweep in one launch (the row-dot path the launches would take)
        if (rrc != kResidentUnavailable) return rrc;
    }
    for (int t = T - 1; t >= 0; --t) {
        P.t = t;
        P.u = b->u_traj + (int64_t)t * b->ld;
        if (!use_wt && !use_spt) launch_wt_partial(P, vec4, st);
        P.v_t = b->traj_v + (int64_t)t * b->ld;
        P.r_t = b->traj_r + (int64_t)t * b->ld;
        P.g_t = b->grad_out ? b->grad_out + (int64_t)t * b->ldg : nullptr;
        if (t > 0)
            set_prep(t - 1);
        else
            P.u_next = nullptr;
        if (use_spt) {
            if (launch_adj_sparse<false>(P, SpkGrad{}, st) != RP_OK) RP_FAIL(RP_ERR_HIP, "sparse adjoint launch failed");
        } else if (use_wt) {
            launch_adj_rowdot<false>(P, SpkGrad{}, vec4_t, st);
        } else {
            hipLaunchKernelGGL(finalize_kernel, dim3(eblocks), dim3(256), 0, st, P);
        }
    }
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

// ---- row-sharded sweep, one step at a time -------------------------------------------------------------------
// Rank p holds rows [row0, row0+rows) of W.  Its strip kernel yields a FULL-length partial W_p^T u_p; the caller
// reduce-scatters those over the ranks (RCCL) and hands back the rank's slice, then the element-wise update runs on
// the local neurons.  All per-neuron vectors of rp_bwd are local (length rows).

static int sharded_params(const rp_net* net, const rp_bwd* b, BwdParams& P, bool& vec4) {
    if (!net || !b) RP_FAIL(RP_ERR_ARG, "net/bwd is null");
    if (net->model != RP_MODEL_LI_TANH && net->model != RP_MODEL_LI_SIGMOID)
        RP_FAIL(RP_ERR_ARG, "the sharded adjoint supports the leaky-integrator models only (model=%d)", net->model);
    if (net->n <= 0 || net->rows <= 0 || net->row0 < 0 || (int64_t)net->row0 + net->rows > net->n || net->ldw < net->n)
        RP_FAIL(RP_ERR_SHAPE, "bad shard: n=%d rows=%d row0=%d", net->n, net->rows, net->row0);
    if (b->steps <= 0) RP_FAIL(RP_ERR_ARG, "steps must be > 0");
    if (!net->W || !b->traj_v || !b->traj_r || !b->lam || !b->u_traj || !b->partials)
        RP_FAIL(RP_ERR_ARG, "W, traj_v, traj_r, lam, u_traj, partials must be non-null");
    if (b->ld < net->rows) RP_FAIL(RP_ERR_SHAPE, "ld < rows");
    if (b->grad_out && b->ldg < net->rows) RP_FAIL(RP_ERR_SHAPE, "ldg < rows");
    if (b->grad_x && b->ldgx < net->rows) RP_FAIL(RP_ERR_SHAPE, "ldgx < rows");
    if (b->grad_k && !b->traj_acc) RP_FAIL(RP_ERR_ARG, "grad_k needs traj_acc");
    vec4 = (net->n % 4 == 0) && (net->ldw % 4 == 0);
    if (vec4 && ((reinterpret_cast<uintptr_t>(net->W) & 15u) || (reinterpret_cast<uintptr_t>(b->partials) & 15u)))
        RP_FAIL(RP_ERR_ARG, "W and partials must be 16-byte aligned");
    P = BwdParams{};
    P.W = net->W;
    P.ldw = net->ldw;
    P.n = net->n;
    P.rows = net->rows;
    P.model = net->model;
    for (int q = 0; q < RP_MAX_PARAMS; ++q) P.p[q] = net->p[q];
    P.dt = net->dt;
    P.rc = chunk_rows(net->n, net->rows);
    P.q = (net->rows + P.rc - 1) / P.rc;
    P.partials = b->partials;
    P.grad_var = b->grad_var;
    P.lam = b->lam;
    P.grad_tau = b->grad_tau;
    P.grad_k = b->grad_k;
    P.grad_eta = b->grad_eta;
    if (net->model == RP_MODEL_LI_SIGMOID) {
        P.grad_act[0] = b->grad_p[RP_P_RMAX];
        P.grad_act[1] = b->grad_p[RP_P_SLOPE];
        P.grad_act[2] = b->grad_p[RP_P_V0];
    }
    return RP_OK;
}

static void sharded_set_prep(const rp_bwd* b, BwdParams& P, int tp) {
    P.u_next = b->u_traj + (int64_t)tp * b->ld;
    P.v_prev = b->traj_v + (int64_t)tp * b->ld;
    P.acc_prev = b->traj_acc ? b->traj_acc + (int64_t)tp * b->ld : nullptr;
    P.gx_prev = b->grad_x ? b->grad_x + (int64_t)tp * b->ldgx : nullptr;
}

int32_t rp_bwd_shard_partial_rows(int32_t n, int32_t rows) {
    if (n <= 0 || rows <= 0) return 0;
    const int rc = chunk_rows(n, rows);
    return (rows + rc - 1) / rc;
}

static int spk_setup(const rp_net* net, const rp_bwd* b, BwdParams& P, SpkGrad& G, bool& vec4);
static void spk_set_prep(const rp_bwd* b, BwdParams& P, int tp);
static void spk_set_step(const rp_bwd* b, BwdParams& P, int t);
static bool is_spiking(const rp_net* net) { return net && net->model >= RP_MODEL_QIF; }

int rp_bwd_prepare(const rp_net* net, const rp_bwd* b, void* stream) {
    BwdParams P;
    bool vec4;
    if (is_spiking(net)) {
        SpkGrad G;
        int rc = spk_setup(net, b, P, G, vec4);
        if (rc != RP_OK) return rc;
        if (b->steps <= 0) RP_FAIL(RP_ERR_ARG, "steps must be > 0");
        spk_set_prep(b, P, b->steps - 1);
        if (net->model == RP_MODEL_IKU || net->model == RP_MODEL_IK_BIEXP)  // the adjoint's population means (unsharded only: spk_setup)
            hipLaunchKernelGGL(adj_mean_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), P);
        hipLaunchKernelGGL(spk_prep_kernel, dim3((P.rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), P, G);
        RP_HIP_CHECK(hipGetLastError());
        return RP_OK;
    }
    int rc = sharded_params(net, b, P, vec4);
    if (rc != RP_OK) return rc;
    sharded_set_prep(b, P, b->steps - 1);
    hipLaunchKernelGGL(prep_kernel, dim3((P.rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), P);
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

int rp_bwd_partial(const rp_net* net, const rp_bwd* b, int32_t t, float* z_full, void* stream) {
    BwdParams P;
    bool vec4;
    SpkGrad G;
    int rc = is_spiking(net) ? spk_setup(net, b, P, G, vec4) : sharded_params(net, b, P, vec4);
    if (rc != RP_OK) return rc;
    if (t < 0 || t >= b->steps || !z_full) RP_FAIL(RP_ERR_ARG, "bad step index or null z_full");
    hipStream_t st = static_cast<hipStream_t>(stream);
    P.u = b->u_traj + (int64_t)t * b->ld;
    launch_wt_partial(P, vec4, st);
    hipLaunchKernelGGL(chunk_sum_kernel, dim3((P.n + 255) / 256), dim3(256), 0, st, P, z_full);
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

int rp_bwd_apply(const rp_net* net, const rp_bwd* b, int32_t t, const float* z_local, void* stream) {
    BwdParams P;
    bool vec4;
    if (is_spiking(net)) {
        SpkGrad G;
        int rc = spk_setup(net, b, P, G, vec4);
        if (rc != RP_OK) return rc;
        if (t < 0 || t >= b->steps || !z_local) RP_FAIL(RP_ERR_ARG, "bad step index or null z_local");
        P.z_direct = z_local;
        spk_set_step(b, P, t);
        if (net->model == RP_MODEL_IKU || net->model == RP_MODEL_IK_BIEXP)
            hipLaunchKernelGGL(adj_mean_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), P);
        hipLaunchKernelGGL(spk_finalize_kernel, dim3((P.rows + 255) / 256), dim3(256), 0,
                           static_cast<hipStream_t>(stream), P, G);
        RP_HIP_CHECK(hipGetLastError());
        return RP_OK;
    }
    int rc = sharded_params(net, b, P, vec4);
    if (rc != RP_OK) return rc;
    if (t < 0 || t >= b->steps || !z_local) RP_FAIL(RP_ERR_ARG, "bad step index or null z_local");
    P.t = t;
    P.z_direct = z_local;
    P.v_t = b->traj_v + (int64_t)t * b->ld;
    P.r_t = b->traj_r + (int64_t)t * b->ld;
    P.g_t = b->grad_out ? b->grad_out + (int64_t)t * b->ldg : nullptr;
    if (t > 0)
        sharded_set_prep(b, P, t - 1);
    else
        P.u_next = nullptr;
    hipLaunchKernelGGL(finalize_kernel, dim3((P.rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), P);
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

// dst[c] += sum of the chunk partials of column c, in chunk order: the transposed edge product of rp_backward_coupled
// lands in the source's output gradient
__global__ void chunk_sum_add_kernel(const BwdParams P, float* dst) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.n) return;
    dst[c] = dst[c] + chunk_sum(P, c);
}


// ---- small coupled circuits: two launches per backward step ---------------------------------------------------------------
// Up to 4 members of at most 4096 neurons joined by at most 8 edges are launch-bound: the per-matrix path above issues two
// launches per edge and three per member and step.  Here ONE launch forms the chunk partials of every transposed product of
// the step (blockIdx.y = matrix: the members' W^T u and the edges' W_e^T gx, all of which depend on the step before only), and
// ONE launch does the rest per (member, neuron): adds the edge products into the row of the member's output gradient they
// belong to (edges in member / edge order, each a chunk-ordered sum: the order of the per-matrix path), sums the chunks of
// W^T u and applies the member's adjoint step.  Same additions in the same order: bit-identical to the per-matrix path.
// (measured with the limit lifted: 2 x 8192 neurons 460.2 vs 460.5 us per fit_bptt step, 2 x 16384 1411 vs 1396 -- above the
// launch-bound sizes the per-matrix launches lose nothing)
constexpr int kFusedMembers = 4, kFusedEdges = 8, kFusedMaxN = 4096;

struct PartArg {  // one transposed product: partials[chunk][j] = sum_{i in chunk} W[i][j] u[i]
    const float* W;
    int64_t ldw;
    const float* u;
    float* partials;
    int n, rows, rc, q;
};
struct PartArgs {
    PartArg a[kFusedMembers + kFusedEdges];
};

// (two plain kernels: this part of the file has C linkage)
#define RP_COUPLED_PARTIALS(NAME, U_, SW_)                                                                                     \
    __global__ __launch_bounds__(kThreads) void NAME(const PartArgs A) {                                                      \
        typedef const __attribute__((address_space(4))) char* karg_ptr;                                                       \
        PartArg a; /* this block's matrix, straight from the kernel-argument segment (scalar loads at a uniform offset) */    \
        __builtin_memcpy(&a, (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(PartArgs, a) + sizeof(PartArg) * blockIdx.y, \
                         sizeof(PartArg));                                                                                    \
        BwdParams P = {};                                                                                                     \
        P.W = a.W;                                                                                                            \
        P.ldw = a.ldw;                                                                                                        \
        P.u = a.u;                                                                                                            \
        P.partials = a.partials;                                                                                              \
        P.n = a.n;                                                                                                            \
        P.rows = a.rows;                                                                                                      \
        P.rc = a.rc;                                                                                                          \
        P.q = a.q;                                                                                                            \
        const int strips = (P.n + 256 * SW_ - 1) / (256 * SW_);                                                               \
        if ((int)blockIdx.x * (kThreads / 64) >= strips * P.q) return; /* (the grid is sized for the largest matrix) */       \
        wt_partial_body<true, U_, SW_>(P);                                                                                    \
    }
RP_COUPLED_PARTIALS(coupled_partials_kernel_8, 8, 1)
RP_COUPLED_PARTIALS(coupled_partials_kernel_2, 2, 1)
#undef RP_COUPLED_PARTIALS

struct EdgeSum {  // the chunk partials of one edge product and where their sum goes (null: the row lies before the window)
    const float* partials;
    float* dst;
    int n, q;
};
struct CoupledFin {
    BwdParams mem[kFusedMembers];
    SpkGrad grad[kFusedMembers];
    EdgeSum in[kFusedMembers][kFusedEdges];  // by SOURCE member, in member / edge order
    int n_in[kFusedMembers];
};

static_assert(sizeof(CoupledFin) <= 4000, "the argument block of coupled_finalize_kernel must fit the kernel-argument segment");

__global__ __launch_bounds__(256) void coupled_finalize_kernel(const CoupledFin F) {
    typedef const __attribute__((address_space(4))) char* karg_ptr;
    const karg_ptr base = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const int m = blockIdx.y;
    BwdParams P;
    SpkGrad G;
    __builtin_memcpy(&P, base + offsetof(CoupledFin, mem) + sizeof(BwdParams) * m, sizeof(BwdParams));
    __builtin_memcpy(&G, base + offsetof(CoupledFin, grad) + sizeof(SpkGrad) * m, sizeof(SpkGrad));
    int n_in;
    __builtin_memcpy(&n_in, base + offsetof(CoupledFin, n_in) + sizeof(int) * m, sizeof(int));
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P.rows) return;
    for (int e = 0; e < n_in; ++e) {
        EdgeSum E;
        __builtin_memcpy(&E, base + offsetof(CoupledFin, in) + sizeof(EdgeSum) * (m * kFusedEdges + e), sizeof(EdgeSum));
        if (!E.dst) continue;
        BwdParams Q = {};
        Q.partials = const_cast<float*>(E.partials);
        Q.n = E.n;
        Q.q = E.q;
        E.dst[j] = E.dst[j] + chunk_sum(Q, j);
    }
    const float z = chunk_sum(P, j);
    if (P.model >= RP_MODEL_QIF)
        spk_finalize_elem(P, G, j, z);
    else
        finalize_elem(P, j, z);
}

int rp_sizeof_bwd_member(void) { return (int)sizeof(rp_bwd_member); }


static int spk_setup(const rp_net* net, const rp_bwd* b, BwdParams& P, SpkGrad& G, bool& vec4);
static void spk_set_step(const rp_bwd* b, BwdParams& P, int t);

// few enough members and edges for ONE argument block of coupled_finalize_kernel; *small: every matrix small and on the
// 16-byte path as well, so that the transposed products of a step share one launch too
static bool coupled_fused_ok(const rp_bwd_member* mem, int count, bool* small) {
    *small = true;
    if (count > kFusedMembers) return false;
    int edges = 0;
    auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    for (int m = 0; m < count; ++m) {
        const rp_net* net = mem[m].net;
        if (net->n > kFusedMaxN || net->n % 4 || net->ldw % 4 || !aligned(net->W) || !aligned(mem[m].bwd->partials)) *small = false;
        int64_t off = 0;
        for (int e = 0; e < mem[m].n_edges; ++e) {
            const int nq = mem[mem[m].edge_source[e]].net->n;
            if (nq % 4 || mem[m].edge_ld[e] % 4 || !aligned(mem[m].edge_w[e]) || !aligned(mem[m].edge_work + off)) *small = false;
            const int rc = chunk_rows(nq, net->rows);
            off += (int64_t)((net->rows + rc - 1) / rc) * nq;
            ++edges;
        }
    }
    return edges <= kFusedEdges;
}

// `one_launch`: the transposed products of a step in ONE launch (launch-bound sizes); else one strip launch per matrix, each
// with the walk its shape wants (HBM-bound sizes) -- and in both cases ONE element-wise launch for all members.  Same partial
// sums, same order of additions.
static int coupled_fused_sweep(const rp_bwd_member* mem, int count, int T, hipStream_t st, bool one_launch) {
    CoupledFin F = {};
    PartArgs A = {};
    bool vec4;
    for (int m = 0; m < count; ++m) {
        const int rc = is_spiking(mem[m].net) ? spk_setup(mem[m].net, mem[m].bwd, F.mem[m], F.grad[m], vec4)
                                              : sharded_params(mem[m].net, mem[m].bwd, F.mem[m], vec4);
        if (rc != RP_OK) return rc;
    }
    int max_rows = 0, max_blocks = 1;
    for (int t = T - 1; t >= 0; --t) {
        int mats = 0;
        for (int m = 0; m < count; ++m) F.n_in[m] = 0;
        for (int m = 0; m < count; ++m) {
            const rp_bwd* b = mem[m].bwd;
            BwdParams& P = F.mem[m];
            A.a[mats++] = PartArg{P.W, P.ldw, b->u_traj + (int64_t)t * b->ld, P.partials, P.n, P.rows, P.rc, P.q};
            int64_t off = 0;
            for (int e = 0; e < mem[m].n_edges; ++e) {
                const int q = mem[m].edge_source[e];
                const int nq = mem[q].net->n;
                const int rc = chunk_rows(nq, P.rows), chunks = (P.rows + rc - 1) / rc;
                float* work = mem[m].edge_work + off;
                off += (int64_t)chunks * nq;
                const int tau = mem[m].edge_mode[e] == RP_EDGE_LAG ? t - 1 : t;
                if (tau < 0) continue;
                A.a[mats++] = PartArg{mem[m].edge_w[e], mem[m].edge_ld[e], b->grad_x + (int64_t)t * b->ldgx, work, nq, P.rows, rc, chunks};
                F.in[q][F.n_in[q]++] = EdgeSum{work, const_cast<float*>(mem[q].bwd->grad_out) + (int64_t)tau * mem[q].bwd->ldg, nq, chunks};
            }
            // this member's own step (rp_bwd_apply)
            if (is_spiking(mem[m].net)) {
                spk_set_step(b, P, t);
            } else {
                P.t = t;
                P.v_t = b->traj_v + (int64_t)t * b->ld;
                P.r_t = b->traj_r + (int64_t)t * b->ld;
                P.g_t = b->grad_out ? b->grad_out + (int64_t)t * b->ldg : nullptr;
                if (t > 0)
                    sharded_set_prep(b, P, t - 1);
                else
                    P.u_next = nullptr;
            }
            P.z_direct = nullptr;
            if (P.rows > max_rows) max_rows = P.rows;
        }
        // the strip shape of launch_wt_partial for the largest matrix (which strip shape a matrix is walked with never changes
        // its sums); the grid is sized for the matrix with the most items
        bool many = false;
        for (int i = 0; i < mats; ++i) many = many || ((A.a[i].n + 255) / 256) * A.a[i].q >= 2560;
        for (int i = 0; i < mats; ++i) {
            int blocks = (((A.a[i].n + 255) / 256) * A.a[i].q + 3) / 4;
            if (many && blocks > 768) blocks = 768;
            if (blocks > max_blocks) max_blocks = blocks;
        }
        if (!one_launch) {
            for (int i = 0; i < mats; ++i) {
                BwdParams Q = {};
                Q.W = A.a[i].W;
                Q.ldw = A.a[i].ldw;
                Q.u = A.a[i].u;
                Q.partials = A.a[i].partials;
                Q.n = A.a[i].n;
                Q.rows = A.a[i].rows;
                Q.rc = A.a[i].rc;
                Q.q = A.a[i].q;
                const bool v4 = (Q.n % 4 == 0) && (Q.ldw % 4 == 0) && !(reinterpret_cast<uintptr_t>(Q.W) & 15u) &&
                                !(reinterpret_cast<uintptr_t>(Q.partials) & 15u);
                launch_wt_partial(Q, v4, st);
            }
        } else if (many)
            hipLaunchKernelGGL(coupled_partials_kernel_2, dim3(max_blocks, mats), dim3(kThreads), 0, st, A);
        else
            hipLaunchKernelGGL(coupled_partials_kernel_8, dim3(max_blocks, mats), dim3(kThreads), 0, st, A);
        hipLaunchKernelGGL(coupled_finalize_kernel, dim3((max_rows + 255) / 256, count), dim3(256), 0, st, F);
    }
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

// lam[j] += c[j] (output variable = the state) or c[j] * act'(r_final[j]) (the source): a post-update read of the LAST step of
// the window ends in the final adjoint of its source (rp_backward_coupled)
__global__ void lam_add_kernel(const BwdParams E, const BwdParams Q, const float* r_final) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E.n) return;
    const float g = chunk_sum(E, j);
    const float gv = (Q.grad_var == RP_OUT_STATE0) ? g : g * act_deriv(Q, j, r_final[j]);
    Q.lam[j] = Q.lam[j] + gv;
}

// the same from a finished product vector c (a packed edge)
__global__ void lam_add_vec_kernel(const float* c, const BwdParams Q, const float* r_final, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float g = c[j];
    const float gv = (Q.grad_var == RP_OUT_STATE0) ? g : g * act_deriv(Q, j, r_final[j]);
    Q.lam[j] = Q.lam[j] + gv;
}

// ---- v18: the adjoint through an edge's memory, and of an activation-node member ----------------------------------------------
// FILTER edge: ly(t) = (chunk sum of W_e^T gx) + (chunk sum of F^T ly(t + 1)); dst += ly(t)
__global__ void filter_adj_kernel(const BwdParams E, const BwdParams F, float* lam, float* dst) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E.n) return;
    const float l = chunk_sum(E, j) + chunk_sum(F, j);
    lam[j] = l;
    dst[j] = dst[j] + l;
}

// DELAY edge: the total of c = W_e^T gx (per-thread chains over j = tid + 1024 m of the columns' chunk sums, 1024-leaf tree)
// goes to ONE element of the source's output gradient
__global__ __launch_bounds__(1024) void delay_adj_kernel(const BwdParams E, float* dst_elem) {
    __shared__ float red[1024];
    const int tid = threadIdx.x;
    float a = 0.f;
    for (int j = tid; j < E.n; j += 1024) a = a + chunk_sum(E, j);
    red[tid] = a;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if (tid < off) red[tid] = red[tid] + red[tid + off];
        __syncthreads();
    }
    if (tid == 0) *dst_elem = *dst_elem + red[0];
}

// activation-node member: gx(t) = g_out(t) * act'(out(t))
__global__ void instant_adj_kernel(const float* out_t, const float* g_t, float* gx_t, int n, int act) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float o = out_t[j];
    const float d = act == RP_ACT_TANH ? fmaf(-o, o, 1.0f) : (act == RP_ACT_SIGMOID ? o * (1.0f - o) : 1.0f);
    gx_t[j] = g_t[j] * d;
}

int rp_backward_coupled(const rp_bwd_member* mem, int32_t count, void* stream) {
    if (!mem || count < 1 || count > RP_MAX_MEMBERS) RP_FAIL(RP_ERR_ARG, "rp_backward_coupled: 1..%d members", RP_MAX_MEMBERS);
    bool plain = true;  // no activation-node member, no edge with memory: the launch-fused sweeps may take the circuit
    const int T = mem[0].bwd ? mem[0].bwd->steps : 0;
    bool post = false;  // an RP_EDGE_LIVE edge from an earlier member: it reads a post-update output
    bool any_packed = false;  // a fixed sparse edge whose transposed matrix comes in packed form (edge_spt)
    for (int m = 0; m < count; ++m) {
        const rp_bwd_member& M = mem[m];
        if (!M.net || !M.bwd) RP_FAIL(RP_ERR_ARG, "member %d: net / bwd is null", m);
        const bool instant = M.net->model == RP_MODEL_INSTANT;
        if (!instant && !M.z) RP_FAIL(RP_ERR_ARG, "member %d: z is null", m);
        if (M.net->rows != M.net->n || M.net->row0 != 0) RP_FAIL(RP_ERR_ARG, "member %d: row shards cannot be coupled", m);
        if (M.bwd->steps != T || T <= 0) RP_FAIL(RP_ERR_ARG, "member %d: every member sweeps the same window (steps > 0)", m);
        if (instant) {
            plain = false;
            if (!M.bwd->traj_r || !M.bwd->grad_out || !M.bwd->grad_x || M.bwd->ld < M.net->n || M.bwd->ldg < M.net->n || M.bwd->ldgx < M.net->n)
                RP_FAIL(RP_ERR_ARG, "member %d (activation node): traj_r (its outputs), grad_out and grad_x are required", m);
            if (M.bwd->spt_vals) RP_FAIL(RP_ERR_ARG, "member %d: an activation-node member has no matrix", m);
        }
        if (M.n_edges < 0 || M.n_edges > RP_MAX_EDGES) RP_FAIL(RP_ERR_ARG, "member %d: n_edges must be in [0, %d]", m, RP_MAX_EDGES);
        if (M.n_edges > 0 && (!M.bwd->grad_x || !M.edge_work)) RP_FAIL(RP_ERR_ARG, "member %d has edges: grad_x and edge_work are required", m);
        if (M.bwd->spt_vals) {
            any_packed = true;
            if (M.net->n % 4 != 0 || M.bwd->ld % 4 != 0 || (reinterpret_cast<uintptr_t>(M.bwd->u_traj) & 15u) || (reinterpret_cast<uintptr_t>(M.z) & 15u))
                RP_FAIL(RP_ERR_ARG, "member %d: the packed W^T needs n %% 4 == 0 and 16-byte aligned u_traj rows / z", m);
        }
        for (int e = 0; e < M.n_edges; ++e) {
            const int q = M.edge_source[e], mode = M.edge_mode[e];
            if (q < 0 || q >= count || !mem[q].net || !mem[q].bwd) RP_FAIL(RP_ERR_ARG, "member %d edge %d: bad source %d", m, e, q);
            if (mode != RP_EDGE_OUT && mode != RP_EDGE_LIVE && mode != RP_EDGE_LAG) RP_FAIL(RP_ERR_ARG, "member %d edge %d: bad mode %d", m, e, mode);
            if (mode == RP_EDGE_OUT && q >= m) RP_FAIL(RP_ERR_ARG, "member %d edge %d: RP_EDGE_OUT needs a source stepped earlier", m, e);
            if (mode == RP_EDGE_LIVE && q == m) RP_FAIL(RP_ERR_ARG, "member %d edge %d: a member cannot read its own live output", m, e);
            if (mode == RP_EDGE_LIVE && q < m) {
                post = true;
                if (mem[q].net->model != RP_MODEL_LI_TANH && mem[q].net->model != RP_MODEL_LI_SIGMOID)
                    RP_FAIL(RP_ERR_ARG, "member %d edge %d: a post-update read needs a rate population as its source", m, e);
                if (mem[q].bwd->grad_var != RP_OUT_STATE0 && !mem[q].final_src)
                    RP_FAIL(RP_ERR_ARG, "member %d edge %d: source %d needs final_src (its output variable is the source)", m, e, q);
            }
            if (!M.edge_w[e] || M.edge_ld[e] < mem[q].net->n) RP_FAIL(RP_ERR_SHAPE, "member %d edge %d: edge_w / edge_ld", m, e);
            if (mem[q].net->model == RP_MODEL_INSTANT && mode == RP_EDGE_LIVE)
                RP_FAIL(RP_ERR_ARG, "member %d edge %d: an edge from an activation node is RP_EDGE_OUT (this step's output) or RP_EDGE_LAG (the previous one)", m, e);
            const int kind = M.edge_mem_kind[e];
            if (kind != RP_EMEM_NONE) {
                plain = false;
                if (kind != RP_EMEM_FILTER && kind != RP_EMEM_DELAY) RP_FAIL(RP_ERR_ARG, "member %d edge %d: no adjoint for edge memory kind %d", m, e, kind);
                if (mode == RP_EDGE_LIVE && q < m) RP_FAIL(RP_ERR_ARG, "member %d edge %d: an edge with memory cannot be a post-update read here", m, e);
                if (M.edge_spt[e].vals) RP_FAIL(RP_ERR_ARG, "member %d edge %d: an edge with memory has no packed transposed form", m, e);
                if (kind == RP_EMEM_FILTER) {
                    if (!M.edge_F[e] || M.edge_ldf[e] < mem[q].net->n || !M.edge_lam[e] || !M.edge_fwork[e] ||
                        (reinterpret_cast<uintptr_t>(M.edge_lam[e]) & 15u) || (reinterpret_cast<uintptr_t>(M.edge_fwork[e]) & 15u))
                        RP_FAIL(RP_ERR_ARG, "member %d edge %d: a filter edge needs edge_F, edge_lam and edge_fwork (16-byte aligned)", m, e);
                } else if (M.edge_delay[e] < 0 || M.edge_chan[e] < 0 || M.edge_chan[e] >= mem[q].net->n)
                    RP_FAIL(RP_ERR_ARG, "member %d edge %d: edge_delay / edge_chan", m, e);
            }
            if (M.edge_spt[e].vals) {
                any_packed = true;
                if (M.net->rows % 4 != 0 || !M.edge_spt[e].cidx || !M.edge_spt[e].rowoff || (reinterpret_cast<uintptr_t>(M.edge_work) & 15u) ||
                    M.bwd->ldgx % 4 != 0 || (reinterpret_cast<uintptr_t>(M.bwd->grad_x) & 15u) || mem[q].bwd->ldg % 4 != 0 ||
                    (reinterpret_cast<uintptr_t>(mem[q].bwd->grad_out) & 15u))
                    RP_FAIL(RP_ERR_ARG, "member %d edge %d: a packed transposed edge needs rows %% 4 == 0 and 16-byte aligned gradient rows", m, e);
            }
            if (!mem[q].bwd->grad_out || mem[q].bwd->ldg < mem[q].net->n)
                RP_FAIL(RP_ERR_ARG, "member %d edge %d: source %d needs a (writable) grad_out buffer", m, e, q);
        }
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the chunk partials of c = W_e^T gx_x(row) for edge e of member x, left in x's edge_work
    auto edge_partials = [&](int x, int e, int row, BwdParams& P) {
        const rp_bwd_member& X = mem[x];
        const int q = X.edge_source[e];
        P = BwdParams{};
        P.W = X.edge_w[e];
        P.ldw = X.edge_ld[e];
        P.n = mem[q].net->n;
        P.rows = X.net->rows;
        P.rc = chunk_rows(P.n, P.rows);
        P.q = (P.rows + P.rc - 1) / P.rc;
        P.partials = X.edge_work;  // (the edges are evaluated one after the other: they share the workspace)
        P.u = X.bwd->grad_x + (int64_t)row * X.bwd->ldgx;
        const bool vec4 = (P.n % 4 == 0) && (P.ldw % 4 == 0) && !(reinterpret_cast<uintptr_t>(P.W) & 15u) &&
                          !(reinterpret_cast<uintptr_t>(P.partials) & 15u);
        launch_wt_partial(P, vec4, st);
    };
    // preparation, last member first: a post-update read of the window's last step ends in its source's final adjoint
    for (int q = count - 1; q >= 0; --q) {
        for (int x = q + 1; post && x < count; ++x)
            for (int e = 0; e < mem[x].n_edges; ++e) {
                if (mem[x].edge_source[e] != q || mem[x].edge_mode[e] != RP_EDGE_LIVE) continue;
                BwdParams E, Q;
                bool v4;
                const int rc = sharded_params(mem[q].net, mem[q].bwd, Q, v4);
                if (rc != RP_OK) return rc;
                if (mem[x].edge_spt[e].vals) {  // c into the workspace, then into lam
                    const int nq = mem[q].net->n;
                    const int prc = rp_internal_packed_matvec(&mem[x].edge_spt[e], mem[x].net->rows, nq,
                                                              mem[x].bwd->grad_x + (int64_t)(T - 1) * mem[x].bwd->ldgx, nullptr, mem[x].edge_work, st);
                    if (prc != RP_OK) RP_FAIL(prc, "packed edge launch failed (device properties / LDS attribute)");
                    hipLaunchKernelGGL(lam_add_vec_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, mem[x].edge_work, Q, mem[q].final_src, nq);
                    continue;
                }
                edge_partials(x, e, T - 1, E);
                hipLaunchKernelGGL(lam_add_kernel, dim3((E.n + 255) / 256), dim3(256), 0, st, E, Q, mem[q].final_src);
            }
        if (mem[q].net->model == RP_MODEL_INSTANT) continue;  // (no state: nothing to prepare)
        const int rc = rp_bwd_prepare(mem[q].net, mem[q].bwd, stream);
        if (rc != RP_OK) return rc;
    }
    bool small = false, means = !plain;  // (population-mean members need their reduction launch in front of every step: per-member path)
    for (int m = 0; m < count; ++m) means = means || mem[m].net->model == RP_MODEL_IKU || mem[m].net->model == RP_MODEL_IK_BIEXP;
    if (!post && !any_packed && !means && coupled_fused_ok(mem, count, &small)) return coupled_fused_sweep(mem, count, T, st, small);
    for (int t = T - 1; t >= 0; --t) {
        // last member first; in front of a member's step the edge products that end in its output gradient: from gx_x(t)
        // (ready since step t + 1) for the reads of step t's output -- into row t -- and of the output one step earlier
        // (RP_EDGE_LAG) -- into row t - 1 --, from gx_x(t - 1) (x comes later in the order: its step t is done) for the
        // post-update reads -- into row t
        for (int q = count - 1; q >= 0; --q) {
            for (int x = 0; x < count; ++x)
                for (int e = 0; e < mem[x].n_edges; ++e) {
                    if (mem[x].edge_source[e] != q) continue;
                    const int mode = mem[x].edge_mode[e];
                    const bool is_post = mode == RP_EDGE_LIVE && q < x;
                    const int row = is_post ? t - 1 : t;                     // the row of gx_x the edge carries back
                    const int tau = mode == RP_EDGE_LAG ? t - 1 : t;        // ... into this row of q's output gradient
                    if (row < 0 || tau < 0) continue;
                    float* dst = const_cast<float*>(mem[q].bwd->grad_out) + (int64_t)tau * mem[q].bwd->ldg;
                    if (mem[x].edge_spt[e].vals) {  // dst[j] = dst[j] + (row j of W_e^T) . gx_x(row): one packed launch
                        const int prc = rp_internal_packed_matvec(&mem[x].edge_spt[e], mem[x].net->rows, mem[q].net->n,
                                                                  mem[x].bwd->grad_x + (int64_t)row * mem[x].bwd->ldgx, dst, dst, st);
                        if (prc != RP_OK) RP_FAIL(prc, "packed edge launch failed (device properties / LDS attribute)");
                        continue;
                    }
                    BwdParams E;
                    edge_partials(x, e, row, E);
                    const int kind = mem[x].edge_mem_kind[e];
                    if (kind == RP_EMEM_FILTER) {  // ly(t) = c(t) + F^T ly(t + 1); ly(t) into the source's output gradient
                        BwdParams Fp = {};
                        Fp.W = mem[x].edge_F[e];
                        Fp.ldw = mem[x].edge_ldf[e];
                        Fp.n = Fp.rows = E.n;
                        Fp.rc = chunk_rows(Fp.n, Fp.rows);
                        Fp.q = (Fp.rows + Fp.rc - 1) / Fp.rc;
                        Fp.partials = mem[x].edge_fwork[e];
                        Fp.u = mem[x].edge_lam[e];
                        const bool v4 = (Fp.n % 4 == 0) && (Fp.ldw % 4 == 0) && !(reinterpret_cast<uintptr_t>(Fp.W) & 15u);
                        launch_wt_partial(Fp, v4, st);
                        hipLaunchKernelGGL(filter_adj_kernel, dim3((E.n + 255) / 256), dim3(256), 0, st, E, Fp, mem[x].edge_lam[e], dst);
                    } else if (kind == RP_EMEM_DELAY) {
                        const int rd = tau - mem[x].edge_delay[e];
                        if (rd >= 0)
                            hipLaunchKernelGGL(delay_adj_kernel, dim3(1), dim3(1024), 0, st, E,
                                               const_cast<float*>(mem[q].bwd->grad_out) + (int64_t)rd * mem[q].bwd->ldg + mem[x].edge_chan[e]);
                    } else
                        hipLaunchKernelGGL(chunk_sum_add_kernel, dim3((E.n + 255) / 256), dim3(256), 0, st, E, dst);
                }
            if (mem[q].net->model == RP_MODEL_INSTANT) {  // gx(t) = g_out(t) * act'(out(t)): ready for the edges into this node
                const rp_bwd* b = mem[q].bwd;
                const int n = mem[q].net->n;
                hipLaunchKernelGGL(instant_adj_kernel, dim3((n + 255) / 256), dim3(256), 0, st, b->traj_r + (int64_t)t * b->ld,
                                   b->grad_out + (int64_t)t * b->ldg, b->grad_x + (int64_t)t * b->ldgx, n, mem[q].net->act);
                continue;
            }
            int rc;
            if (mem[q].bwd->spt_vals) {  // a fixed sparse W: z = W^T u by the packed rows of W^T (rp_bwd.spt_*), one launch
                const rp_bwd* b = mem[q].bwd;
                rp_packed S = {};
                S.vals = b->spt_vals;
                S.cidx = b->spt_cidx;
                S.rowoff = b->spt_rowoff;
                S.row_ids = b->spt_row_ids;
                S.batch_class = b->spt_batch_class;
                S.phase_window = b->spt_phase_window;
                S.phases = b->spt_phases;
                S.batches = b->spt_batches;
                S.block_rows = b->spt_block_rows;
                rc = rp_internal_packed_matvec(&S, mem[q].net->n, mem[q].net->n, b->u_traj + (int64_t)t * b->ld, nullptr, mem[q].z, st);
                if (rc != RP_OK) RP_FAIL(rc, "packed W^T launch failed (device properties / LDS attribute)");
            } else {
                rc = rp_bwd_partial(mem[q].net, mem[q].bwd, t, mem[q].z, stream);
                if (rc != RP_OK) return rc;
            }
            rc = rp_bwd_apply(mem[q].net, mem[q].bwd, t, mem[q].z, stream);
            if (rc != RP_OK) return rc;
        }
    }
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

// RP_COMM_DIRECT: slice q of this rank's full-length partial goes into slot `rank` of owner q's inbox (peer memory mapped
// through hipIpc; blockIdx.y = peer), 16 bytes per lane.
struct ContribPush {
    const float* z_full;
    float* dst[63];   // peer q's inbox slot of this rank and parity
    int owner[63];    // q
    int quads, rows;
};
__global__ __launch_bounds__(256) void push_contrib_kernel(const ContribPush Q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < Q.quads)
        reinterpret_cast<rp_f4*>(Q.dst[blockIdx.y])[i] =
            reinterpret_cast<const rp_f4*>(Q.z_full + (int64_t)Q.owner[blockIdx.y] * Q.rows)[i];
}

// z_local[j] = the world contributions to neuron j of this rank added in rank order 0, 1, ..., world - 1 (the rank's own
// one read from its partial): the fixed cross-rank summation order of the direct sharded adjoint
__global__ __launch_bounds__(256) void reduce_contrib_kernel(const float* inbox, const float* own, float* z_local, int rows,
                                                            int world, int rank) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= rows) return;
    float s = rank == 0 ? own[j] : inbox[j];
    for (int p = 1; p < world; ++p) s = s + (p == rank ? own[j] : inbox[(int64_t)p * rows + j]);
    z_local[j] = s;
}

int rp_backward_sharded(const rp_net* net, const rp_bwd* b, const rp_comm* comm, void* stream) {
    typedef int (*reduce_scatter_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    if (!net || !b || !comm) RP_FAIL(RP_ERR_ARG, "net/bwd/comm is null");
    const bool direct = comm->mode == RP_COMM_DIRECT && comm->peer_inbox;
    if (direct) {
        if (!comm->work || !comm->inbox || !comm->peer_flags || !comm->flags)
            RP_FAIL(RP_ERR_ARG, "the direct sharded adjoint needs work (n + rows floats), inbox, peer_flags and flags");
        if (comm->world > 64 || net->rows % 4 != 0) RP_FAIL(RP_ERR_ARG, "the direct sharded adjoint: at most 64 ranks, rows %% 4 == 0");
        for (int p = 0; p < comm->world; ++p)
            if (!comm->peer_inbox[p] || !comm->peer_flags[p]) RP_FAIL(RP_ERR_ARG, "the direct sharded adjoint: peer %d has a null buffer", p);
        if (comm->rank >= 0 && comm->rank < comm->world && comm->peer_inbox[comm->rank] != comm->inbox)
            RP_FAIL(RP_ERR_ARG, "the direct sharded adjoint: this rank's entry of peer_inbox must be its own inbox");
    } else if (!comm->comm || !comm->reduce_scatter || !comm->work)
        RP_FAIL(RP_ERR_ARG, "rp_comm needs comm, reduce_scatter and work (n + rows floats)");
    if (comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world ||
        (int64_t)net->rows * comm->world != net->n || net->row0 != comm->rank * net->rows)
        RP_FAIL(RP_ERR_SHAPE, "shard rows=%d row0=%d does not match rank %d of %d over n=%d", net->rows, net->row0,
                comm->rank, comm->world, net->n);
    if (reinterpret_cast<uintptr_t>(comm->work) & 15u) RP_FAIL(RP_ERR_ARG, "work must be 16-byte aligned");
    float* z_full = comm->work;
    float* z_local = comm->work + net->n;
    int rc = rp_bwd_prepare(net, b, stream);
    if (rc != RP_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int t = b->steps - 1; t >= 0; --t) {
        rc = rp_bwd_partial(net, b, t, z_full, stream);
        if (rc != RP_OK) return rc;
        if (direct) {
            rp_comm* c = const_cast<rp_comm*>(comm);
            const int parity = (int)(c->seq & 1u);  // alternates with every exchange on these buffers
            ContribPush Q = {};
            Q.z_full = z_full;
            Q.rows = net->rows;
            Q.quads = net->rows / 4;
            int peers = 0;
            for (int q = 0; q < comm->world; ++q)
                if (q != comm->rank) {
                    Q.dst[peers] = comm->peer_inbox[q] + (int64_t)parity * net->n + (int64_t)comm->rank * net->rows;
                    Q.owner[peers++] = q;
                }
            if (peers > 0)
                hipLaunchKernelGGL(push_contrib_kernel, dim3((Q.quads + 255) / 256, peers), dim3(256), 0, st, Q);
            c->seq += 1;
            for (int q = 0; q < comm->world; ++q)
                if (q != comm->rank) RP_HIP_CHECK(hipStreamWriteValue32(st, comm->peer_flags[q] + comm->rank, c->seq, 0));
            for (int q = 0; q < comm->world; ++q)  // every peer's contribution of this step has landed in the inbox
                if (q != comm->rank)
                    RP_HIP_CHECK(hipStreamWaitValue32(st, comm->flags + q, c->seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(reduce_contrib_kernel, dim3((net->rows + 255) / 256), dim3(256), 0, st,
                               comm->inbox + (int64_t)parity * net->n, z_full + (int64_t)comm->rank * net->rows, z_local,
                               net->rows, comm->world, comm->rank);
        } else {
            const int nrc = reinterpret_cast<reduce_scatter_fn>(comm->reduce_scatter)(
                z_full, z_local, (size_t)net->rows, 7 /* ncclFloat32 */, 0 /* ncclSum */, comm->comm, st);
            if (nrc != 0) RP_FAIL(RP_ERR_COMM, "ncclReduceScatter returned %d at step %d", nrc, t);
        }
        rc = rp_bwd_apply(net, b, t, z_local, stream);
        if (rc != RP_OK) return rc;
    }
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

#ifdef RP_MEASUREMENT
int rp_set_bwd_chunks(int max_chunks) {
    g_bwd_chunks = max_chunks;
    return RP_OK;
}

int rp_set_bwd_grid(int max_workgroups) {
    g_bwd_blocks = max_workgroups;
    return RP_OK;
}

int rp_set_bwd_tuning(int unroll, int strip_groups) {
    g_bwd_U = unroll;
    g_bwd_SW = strip_groups;
    return RP_OK;
}
#endif

// Validation and parameter block of a spiking sweep over the `net->rows` local neurons (rows == n: whole population;
// rows < n: a row shard, all per-neuron vectors of rp_bwd local).
static int spk_setup(const rp_net* net, const rp_bwd* b, BwdParams& P, SpkGrad& G, bool& vec4) {
    if (!net || !b) RP_FAIL(RP_ERR_ARG, "net/bwd is null");
    const int model = net->model;
    if (model < RP_MODEL_QIF || model > RP_MODEL_IK_BIEXP)
        RP_FAIL(RP_ERR_ARG, "the spiking sweep needs a spiking model (model=%d)", model);
    const bool ik_family = model >= RP_MODEL_IK;
    const bool mean_coupled = model == RP_MODEL_IKU || model == RP_MODEL_IK_BIEXP;
    const bool has_aux = model == RP_MODEL_QIF_SFA || ik_family;
    const int rows = net->rows;
    if (net->n <= 0 || rows <= 0 || net->row0 < 0 || (int64_t)net->row0 + rows > net->n || net->ldw < net->n)
        RP_FAIL(RP_ERR_SHAPE, "bad shape: n=%d rows=%d row0=%d ldw=%lld", net->n, rows, net->row0, (long long)net->ldw);
    if (mean_coupled && rows != net->n) RP_FAIL(RP_ERR_ARG, "population-mean models cannot be row-sharded");
    if (b->steps < 0) RP_FAIL(RP_ERR_ARG, "steps < 0");
    if (!net->W || !b->traj_v || !b->traj_r || !b->lam || !b->lam2 || !b->u_traj || !b->partials)
        RP_FAIL(RP_ERR_ARG, "W, traj_v, traj_r (s_t), lam, lam2, u_traj, partials must be non-null");
    if (has_aux && (!b->traj_aux || !b->lam3)) RP_FAIL(RP_ERR_ARG, "model %d needs traj_aux and lam3", model);
    if (ik_family && !b->traj_acc) RP_FAIL(RP_ERR_ARG, "the IK adjoint needs traj_acc (W s_t)");
    if (mean_coupled && !b->means) RP_FAIL(RP_ERR_ARG, "model %d needs the means scratch (4 floats)", model);
    if (model == RP_MODEL_IK_BIEXP && (!b->traj_aux2 || !b->lam4))
        RP_FAIL(RP_ERR_ARG, "IK_BIEXP needs traj_aux2 and lam4");
    if (b->ld < rows) RP_FAIL(RP_ERR_SHAPE, "ld < rows");
    if (b->grad_out && b->ldg < rows) RP_FAIL(RP_ERR_SHAPE, "ldg < rows");
    if (b->grad_x && b->ldgx < rows) RP_FAIL(RP_ERR_SHAPE, "ldgx < rows");
    if (b->grad_var != RP_OUT_STATE0 && b->grad_var != RP_OUT_SOURCE) RP_FAIL(RP_ERR_ARG, "bad grad_var");
    if (b->x_mode != RP_X_NONE && b->x_mode != RP_X_FULL && b->x_mode != RP_X_BCAST)
        RP_FAIL(RP_ERR_ARG, "x_mode must be NONE, FULL or BCAST");
    if (b->x_mode != RP_X_NONE && (!b->x || b->ldx < (b->x_mode == RP_X_FULL ? rows : 1)))
        RP_FAIL(RP_ERR_SHAPE, "x/ldx inconsistent with x_mode");
    if (b->x_target != RP_XT_MEMBRANE && !(b->x_target == RP_XT_SYNAPSE && model == RP_MODEL_LIF))
        RP_FAIL(RP_ERR_ARG, "x_target %d is not available for model %d", b->x_target, model);

    // parameter-gradient targets by slot; the named fields of rp_bwd cover the common four
    G = SpkGrad{};
    for (int q = 0; q < RP_MAX_PARAMS; ++q) G.p[q] = b->grad_p[q];
    if (!ik_family && b->grad_tau) G.p[RP_P_TAU] = b->grad_tau;
    if (b->grad_k) G.p[RP_P_K] = b->grad_k;
    if (b->grad_eta) G.p[RP_P_ETA] = b->grad_eta;
    if (b->grad_tau_s) G.p[RP_P_TAU_S] = b->grad_tau_s;
    {
        bool ok[RP_MAX_PARAMS] = {};
        ok[RP_P_K] = ok[RP_P_ETA] = ok[RP_P_TAU_S] = true;
        if (ik_family) {
            ok[RP_P_IK_C] = ok[RP_P_G] = ok[RP_P_B] = ok[RP_P_TAU_U] = ok[RP_P_KAPPA] = true;
            ok[RP_P_V_R] = ok[RP_P_V_THETA] = ok[RP_P_E_R] = true;
            ok[RP_P_TAU_R] = model == RP_MODEL_IK_BIEXP;
        } else {
            ok[RP_P_TAU] = true;
            if (model == RP_MODEL_QIF_SFA) ok[RP_P_ALPHA] = ok[RP_P_TAU_X] = true;
        }
        for (int q = 0; q < RP_MAX_PARAMS; ++q)
            if (G.p[q] && !ok[q]) RP_FAIL(RP_ERR_ARG, "no gradient kernel for parameter slot %d of model %d", q, model);
    }
    if (G.p[RP_P_K] && !b->traj_acc) RP_FAIL(RP_ERR_ARG, "grad_k needs traj_acc");

    vec4 = (net->n % 4 == 0) && (net->ldw % 4 == 0);
    if (vec4 && ((reinterpret_cast<uintptr_t>(net->W) & 15u) || (reinterpret_cast<uintptr_t>(b->partials) & 15u)))
        RP_FAIL(RP_ERR_ARG, "W and partials must be 16-byte aligned");

    P = BwdParams{};
    P.W = net->W;
    P.ldw = net->ldw;
    P.n = net->n;
    P.rows = rows;
    P.model = model;
    for (int q = 0; q < RP_MAX_PARAMS; ++q) P.p[q] = net->p[q];
    P.dt = net->dt;
    P.inv_dt = 1.0f / net->dt;
    P.rc = chunk_rows(net->n, rows);
    P.q = (rows + P.rc - 1) / P.rc;
    P.partials = b->partials;
    P.grad_var = b->grad_var;
    P.lam = b->lam;
    P.lam_s = b->lam2;
    P.lam_x = has_aux ? b->lam3 : nullptr;
    P.lam_x2 = model == RP_MODEL_IK_BIEXP ? b->lam4 : nullptr;
    P.means = b->means;
    P.slope = b->slope;
    P.theta = net->theta;
    P.x_mode = b->x_mode;
    P.x_target = b->x_target;
    return RP_OK;
}

static void spk_set_prep(const rp_bwd* b, BwdParams& P, int tp) {
    P.u_next = b->u_traj + (int64_t)tp * b->ld;
    P.v_prev = b->traj_v + (int64_t)tp * b->ld;
    P.s_prev = b->traj_r + (int64_t)tp * b->ld;
    P.aux_prev = b->traj_aux ? b->traj_aux + (int64_t)tp * b->ld : nullptr;
    P.aux2_prev = b->traj_aux2 ? b->traj_aux2 + (int64_t)tp * b->ld : nullptr;
    P.acc_prev = b->traj_acc ? b->traj_acc + (int64_t)tp * b->ld : nullptr;
    P.gx_prev = b->grad_x ? b->grad_x + (int64_t)tp * b->ldgx : nullptr;
    P.x_prev = b->x ? b->x + (int64_t)tp * b->ldx : nullptr;
}

static void spk_set_step(const rp_bwd* b, BwdParams& P, int t) {
    P.t = t;
    P.v_t = b->traj_v + (int64_t)t * b->ld;
    P.acc_t = b->traj_acc ? b->traj_acc + (int64_t)t * b->ld : nullptr;
    P.g_t = b->grad_out ? b->grad_out + (int64_t)t * b->ldg : nullptr;
    if (t > 0)
        spk_set_prep(b, P, t - 1);
    else
        P.u_next = nullptr;
}

int rp_qif_backward(const rp_net* net, const rp_bwd* b, void* stream) {
    if (const int prc = rp_resident_poll(); prc != RP_OK) return prc;
    BwdParams P;
    SpkGrad G;
    bool vec4;
    int rc = spk_setup(net, b, P, G, vec4);
    if (rc != RP_OK) return rc;
    if (net->rows != net->n || net->row0 != 0)
        RP_FAIL(RP_ERR_ARG, "rp_qif_backward needs the unsharded matrix (rows=%d n=%d)", net->rows, net->n);
    const bool mean_coupled = net->model == RP_MODEL_IKU || net->model == RP_MODEL_IK_BIEXP;
    bool use_wt, vec4_t;
    rc = transposed_mode(net, b, use_wt, vec4_t);
    if (rc != RP_OK) return rc;
    if (b->steps == 0) return RP_OK;

    hipStream_t st = static_cast<hipStream_t>(stream);
    const int T = b->steps;
    bool use_spt;
    rc = sparse_t_mode(net, b, P, use_spt);
    if (rc != RP_OK) return rc;
    if (use_spt) use_wt = false;
    if (use_wt) {
        P.W = b->wt;
        P.ldw = b->ldwt;
    }
    const int eblocks = (P.n + 255) / 256;

    if (use_wt) launch_transpose(net, b, st);
    spk_set_prep(b, P, T - 1);
    if (mean_coupled) hipLaunchKernelGGL(adj_mean_kernel, dim3(1), dim3(1024), 0, st, P);
    hipLaunchKernelGGL(spk_prep_kernel, dim3(eblocks), dim3(256), 0, st, P, G);
    if (use_wt && !mean_coupled && rp_resident_enabled() && T >= 4 &&
        (vec4_t ? (rp_resident_shape_wide(P.n, b->ldwt) && b->ld % 4 == 0 && !(reinterpret_cast<uintptr_t>(b->u_traj) & 15u))
                : rp_resident_shape_any(P.n)))
    {
        const int rrc = launch_resident_spk_adj(P, G, b, vec4_t, st);   // the whole sweep in one launch
        if (rrc != kResidentUnavailable) return rrc;
    }
    for (int t = T - 1; t >= 0; --t) {
        P.u = b->u_traj + (int64_t)t * b->ld;
        if (!use_wt && !use_spt) launch_wt_partial(P, vec4, st);
        spk_set_step(b, P, t);
        if (mean_coupled) hipLaunchKernelGGL(adj_mean_kernel, dim3(1), dim3(1024), 0, st, P);
        if (use_spt) {
            if (launch_adj_sparse<true>(P, G, st) != RP_OK) RP_FAIL(RP_ERR_HIP, "sparse adjoint launch failed");
        } else if (use_wt) {
            launch_adj_rowdot<true>(P, G, vec4_t, st);
        } else {
            hipLaunchKernelGGL(spk_finalize_kernel, dim3(eblocks), dim3(256), 0, st, P, G);
        }
    }
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

}  // extern "C"
