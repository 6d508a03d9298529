// rnn_forward.hip -- fused Euler step of a dense recurrent population on gfx950.
//
// One launch = one integration step of
//   LI   (leaky_integrator.yaml:10,22,33)  v' = -v/tau + k*(W r) + I_ext + eta, r = act(v)
//   QIF  (qif.yaml:11-12, nodes.py:382-392) v' = (v^2+eta+I_ext)/tau + k*(W s), s' = -s/tau_s + spike/dt
// i.e. what RateNet.forward / SpikeResetNet.forward + the PyRates-generated field
// do with ~10 ATen ops.  The step is HBM-bound (4*n^2 bytes of W, 2*n^2 flop), so
// the kernel is a streaming mat-vec with the element-wise field, Euler update,
// nonlinearity / spike+reset, output sampling and trajectory stores fused into
// the epilogue of the row that produced the dot product.
//
// Mapping: one 64-lane wave owns R consecutive rows.  Lane l streams columns
// 256*m + 4*l .. +3 of each row with one 16-byte non-temporal load per row per
// m (1 KiB per wave-instruction, fully coalesced), multiplies by the matching
// 16 bytes of the source vector (L2-resident, shared by the R rows), and keeps
// 4 partial sums per row.  The wave then reduces with a fixed butterfly.  The
// summation tree per row is therefore independent of grid size, R, unrolling
// and row sharding; oracle/rnn_oracle.c mirrors it bit for bit.

#include <algorithm>
#include <atomic>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "rp_common.h"
#include "rp_dot.h"
#include "rp_resident.h"

thread_local char rp_err_buf[512] = "";

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWavesPerBlock * RP_WAVE;

constexpr int FLAG_BUFFER = 1;  // step >= cutoff: output joins the sampling buffer
constexpr int FLAG_SAMPLE = 2;  // record mean of the buffer at this step
constexpr int FLAG_FIRST = 4;   // first output of a buffer (acc_count was 0): the accumulator's old contents are not read

// pseudo-models of the step kernels (not part of the C ABI's model table):
constexpr int kModelEdgeSum = 64;  // a packed edge matrix of a coupled member: row sums chained into rp_run.edge_sum (below)
constexpr int kModelFilter = 65;   // the recursive filter of an edge with memory (rp_edge_mem): y' = F y (+ x), no state but src

struct EdgeArg {  // one dense edge into the population (rp_run.edge_*): the source vector of THIS step; W == nullptr: identity
    const float* W;
    int64_t ld;
    const float* src;
    int n, vec4;
};

struct StepParams {
    const float* W;
    int64_t ldw;
    int n, rows, row0;
    const float* p[RP_MAX_PARAMS];
    float* y[RP_MAX_STATE];
    const float* src_cur;
    float* src_next;
    float dt, inv_dt, theta, v_reset;
    float* scratch;
    int x_mode, m;
    const float* x_t;  // row t of the input
    const float* w_in;
    int64_t ldwin;
    int out_var, x_target;
    float* traj_t;
    float* save_state_t[RP_MAX_STATE];
    float* save_src_t;
    float* save_acc_t;
    float* save_in_t;
    float* acc;
    int flags;
    float count;
    float* sample_row;
    float* rec_row[RP_MAX_STATE];
    int* spike_events;
    int spike_capacity;
    int* spike_count;
    const float* spike_src;  // this step detects spikes on these values instead of v (rp_run.spike_src, first step only)
    int t_abs;
    // packed W (rp_net.sp_*): streamed instead of W when non-null
    const float* sp_vals;
    const uint16_t* sp_cidx;
    const int32_t* sp_rowoff;
    const int32_t* sp_row_ids;
    const int32_t* sp_batch_class;
    const int32_t* sp_phase_window;
    int sp_phases, sp_batches, sp_block_rows;
    // dense edges from other vectors into this population (rp_run.edge_*): the source vector of THIS step
    int n_edges;
    EdgeArg edge[RP_MAX_EDGES];
    float* hist_t;     // optional: this step's pre-update output (rp_run.hist[t_abs % 3])
    float* hist_next;  // optional (shared launch of rp_forward_coupled): the NEXT step's ring entry, i.e. this step's post-update output
    int model, vec4;   // (read by the shared-launch kernel only: the member's model and row-dot path)
    // the packed mat-vec of one dense-edge-turned-sparse (kModelEdgeSum): esum_out[i] = esum_prev[i] + (W_e s)_i, and behind the
    // member's last edge + its own input -- the node input I the member's step then reads as a full-width input
    float* esum_out;
    const float* esum_prev;
    int esum_final;
    int act;           // RP_MODEL_INSTANT: RP_ACT_*
};
// by-value kernel arguments: the block (and four of them in the shared launch) must fit the kernel-argument segment, and the
// edge descriptors are read from it at a runtime offset (step_edges_kernel)
static_assert(sizeof(StepParams) <= 1000, "StepParams grew: check MergedParams against the kernel-argument segment");
static_assert(offsetof(StepParams, edge) % 8 == 0 && sizeof(EdgeArg) == 32, "EdgeArg is read with scalar loads at a runtime offset");

// ---- row dot products (rp_dot.h) ---------------------------------------------

__device__ __forceinline__ RowDotArgs dot_args(const StepParams& P) {
    return RowDotArgs{P.W, P.ldw, P.n, P.rows, P.row0, P.src_cur};
}

template <int R, int U, bool NT>
__device__ __forceinline__ void dot_rows_v4(const StepParams& P, int row_base, int lane, float (&sum)[R]) {
    rp_dot_rows_v4<R, U, NT>(dot_args(P), row_base, lane, sum);
}

template <int R>
__device__ __forceinline__ void dot_rows_s1(const StepParams& P, int row_base, int lane, float (&sum)[R]) {
    rp_dot_rows_s1<R>(dot_args(P), row_base, lane, sum);
}

// ---- per-neuron epilogues -----------------------------------------------------

__device__ __forceinline__ float input_current(const StepParams& P, int i) {
    switch (P.x_mode) {
        case RP_X_FULL: return P.x_t[i];
        case RP_X_BCAST: return P.x_t[0];
        case RP_X_PROJ: {  // the input edge `weights @ x` (edges.py:48-49), sequential in j
            float s = 0.f;
            const float* wr = P.w_in + (int64_t)i * P.ldwin;
            for (int j = 0; j < P.m; ++j) s = fmaf(wr[j], P.x_t[j], s);
            return s;
        }
        default: return 0.f;
    }
}

// running-buffer mean of the node output (network.py:590-597)
__device__ __forceinline__ void sample_output(const StepParams& P, int i, float out) {
    if (P.traj_t) P.traj_t[i] = out;
    if (P.hist_t) P.hist_t[i] = out;
    if (P.flags & FLAG_BUFFER) {
        float a = ((P.flags & FLAG_FIRST) ? 0.0f : P.acc[i]) + out;
        if (P.flags & FLAG_SAMPLE) {
            if (P.sample_row) P.sample_row[i] = a / P.count;
            a = 0.f;
        }
        P.acc[i] = a;
    }
}

// A rate neuron's parameters in registers (the resident kernel reads them once per call) and -v / tau of the coming step, formed
// while the wave waits for the source vector
struct RateRegs {
    float tau, k, eta, rmax, slope, v0, mv_tau;
};
template <int MODEL>
__device__ __forceinline__ RateRegs rate_regs_load(const StepParams& P, int i) {
    RateRegs R = {};
    R.tau = P.p[RP_P_TAU][i];
    R.k = P.p[RP_P_K][i];
    R.eta = P.p[RP_P_ETA][i];
    if constexpr (MODEL == RP_MODEL_LI_SIGMOID) {
        R.rmax = P.p[RP_P_RMAX][i];
        R.slope = P.p[RP_P_SLOPE][i];
        R.v0 = P.p[RP_P_V0][i];
    }
    return R;
}

template <int MODEL>
__device__ __forceinline__ float activation(const RateRegs& R, float v) {
    if constexpr (MODEL == RP_MODEL_LI_TANH) {
        return rp_tanhf(v);
    } else {  // r = r_max / (1 + exp(s*(v0 - v)))   leaky_integrator.yaml:22
        float e = rp_expf(R.slope * (R.v0 - v));
        return R.rmax / (1.0f + e);
    }
}

// EDGES: `esum` = ((E_0 + E_1) + ...) of the population's dense edges joins the node input (rp_run.edge_*)
// `I_x`: the step's external input of neuron i (input_current), `v` / `s`: its state and source value BEFORE the step -- read by
// the caller (the per-step kernels: from memory right here; the resident kernel: its own values of the step before, kept in
// registers, and the input fetched while it waits for the source vector).  Returns the new source value; *v_new: the new state.
// `regs` (rate models, optional): the neuron's parameters and -v / tau already in registers
template <int MODEL, bool EDGES = false>
__device__ __forceinline__ float epilogue_core(const StepParams& P, int i, float wsum, float esum, float I_x, float v, float s,
                                               float* v_new_out, const RateRegs* regs = nullptr) {
    const int gi = P.row0 + i;
    float I = I_x;
    if constexpr (EDGES)
        if (P.n_edges > 0) I = P.x_mode != RP_X_NONE ? I + esum : esum;
    if (P.save_in_t) P.save_in_t[i] = I;
    if constexpr (MODEL == RP_MODEL_INSTANT) {  // InstantNode.forward (nodes.py:47-51): out = act(I); no state
        float o = I;
        if (P.act == RP_ACT_TANH)
            o = rp_tanhf(I);
        else if (P.act == RP_ACT_SIGMOID)
            o = 1.0f / (1.0f + rp_expf(-I));
        P.src_next[gi] = o;
        sample_output(P, i, o);
        return o;
    } else if constexpr (MODEL >= RP_MODEL_QIF) {
        // SpikeResetNet.forward (nodes.py:382-392) with the field of the spiking template:
        //   QIF      qif.yaml:11-12     v' = (v^2 + eta + I)/tau + k s_in;          s' = -s/tau_s + spike
        //   QIF_SFA  qif.yaml:25-35     eta -> eta - x;                             x' = -x/tau_x + alpha spike
        //   LIF      lif.yaml:10-16     v' = -v/tau + k s_in + I + eta;             s' = -s/tau_s + spike + s_ext
        //   IK       ik.yaml:8-12       v' = (k(v-v_r)(v-v_theta) - u + I + eta + g s_in (E_r - v))/C
        //                               u' = (b(v-v_r) - u)/tau_u + kappa spike;    s' = -s/tau_s + spike
        const float v_det = P.spike_src ? P.spike_src[i] : v;     // MultiSpikeResetNet's detector copy (nodes.py:451-453)
        const float spike = (v_det - P.theta >= 0.0f) ? 1.0f : 0.0f;  // heaviside(v-theta, 1): equality spikes
        const float spike_in = spike * P.inv_dt;                   // nodes.py:385
        float dv, aux = 0.f, aux_new = 0.f, aux2 = 0.f, aux2_new = 0.f;
        float ds = -s / P.p[RP_P_TAU_S][i] + spike_in;
        if constexpr (MODEL == RP_MODEL_QIF) {
            dv = ((v * v + P.p[RP_P_ETA][i]) + I) / P.p[RP_P_TAU][i] + P.p[RP_P_K][i] * wsum;
        } else if constexpr (MODEL == RP_MODEL_QIF_SFA) {
            aux = P.y[RP_Y_AUX][i];
            dv = ((v * v + (P.p[RP_P_ETA][i] - aux)) + I) / P.p[RP_P_TAU][i] + P.p[RP_P_K][i] * wsum;
            aux_new = aux + P.dt * (-aux / P.p[RP_P_TAU_X][i] + P.p[RP_P_ALPHA][i] * spike_in);
        } else if constexpr (MODEL == RP_MODEL_LIF) {
            const float Iv = P.x_target == RP_XT_MEMBRANE ? I : 0.f;
            const float Is = P.x_target == RP_XT_SYNAPSE ? I : 0.f;
            dv = ((-v / P.p[RP_P_TAU][i] + P.p[RP_P_K][i] * wsum) + Iv) + P.p[RP_P_ETA][i];
            ds = ds + Is;
        } else {  // IK family (ik.yaml:8-68)
            aux = P.y[RP_Y_AUX][i];
            const float dvr = v - P.p[RP_P_V_R][i];
            dv = (((((P.p[RP_P_K][i] * dvr) * (v - P.p[RP_P_V_THETA][i]) - aux) + I) + P.p[RP_P_ETA][i]) +
                  (P.p[RP_P_G][i] * wsum) * (P.p[RP_P_E_R][i] - v)) / P.p[RP_P_IK_C][i];
            if constexpr (MODEL == RP_MODEL_IK) {
                aux_new = aux + P.dt * ((P.p[RP_P_B][i] * dvr - aux) / P.p[RP_P_TAU_U][i] + P.p[RP_P_KAPPA][i] * spike_in);
            } else {  // iku_op / ik_biexp_op: recovery driven by the population means (ik.yaml:34-46)
                const float mean_v = P.scratch[0], mean_spike = P.scratch[1];
                aux_new = aux + P.dt * ((P.p[RP_P_B][i] * (mean_v - P.p[RP_P_V_R][i]) - aux) / P.p[RP_P_TAU_U][i] +
                                        P.p[RP_P_KAPPA][i] * mean_spike);
                if constexpr (MODEL == RP_MODEL_IK_BIEXP) {  // s' = -s/tau_d + x ; x' = -x/tau_r + spike
                    aux2 = P.y[RP_Y_AUX2][i];
                    ds = -s / P.p[RP_P_TAU_S][i] + aux2;
                    aux2_new = aux2 + P.dt * (-aux2 / P.p[RP_P_TAU_R][i] + spike_in);
                }
            }
        }
        float v_new = v + P.dt * dv;
        const float s_new = s + P.dt * ds;
        v_new = v_new * (1.0f - spike) + spike * P.v_reset;
        P.y[RP_Y_V][i] = v_new;
        P.src_next[gi] = s_new;
        if (P.hist_next) P.hist_next[i] = P.out_var == RP_OUT_STATE0 ? v_new : s_new;
        if constexpr (MODEL == RP_MODEL_QIF_SFA || MODEL >= RP_MODEL_IK) {
            P.y[RP_Y_AUX][i] = aux_new;
            if (P.save_state_t[RP_Y_AUX]) P.save_state_t[RP_Y_AUX][i] = aux;
        }
        if constexpr (MODEL == RP_MODEL_IK_BIEXP) {
            P.y[RP_Y_AUX2][i] = aux2_new;
            if (P.save_state_t[RP_Y_AUX2]) P.save_state_t[RP_Y_AUX2][i] = aux2;
        }
        if (P.save_state_t[0]) P.save_state_t[0][i] = v;
        if (P.save_src_t) P.save_src_t[i] = s;
        if (P.save_acc_t) P.save_acc_t[i] = wsum;
        sample_output(P, i, P.out_var == RP_OUT_STATE0 ? v : s);
        if (P.flags & FLAG_SAMPLE) {  // pre-update state (nodes.py:387-392: .y is never advanced)
            if (P.rec_row[RP_REC_V]) P.rec_row[RP_REC_V][i] = v;
            if (P.rec_row[RP_REC_SOURCE]) P.rec_row[RP_REC_SOURCE][i] = s;
            if (P.rec_row[RP_REC_AUX]) P.rec_row[RP_REC_AUX][i] = aux;
            if (P.rec_row[RP_REC_AUX2]) P.rec_row[RP_REC_AUX2][i] = aux2;
        }
        if (spike != 0.0f && P.spike_count) {
            int slot = atomicAdd(P.spike_count, 1);
            if (slot < P.spike_capacity) {
                P.spike_events[2 * slot] = P.t_abs;
                P.spike_events[2 * slot + 1] = gi;
            }
        }
        *v_new_out = v_new;
        return s_new;
    } else {
        // RateNet.forward, nodes.py:166-170; field leaky_integrator.yaml:10
        RateRegs R;
        if (regs) {
            R = *regs;
        } else {
            R = rate_regs_load<MODEL>(P, i);
            R.mv_tau = -v / R.tau;
        }
        const float dv = ((R.mv_tau + R.k * wsum) + I) + R.eta;
        const float v_new = v + P.dt * dv;
        const float r_new = activation<MODEL>(R, v_new);
        P.y[RP_Y_V][i] = v_new;
        P.src_next[gi] = r_new;
        if (P.hist_next) P.hist_next[i] = P.out_var == RP_OUT_STATE0 ? v_new : r_new;
        if (P.save_state_t[0]) P.save_state_t[0][i] = v;
        if (P.save_src_t) P.save_src_t[i] = s;
        if (P.save_acc_t) P.save_acc_t[i] = wsum;
        sample_output(P, i, P.out_var == RP_OUT_STATE0 ? v : s);
        if (P.flags & FLAG_SAMPLE) {  // post-update values (network.py:147-148)
            if (P.rec_row[0]) P.rec_row[0][i] = v_new;
            if (P.rec_row[1]) P.rec_row[1][i] = r_new;
        }
        *v_new_out = v_new;
        return r_new;
    }
}

template <int MODEL, bool EDGES = false>
__device__ __forceinline__ void epilogue(const StepParams& P, int i, float wsum, float esum = 0.f) {
    if constexpr (MODEL == kModelFilter) {  // rp_edge_mem: y'_i = (F y)_i + x_i  (edges.py:119, 144)
        P.src_next[i] = P.x_mode != RP_X_NONE ? wsum + P.x_t[i] : wsum;
    } else {
        float v = 0.f, s = 0.f, v_new;
        if constexpr (MODEL != RP_MODEL_INSTANT) {
            v = P.y[RP_Y_V][i];
            s = P.src_cur[P.row0 + i];
        }
        (void)epilogue_core<MODEL, EDGES>(P, i, wsum, esum, input_current(P, i), v, s, &v_new);
    }
}

template <int MODEL, int R, int U, bool VEC4, bool NT>
__device__ __forceinline__ void step_body(const StepParams& P) {
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int groups = (P.rows + R - 1) / R;
    for (int g = blockIdx.x * kWavesPerBlock + wave; g < groups; g += gridDim.x * kWavesPerBlock) {
        const int row_base = g * R;
        float sum[R];
        if constexpr (VEC4)
            dot_rows_v4<R, U, NT>(P, row_base, lane, sum);
        else
            dot_rows_s1<R>(P, row_base, lane, sum);
        float mine = 0.f;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (lane == j) mine = sum[j];
        if (lane < R && row_base + lane < P.rows) epilogue<MODEL>(P, row_base + lane, mine);
    }
}

template <int MODEL, int R, int U, bool VEC4, bool NT>
__global__ __launch_bounds__(kThreads) void step_kernel(const StepParams P) {
    step_body<MODEL, R, U, VEC4, NT>(P);
}

// ---- launch-bound sizes: ALL steps of a call in ONE launch (r05; the machinery: rp_resident.h) --------------------------------
// At n <= 1024 a step is bounded by the dependent-launch interval of the stream (about 4 us; W is 1 MiB at n = 512), not by
// bytes.  Every wave keeps its rows of W in registers, the source vector goes from wave to wave as tagged granules, the row dot
// products visit the chunks in the contract's order and reduce through the same butterflies (rp_dot.h), the epilogue is the
// per-step kernel's (its lane keeps the neuron's state and source value in registers and fetches the step's input while it
// waits for the vector): bit-identical to a launch per step.  (r04 tried a grid barrier per step -- arrive, poll, then
// agent-scope loads: four to five dependent trips through the fabric -- and lost against the launches: 4.95-6.0 us.)
struct ResidentArgs {
    rp_resident_ctl C;
    int steps, cur0;
    float* src[2];
    const float* x;
    int64_t ldx;
    float* traj;
    int64_t ldtraj;
    float* hist[3];
    float* save_state[RP_MAX_STATE];
    float* save_src;
    float* save_acc;
    float* save_in;
    int64_t ldsave;
    int64_t step0, cutoff;
    int sampling_steps, acc_count0, sample_idx0, has_acc;
    float* samples;
    int64_t ldsamples;
    float* rec[RP_MAX_STATE];
};

// set_step on the device (the same bookkeeping, step by step)
__device__ __forceinline__ void resident_set_step(StepParams& P, const ResidentArgs& A, int t, int& acc_count, int& sample_idx) {
    const int64_t t_abs = A.step0 + t;
    const int cur = A.cur0 ^ (t & 1);
    P.src_cur = A.src[cur];
    P.src_next = A.src[cur ^ 1];
    P.x_t = A.x ? A.x + (int64_t)t * A.ldx : nullptr;
    P.traj_t = A.traj ? A.traj + (int64_t)t * A.ldtraj : nullptr;
    P.hist_t = A.hist[0] ? A.hist[t_abs % 3] : nullptr;
    P.hist_next = nullptr;
#pragma unroll
    for (int q = 0; q < RP_MAX_STATE; ++q) P.save_state_t[q] = A.save_state[q] ? A.save_state[q] + (int64_t)t * A.ldsave : nullptr;
    P.save_src_t = A.save_src ? A.save_src + (int64_t)t * A.ldsave : nullptr;
    P.save_acc_t = A.save_acc ? A.save_acc + (int64_t)t * A.ldsave : nullptr;
    P.save_in_t = A.save_in ? A.save_in + (int64_t)t * A.ldsave : nullptr;
    P.t_abs = (int)t_abs;
    P.flags = 0;
    P.sample_row = nullptr;
#pragma unroll
    for (int q = 0; q < RP_MAX_STATE; ++q) P.rec_row[q] = nullptr;
    if (A.has_acc && t_abs >= A.cutoff) {
        P.flags |= FLAG_BUFFER;
        if (acc_count == 0) P.flags |= FLAG_FIRST;
        acc_count += 1;
        if (t_abs % A.sampling_steps == 0) {
            P.flags |= FLAG_SAMPLE;
            P.count = (float)acc_count;
            const int64_t off = (int64_t)sample_idx * A.ldsamples;
            P.sample_row = A.samples ? A.samples + off : nullptr;
#pragma unroll
            for (int q = 0; q < RP_MAX_STATE; ++q) P.rec_row[q] = A.rec[q] ? A.rec[q] + off : nullptr;
            sample_idx += 1;
            acc_count = 0;
        }
    }
}

template <int MODEL, class L>
__global__ __launch_bounds__(kThreads) void resident_kernel(const StepParams P0, const ResidentArgs A) {
    static_assert(kWavesPerBlock == kResidentWaves, "rp_resident.h counts four waves per workgroup");
    constexpr int RW = L::RW;
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int block = rp_resident_join(A.C);
    if (block < 0) return;
    const int row_w0 = (block * kWavesPerBlock + (threadIdx.x >> 6)) * RW;   // this wave's rows: [row_w0, row_w0 + RW)
    StepParams P = P0;
    typename L::Rows rows;
    L::load_rows(rows, P.W, P.ldw, row_w0, lane, P.n);
    int acc_count = A.acc_count0, sample_idx = A.sample_idx0;
    const int my_row = row_w0 + rp_rows_allreduce_row<RW>(lane);
    const bool epi_lane = (lane & (64 / RW - 1)) == 0 && my_row < P.n;   // (the lanes of a group hold the same row total after the butterfly)
    // the epilogue lane's own state and source value stay in registers from step to step
    float v_reg = 0.f, s_reg = 0.f;
    RateRegs R = {};
    constexpr bool kRate = MODEL == RP_MODEL_LI_TANH || MODEL == RP_MODEL_LI_SIGMOID;
    if (epi_lane) {
        v_reg = P0.y[RP_Y_V][my_row];
        s_reg = A.src[A.cur0][P0.row0 + my_row];
        if constexpr (kRate) R = rate_regs_load<MODEL>(P0, my_row);
    }
    for (int t = 0; t < A.steps; ++t) {
        resident_set_step(P, A, t, acc_count, sample_idx);
        P.spike_src = t == 0 ? P0.spike_src : nullptr;
        const float I_x = epi_lane ? input_current(P, my_row) : 0.f;   // (requested before the wait for the source vector)
        if constexpr (kRate) R.mv_tau = -v_reg / R.tau;
        typename L::Vec r;
        if (t == 0) {
            L::load_vec(r, P.src_cur, lane, P.n);
        } else if (!L::sweep(A.C, (t - 1) & 1, (unsigned)t, r, lane, P.n)) {
            return;
        }
        const float total = L::dots(rows, r, P.row0 + row_w0, lane);
        if (epi_lane) {
            float v_new;
            s_reg = epilogue_core<MODEL>(P, my_row, total, 0.f, I_x, v_reg, s_reg, &v_new, kRate ? &R : nullptr);
            v_reg = v_new;
            if (t + 1 < A.steps) rp_resident_publish(A.C, t & 1, my_row, (unsigned)(t + 1), s_reg);
        }
    }
}

// Step of a population with dense edges from other vectors (rp_run.edge_*, rp_forward_coupled): the wave that owns a
// row group streams its rows of W and then its rows of every edge matrix -- each a row dot product with the recurrent
// one's summation tree (rp_dot.h) on the edge's own width --, adds the edge sums in edge order and hands them to the
// epilogue as part of the node input.  One launch per population and Euler step whatever the number of edges.
template <int MODEL, int R, int U, bool VEC4>
__global__ __launch_bounds__(kThreads) void step_edges_kernel(const StepParams P) {
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int groups = (P.rows + R - 1) / R;
    for (int g = blockIdx.x * kWavesPerBlock + wave; g < groups; g += gridDim.x * kWavesPerBlock) {
        const int row_base = g * R;
        float sum[R], esum[R];
#pragma unroll
        for (int j = 0; j < R; ++j) esum[j] = 0.f;
        if constexpr (MODEL == RP_MODEL_INSTANT) {  // no recurrent matrix
#pragma unroll
            for (int j = 0; j < R; ++j) sum[j] = 0.f;
        } else if constexpr (VEC4)
            dot_rows_v4<R, U, true>(P, row_base, lane, sum);
        else
            dot_rows_s1<R>(P, row_base, lane, sum);
        for (int e = 0; e < P.n_edges; ++e) {
            // Edge e's descriptor straight from the kernel-argument segment (a scalar load at a runtime offset): indexing the
            // by-value block `P` with a runtime e would copy it to scratch, selecting among four copies holds 32 more SGPRs.
            typedef const __attribute__((address_space(4))) char* karg_ptr;
            typedef const __attribute__((address_space(4))) EdgeArg* edge_ptr;
            const edge_ptr Ep = (edge_ptr)((karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(StepParams, edge)) + e;
            const EdgeArg E{Ep->W, Ep->ld, Ep->src, Ep->n, Ep->vec4};
            float es[R];
            if (E.W == nullptr) {  // the identity edge: E_e[i] = s_e[i]
#pragma unroll
                for (int j = 0; j < R; ++j) es[j] = row_base + j < P.rows ? E.src[row_base + j] : 0.f;
            } else if (E.vec4)
                rp_dot_rows_v4<R, U, true>(RowDotArgs{E.W, E.ld, E.n, P.rows, P.row0, E.src}, row_base, lane, es);
            else
                rp_dot_rows_s1<R>(RowDotArgs{E.W, E.ld, E.n, P.rows, P.row0, E.src}, row_base, lane, es);
#pragma unroll
            for (int j = 0; j < R; ++j) esum[j] = e == 0 ? es[j] : esum[j] + es[j];
        }
        float mine = 0.f, mine_e = 0.f;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (lane == j) {
                mine = sum[j];
                mine_e = esum[j];
            }
        if (lane < R && row_base + lane < P.rows) epilogue<MODEL, true>(P, row_base + lane, mine, mine_e);
    }
}

// ---- all members of a coupled circuit in ONE launch (rp_forward_coupled, launch-bound sizes) ---------------------------
// At a few hundred to a few thousand neurons per population a step is bounded by the dependent-launch interval of the
// stream, not by bytes: one launch per Euler step for the whole circuit instead of one per member.  Legal when every edge
// reads a value from before the step (coupled_impl decides); state-variable outputs are then read from the members' hist
// rings, never from a state vector this launch updates.  A workgroup belongs to one member (first_block); the member's
// argument block is read in place, at a runtime offset into the kernel arguments.
constexpr int kMergeMax = 4;
constexpr int kMergeMaxEdges = 4;  // (edges into one member of the shared launch: merged_rows selects among four descriptors)
struct MergedParams {
    int count;
    int first_block[kMergeMax + 1];
    StepParams P[kMergeMax];
};
static_assert(sizeof(MergedParams) <= 4000, "the argument block of step_merged_kernel must fit the kernel-argument segment");

template <int MODEL, int R, int U>
__device__ __forceinline__ void merged_rows(const StepParams& P, int block) {
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int groups = (P.rows + R - 1) / R;
    const int g = block * kWavesPerBlock + wave;
    if (g >= groups) return;
    const int row_base = g * R;
    float sum[R], esum[R];
#pragma unroll
    for (int j = 0; j < R; ++j) esum[j] = 0.f;
    if (P.vec4)
        rp_dot_rows_v4_ring<R, U, true>(dot_args(P), row_base, lane, sum);
    else
        dot_rows_s1<R>(P, row_base, lane, sum);
    for (int e = 0; e < P.n_edges; ++e) {
        // (field by field: a select between whole EdgeArg objects keeps the argument block in memory)
#define RP_EDGE_FIELD(f) (e == 0 ? P.edge[0].f : (e == 1 ? P.edge[1].f : (e == 2 ? P.edge[2].f : P.edge[3].f)))
        const float* eW = RP_EDGE_FIELD(W);
        const float* eS = RP_EDGE_FIELD(src);
        const int64_t eLd = RP_EDGE_FIELD(ld);
        const int eN = RP_EDGE_FIELD(n), eV4 = RP_EDGE_FIELD(vec4);
#undef RP_EDGE_FIELD
        float es[R];
        if (eV4)
            rp_dot_rows_v4_ring<R, U, true>(RowDotArgs{eW, eLd, eN, P.rows, P.row0, eS}, row_base, lane, es);
        else
            rp_dot_rows_s1<R>(RowDotArgs{eW, eLd, eN, P.rows, P.row0, eS}, row_base, lane, es);
#pragma unroll
        for (int j = 0; j < R; ++j) esum[j] = e == 0 ? es[j] : esum[j] + es[j];
    }
    float mine = 0.f, mine_e = 0.f;
#pragma unroll
    for (int j = 0; j < R; ++j)
        if (lane == j) {
            mine = sum[j];
            mine_e = esum[j];
        }
    if (lane < R && row_base + lane < P.rows) epilogue<MODEL, true>(P, row_base + lane, mine, mine_e);
}

template <int R, int U>
__global__ __launch_bounds__(kThreads) void step_merged_kernel(const MergedParams M) {
    int m = 0;
    while (m + 1 < M.count && (int)blockIdx.x >= M.first_block[m + 1]) ++m;
    // the member's argument block, copied out of the kernel-argument segment (uniform address: scalar loads of the fields
    // that are used; indexing the by-value argument M.P[m] with a runtime m would move all of M to scratch)
    typedef const __attribute__((address_space(4))) char* karg_ptr;
    StepParams P;
    __builtin_memcpy(&P, (const void*)((karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MergedParams, P) +
                                       (size_t)m * sizeof(StepParams)), sizeof(StepParams));
    const int block = (int)blockIdx.x - M.first_block[m];
    switch (P.model) {  // (population-mean models are never merged: they need their reduction launch in front)
        case RP_MODEL_LI_TANH: merged_rows<RP_MODEL_LI_TANH, R, U>(P, block); break;
        case RP_MODEL_LI_SIGMOID: merged_rows<RP_MODEL_LI_SIGMOID, R, U>(P, block); break;
        case RP_MODEL_QIF: merged_rows<RP_MODEL_QIF, R, U>(P, block); break;
        case RP_MODEL_QIF_SFA: merged_rows<RP_MODEL_QIF_SFA, R, U>(P, block); break;
        case RP_MODEL_LIF: merged_rows<RP_MODEL_LIF, R, U>(P, block); break;
        default: merged_rows<RP_MODEL_IK, R, U>(P, block); break;
    }
}

// ---- ... and ALL steps of a coupled circuit in ONE launch (r05: the resident form of the shared launch; machinery: rp_resident.h) ---
// The shared launch above is bounded by the host's launch rate (5.6 us per step; a population alone runs resident at 2-3 us).
// Here a wave keeps its rows of the member's own matrix AND of the member's in-edge matrices in registers (slots of NC chunks;
// NSEG slots, NSEG x NC <= 4 chunks), every member hands its source vector -- and its state vector, when an edge reads that --
// from wave to wave as tagged granules (one stream per member and vector, two parities), and every wave waits for the source
// stream of EVERY member before it steps (the circuit moves in lockstep: a member whose output nobody in a cycle reads cannot
// run ahead and overwrite granules its readers have not consumed).  A lagging edge (RP_EDGE_LAG) multiplies with the vector
// the wave received one step earlier.  The row dots are each matrix's own contract, the edge sums are added in edge order, the
// epilogue is the per-step kernels': bit-identical to the shared launch.
constexpr int kRcCap = 1024;      // granules per stream and parity (members of up to 1024 neurons)
constexpr int kRcSlots = 4;
struct RcMember {
    StepParams P0;                // the member's step-0 arguments (set_step(.., 0))
    ResidentArgs A;               // (A.C unused: the launch's control block is a kernel argument)
    int first_block, nslots, publish_v, model;
    int slot_src[kRcSlots];       // slot 0: the member itself; slot k: the source member of edge k - 1
    int slot_state[kRcSlots];     // the slot reads the source's state vector (its output variable), not its source vector
    int slot_lag[kRcSlots];       // RP_EDGE_LAG: the vector of one step earlier
    int slot_n[kRcSlots];         // columns of the slot's matrix
    const float* slot_w[kRcSlots];
    int64_t slot_ld[kRcSlots];
    const float* slot_vec0[kRcSlots];   // the slot's vector at the call's first step, as it lies in memory
    const float* slot_lag0[kRcSlots];   // ... and, for a lagging slot, the one before it
    int all_rows[kMergeMax];            // neurons of every member of the circuit (the lockstep sweeps)
};
struct RcParams {
    int count;
    RcMember M[kMergeMax];
};

__device__ __forceinline__ unsigned long long* rc_stream(const rp_resident_ctl& C, int member, int state, int parity) {
    return C.gran[0] + (size_t)((member * 2 + state) * 2 + parity) * kRcCap;
}

template <int MODEL, int NSEG, int NC, int RW>
__device__ __forceinline__ void resident_coupled_member(const rp_resident_ctl& C, const RcMember& M, int m, int count, int block) {
    typedef rp_layout_v4<NC, RW, false> L;   // (the vector's lane layout and the sweep; the dots go through rp_resident_dots_slot)
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const ResidentArgs& A = M.A;
    const int row_w0 = (block * kWavesPerBlock + (threadIdx.x >> 6)) * RW;
    StepParams P = M.P0;
    rp_f4 w[NSEG][RW][NC];
#pragma unroll
    for (int k = 0; k < NSEG; ++k) {
        if (k < M.nslots)
            rp_resident_load_rows_rect<NC, RW>(w[k], M.slot_w[k], M.slot_ld[k], row_w0, lane, P.rows, M.slot_n[k]);
        else
#pragma unroll
            for (int j = 0; j < RW; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) w[k][j][c] = rp_f4{0.f, 0.f, 0.f, 0.f};
    }
    // the members whose source stream no slot of this wave sweeps: their tags are waited for all the same (lockstep)
    unsigned swept = 0u;
#pragma unroll
    for (int k = 0; k < NSEG; ++k)
        if (k < M.nslots && !M.slot_state[k]) swept |= 1u << M.slot_src[k];
    int acc_count = A.acc_count0, sample_idx = A.sample_idx0;
    const int my_row = row_w0 + rp_rows_allreduce_row<RW>(lane);
    const bool epi_lane = (lane & (64 / RW - 1)) == 0 && my_row < P.rows;
    float v_reg = 0.f, s_reg = 0.f;
    RateRegs R = {};
    constexpr bool kRate = MODEL == RP_MODEL_LI_TANH || MODEL == RP_MODEL_LI_SIGMOID;
    if (epi_lane) {
        v_reg = M.P0.y[RP_Y_V][my_row];
        s_reg = A.src[A.cur0][my_row];
        if constexpr (kRate) R = rate_regs_load<MODEL>(M.P0, my_row);
    }
    typename L::Vec prev[NSEG];   // lagging slots: the vector of the step before
#pragma unroll
    for (int k = 1; k < NSEG; ++k)
        if (k < M.nslots && M.slot_lag[k]) L::load_vec(prev[k], M.slot_lag0[k], lane, M.slot_n[k]);
    rp_resident_ctl Cs = C;   // (the sweep and the publication take the stream's granules as "parity 0")
    for (int t = 0; t < A.steps; ++t) {
        resident_set_step(P, A, t, acc_count, sample_idx);
        P.hist_t = nullptr;   // as in the shared launch: the ring is written one step ahead, nothing reads this step's entry late
        P.hist_next = A.hist[0] ? A.hist[(A.step0 + t + 1) % 3] : nullptr;
        P.spike_src = nullptr;
        const float I_x = epi_lane ? input_current(P, my_row) : 0.f;
        if constexpr (kRate) R.mv_tau = -v_reg / R.tau;
        float wsum = 0.f, esum = 0.f;
#pragma unroll
        for (int k = 0; k < NSEG; ++k) {
            if (k >= M.nslots) continue;   // (wave-uniform)
            typename L::Vec cur;
            if (t == 0) {
                L::load_vec(cur, M.slot_vec0[k], lane, M.slot_n[k]);
            } else {
                Cs.gran[0] = rc_stream(C, M.slot_src[k], M.slot_state[k], (t - 1) & 1);
                if (!L::sweep(Cs, 0, (unsigned)t, cur, lane, M.slot_n[k])) return;
            }
            float total;
            if (k > 0 && M.slot_lag[k]) {
                total = rp_resident_dots_slot<NC, RW>(w[k], prev[k].c, M.slot_n[k], P.row0 + row_w0, lane);
                prev[k] = cur;
            } else {
                total = rp_resident_dots_slot<NC, RW>(w[k], cur.c, M.slot_n[k], P.row0 + row_w0, lane);
            }
            if (k == 0)
                wsum = total;
            else
                esum = k == 1 ? total : esum + total;
        }
        if (t > 0)
            for (int q = 0; q < count; ++q) {   // the other members' progress
                if ((swept >> q) & 1u) continue;
                typename L::Vec seen;
                Cs.gran[0] = rc_stream(C, q, 0, (t - 1) & 1);
                if (!L::sweep(Cs, 0, (unsigned)t, seen, lane, M.all_rows[q])) return;
            }
        if (epi_lane) {
            float v_new;
            s_reg = epilogue_core<MODEL, true>(P, my_row, wsum, esum, I_x, v_reg, s_reg, &v_new, kRate ? &R : nullptr);
            v_reg = v_new;
            if (t + 1 < A.steps) {
                if (M.publish_v) {
                    Cs.gran[0] = rc_stream(C, m, 1, t & 1);
                    rp_resident_publish(Cs, 0, my_row, (unsigned)(t + 1), v_new);
                }
                Cs.gran[0] = rc_stream(C, m, 0, t & 1);
                rp_resident_publish(Cs, 0, my_row, (unsigned)(t + 1), s_reg);
            }
        }
    }
}

template <int NSEG, int NC, int RW>
__global__ __launch_bounds__(kThreads) void resident_coupled_kernel(const rp_resident_ctl C, const RcParams* __restrict__ params) {
    static_assert(kWavesPerBlock == kResidentWaves, "rp_resident.h counts four waves per workgroup");
    static_assert(NSEG * NC * RW <= 32 && NSEG <= kRcSlots && NC <= 4, "a wave's rows: at most 128 registers");
    const int work = rp_resident_join(C);
    if (work < 0) return;
    // the member this workgroup belongs to; its argument block is read in place (read-only, uniform addresses: scalar loads)
    const int count = params->count;
    int m = 0;
    while (m + 1 < count && work >= params->M[m + 1].first_block) ++m;
    const RcMember& M = params->M[m];
    const int block = work - M.first_block;
    switch (M.model) {
        case RP_MODEL_LI_TANH: resident_coupled_member<RP_MODEL_LI_TANH, NSEG, NC, RW>(C, M, m, count, block); break;
        case RP_MODEL_LI_SIGMOID: resident_coupled_member<RP_MODEL_LI_SIGMOID, NSEG, NC, RW>(C, M, m, count, block); break;
        case RP_MODEL_QIF: resident_coupled_member<RP_MODEL_QIF, NSEG, NC, RW>(C, M, m, count, block); break;
        case RP_MODEL_QIF_SFA: resident_coupled_member<RP_MODEL_QIF_SFA, NSEG, NC, RW>(C, M, m, count, block); break;
        case RP_MODEL_LIF: resident_coupled_member<RP_MODEL_LIF, NSEG, NC, RW>(C, M, m, count, block); break;
        default: resident_coupled_member<RP_MODEL_IK, NSEG, NC, RW>(C, M, m, count, block); break;
    }
}

// The same code under its own name: the launches rp_tune_forward times on its scratch population, so that a kernel
// trace keeps them apart from the steps of a run.
template <int R, int U>
__global__ __launch_bounds__(kThreads) void step_kernel_tuning_probe(const StepParams P) {
    step_body<RP_MODEL_LI_TANH, R, U, true, true>(P);
}

// ---- packed-W step (rp_net.sp_*) --------------------------------------------------------------------------------
// An edge matrix of a coupled member in packed form (rp_run.edge_sp): the sparse kernels below with this pseudo-model form
// the edge's row sums -- the bits of the dense row dot, as for a packed W -- and chain them in edge order,
// E = (E_0 + E_1) + ..., the last one adding the member's own input: I = x + E (the epilogue's order).
__device__ __forceinline__ void edge_sum_result(const StepParams& P, int i, float wsum) {
    float e = P.esum_prev ? P.esum_prev[i] + wsum : wsum;
    if (P.esum_final && P.x_mode != RP_X_NONE) e = input_current(P, i) + e;
    P.esum_out[i] = e;
}

template <int MODEL>
__device__ __forceinline__ void sparse_row_result(const StepParams& P, int row, float sum) {
    if constexpr (MODEL == kModelEdgeSum)
        edge_sum_result(P, row, sum);
    else
        epilogue<MODEL>(P, row, sum);
}

// Same arithmetic as step_kernel's 16-byte path, streaming only the stored nonzeros: lane l still owns the slots
// 4l..4l+3 of a row and keeps their four accumulators; its nonzeros come as ONE list in the order the dense kernel
// meets them (value + code = component << 14 | column chunk), four list entries per lane per group, so every
// accumulator sees the same fmaf sequence and the result is bit-identical.  One row per wave at a time, U groups
// (1.5 KiB each) in flight.  The source vector is gathered per entry: from LDS when it fits (n <= 32768: one
// 1024-thread workgroup per CU keeps all of it, stored component-major inside every 256-column chunk so that the
// 64 lanes of a gather hit 64 different banks whatever their chunks and components), else from L2.
template <int MODEL, bool LDS_SRC, int R, int U>
__global__ __launch_bounds__(kSparseThreads) void step_sparse_kernel(const StepParams P) {
    extern __shared__ float lds_src[];
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    constexpr int kWaves = kSparseThreads / RP_WAVE;
    if constexpr (LDS_SRC) rp_sparse_stage_source(lds_src, P.src_cur, P.n, (P.n + 255) & ~255);
    // LDS: the position of this lane's columns inside a staged chunk; global: first of its four columns
    const int lane_base = LDS_SRC ? lane : 4 * lane;  // (LDS: the lane's position inside a staged chunk, rp_sparse_stage_source)
    const int lane_rot = 0;
    const float* xs = LDS_SRC ? lds_src : P.src_cur;
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    // R consecutive rows per wave, interleaved (R * U groups in flight, R independent accumulator sets); row lengths
    // are wave-uniform, so the guards do not diverge.  Production uses R = 1.
    const int groups = (P.rows + R - 1) / R;
    for (int g = blockIdx.x * kWaves + wave; g < groups; g += gridDim.x * kWaves) {
        const int row_base = g * R;
        const float* vp[R];
        const uint16_t* ip[R];
        int len[R];
        int nmax = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int row = row_base + j < P.rows ? row_base + j : P.rows - 1;
            const int g0 = P.sp_rowoff[row];
            len[j] = row_base + j < P.rows ? P.sp_rowoff[row + 1] - g0 : 0;
            vp[j] = P.sp_vals + (int64_t)g0 * 256 + 4 * lane;
            ip[j] = P.sp_cidx + (int64_t)g0 * 256 + 4 * lane;
            nmax = len[j] > nmax ? len[j] : nmax;
        }
        rp_f4 acc[R];
#pragma unroll
        for (int j = 0; j < R; ++j) acc[j] = rp_f4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nmax; k += U) {
            rp_f4 v[R][U];
            us4 m[R][U];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < R; ++j)
                    if (k + u < len[j]) {
                        v[j][u] = rp_ld4_stream(vp[j] + (int64_t)(k + u) * 256);
                        m[j][u] = __builtin_nontemporal_load(reinterpret_cast<const us4*>(ip[j] + (int64_t)(k + u) * 256));
                    }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < R; ++j)
                    if (k + u < len[j]) {
                        sparse_entry<LDS_SRC>(v[j][u].x, m[j][u].x, xs, lane_base, lane_rot, acc[j]);
                        sparse_entry<LDS_SRC>(v[j][u].y, m[j][u].y, xs, lane_base, lane_rot, acc[j]);
                        sparse_entry<LDS_SRC>(v[j][u].z, m[j][u].z, xs, lane_base, lane_rot, acc[j]);
                        sparse_entry<LDS_SRC>(v[j][u].w, m[j][u].w, xs, lane_base, lane_rot, acc[j]);
                    }
        }
        float mine = 0.f;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float sum = rp_wave_allreduce(rp_lane_sum4(acc[j]));
            if (lane == j) mine = sum;
        }
        if (lane < R && row_base + lane < P.rows) sparse_row_result<MODEL>(P, row_base + lane, mine);
    }
}

// Windowed form (n > 32768): rp_sparse_windowed_rows (rp_dot.h) with the step epilogue.
template <int MODEL, int U>
__global__ __launch_bounds__(kSparseThreads) void step_sparse_windowed_kernel(const StepParams P) {
    extern __shared__ float lds_src[];
    const SparseWindowed S{P.sp_vals, P.sp_cidx, P.sp_rowoff, P.sp_row_ids, P.sp_batch_class, P.sp_phase_window,
                           P.sp_phases, P.sp_batches};
    rp_sparse_windowed_rows<U>(S, P.src_cur, P.n, lds_src, [&](int row, float sum) { sparse_row_result<MODEL>(P, row, sum); });
}

// Blocked form (n <= 32768): rp_sparse_blocked_rows (rp_dot.h) with the step epilogue.  Dynamic LDS = the staged
// source vector + WAVES x sp_block_rows x 64 parked lane sums.
template <int MODEL, int WAVES, int U>
__global__ __launch_bounds__(WAVES * RP_WAVE) void step_sparse_blocked_kernel(const StepParams P) {
    extern __shared__ float lds_src[];
    const int n_pad = (P.n + 255) & ~255;
    rp_sparse_blocked_rows<WAVES, U>(P.sp_vals, P.sp_cidx, P.sp_rowoff, P.rows, P.sp_block_rows, lds_src, lds_src + n_pad,
                                     [&] { rp_sparse_stage_source<WAVES * RP_WAVE>(lds_src, P.src_cur, P.n, n_pad); },
                                     [&](int row, float sum) { sparse_row_result<MODEL>(P, row, sum); });
}

#ifdef RP_MEASUREMENT
// measurement build only: waves per workgroup / groups in flight of the blocked sparse step (0 = built-in)
std::atomic<int> g_blocked_waves{0}, g_blocked_U{0};
#endif

template <int MODEL, int WAVES, int U>
int launch_step_sparse_blocked_cfg(const StepParams& P, hipStream_t st, int n_cu, size_t lds) {
    if (rp_allow_dynamic_lds<&step_sparse_blocked_kernel<MODEL, WAVES, U>>(rp_lds_limit()) != RP_OK) return RP_ERR_HIP;
    const int nblocks = (P.rows + P.sp_block_rows - 1) / P.sp_block_rows;
    int blocks = (nblocks + WAVES - 1) / WAVES;
    if (blocks > n_cu) blocks = n_cu;  // one resident workgroup per CU owns the LDS copy of the source vector
    hipLaunchKernelGGL((step_sparse_blocked_kernel<MODEL, WAVES, U>), dim3(blocks), dim3(WAVES * RP_WAVE), lds, st, P);
    return RP_OK;
}

template <int MODEL>
int launch_step_sparse_blocked(const StepParams& P, hipStream_t st, int n_cu) {
    const size_t src_bytes = sizeof(float) * (size_t)((P.n + 255) & ~255);
    const size_t per_wave = sizeof(float) * RP_WAVE * (size_t)P.sp_block_rows;
    const size_t limit = rp_lds_limit();
    // 16 waves per workgroup when source + parked sums fit the LDS, else 8 (n = 32768 with 16-row blocks: 128 + 32 KiB)
    int waves = src_bytes + 16 * per_wave <= limit ? 16 : 8;
#ifdef RP_MEASUREMENT
    if (g_blocked_waves == 8 || g_blocked_waves == 16) waves = g_blocked_waves;
#endif
    const size_t lds = src_bytes + waves * per_wave;
    if (lds > limit) return RP_ERR_SHAPE;
#ifdef RP_MEASUREMENT
    if constexpr (MODEL == RP_MODEL_LI_TANH) {  // sweep (rp_set_blocked_tuning)
        if (g_blocked_U == 4) return waves == 16 ? launch_step_sparse_blocked_cfg<MODEL, 16, 4>(P, st, n_cu, lds)
                                                 : launch_step_sparse_blocked_cfg<MODEL, 8, 4>(P, st, n_cu, lds);
        if (g_blocked_U == 8) return waves == 16 ? launch_step_sparse_blocked_cfg<MODEL, 16, 8>(P, st, n_cu, lds)
                                                 : launch_step_sparse_blocked_cfg<MODEL, 8, 8>(P, st, n_cu, lds);
        if (g_blocked_U == 2) return waves == 16 ? launch_step_sparse_blocked_cfg<MODEL, 16, 2>(P, st, n_cu, lds)
                                                 : launch_step_sparse_blocked_cfg<MODEL, 8, 2>(P, st, n_cu, lds);
        if (g_blocked_U == 1) return waves == 16 ? launch_step_sparse_blocked_cfg<MODEL, 16, 1>(P, st, n_cu, lds)
                                                 : launch_step_sparse_blocked_cfg<MODEL, 8, 1>(P, st, n_cu, lds);
        if (g_blocked_U == 3) return waves == 16 ? launch_step_sparse_blocked_cfg<MODEL, 16, 3>(P, st, n_cu, lds)
                                                 : launch_step_sparse_blocked_cfg<MODEL, 8, 3>(P, st, n_cu, lds);
    }
#endif
    // groups per register set -- two sets are in flight per lane since r05 (scripts/probes/sparse_blocked_sweep.py, us per step,
    // 16 waves, n = 32768, 8-row blocks): U=1 115.9, U=2 119.2, U=3 123.1, U=4 124.3 (r02-r04, one set: U=2 117.7-121, U=1 133.9)
    return waves == 16 ? launch_step_sparse_blocked_cfg<MODEL, 16, 1>(P, st, n_cu, lds)
                       : launch_step_sparse_blocked_cfg<MODEL, 8, 8>(P, st, n_cu, lds);
}

#ifdef RP_MEASUREMENT
// measurement build only: rows per wave / groups in flight per row of the sparse step (0 = built-in)
std::atomic<int> g_sparse_R{0}, g_sparse_U{0};
#endif

template <int MODEL, int R, int U>
int launch_step_sparse_cfg(const StepParams& P, hipStream_t st, int n_cu) {
    constexpr int kWaves = kSparseThreads / RP_WAVE;
    int blocks = ((P.rows + R - 1) / R + kWaves - 1) / kWaves;
    if (P.n <= kSparseLdsMaxN) {
        if (blocks > n_cu) blocks = n_cu;  // one resident workgroup per CU owns the LDS copy of the source vector
        const size_t lds = sizeof(float) * (size_t)((P.n + 255) & ~255);
        if (rp_allow_dynamic_lds<&step_sparse_kernel<MODEL, true, R, U>>(sizeof(float) * kSparseLdsMaxN) != RP_OK)
            return RP_ERR_HIP;
        hipLaunchKernelGGL((step_sparse_kernel<MODEL, true, R, U>), dim3(blocks), dim3(kSparseThreads), lds, st, P);
    } else {
        if (blocks > 2 * n_cu) blocks = 2 * n_cu;
        hipLaunchKernelGGL((step_sparse_kernel<MODEL, false, R, U>), dim3(blocks), dim3(kSparseThreads), 0, st, P);
    }
    return RP_OK;
}

template <int MODEL>
int launch_step_sparse(const StepParams& P, hipStream_t st) {
    const int n_cu = rp_cu_count();
    if (n_cu == 0) return RP_ERR_HIP;
    if (P.sp_block_rows > 0) return launch_step_sparse_blocked<MODEL>(P, st, n_cu);
    if (P.sp_row_ids) {  // windowed form
        if (rp_allow_dynamic_lds<&step_sparse_windowed_kernel<MODEL, 2>>(sizeof(float) * kWindowChunks * 256) != RP_OK)
            return RP_ERR_HIP;
        const int blocks = P.sp_batches < n_cu ? P.sp_batches : n_cu;
        if (blocks > 0)
            hipLaunchKernelGGL((step_sparse_windowed_kernel<MODEL, 2>), dim3(blocks), dim3(kSparseThreads),
                               sizeof(float) * kWindowChunks * 256, st, P);
        return RP_OK;
    }
#ifdef RP_MEASUREMENT
    if constexpr (MODEL == RP_MODEL_LI_TANH) {  // sweep (rp_set_sparse_tuning)
        const int key = g_sparse_R * 10 + g_sparse_U;
        switch (key) {
            case 14: return launch_step_sparse_cfg<MODEL, 1, 4>(P, st, n_cu);
            case 18: return launch_step_sparse_cfg<MODEL, 1, 8>(P, st, n_cu);
            case 22: return launch_step_sparse_cfg<MODEL, 2, 2>(P, st, n_cu);
            case 28: return launch_step_sparse_cfg<MODEL, 2, 8>(P, st, n_cu);
            case 24: return launch_step_sparse_cfg<MODEL, 2, 4>(P, st, n_cu);
            case 42: return launch_step_sparse_cfg<MODEL, 4, 2>(P, st, n_cu);
            case 44: return launch_step_sparse_cfg<MODEL, 4, 4>(P, st, n_cu);
            case 81: return launch_step_sparse_cfg<MODEL, 8, 1>(P, st, n_cu);
            case 82: return launch_step_sparse_cfg<MODEL, 8, 2>(P, st, n_cu);
            default: break;
        }
    }
#endif
    // measured (scripts/archive/tune_sparse.py): one row per wave with 4 groups in flight is the fastest or tied at every size
    return launch_step_sparse_cfg<MODEL, 1, 4>(P, st, n_cu);
}

int launch_step_sparse_model(int model, const StepParams& P, hipStream_t st) {
    switch (model) {
        case RP_MODEL_LI_TANH: return launch_step_sparse<RP_MODEL_LI_TANH>(P, st);
        case RP_MODEL_LI_SIGMOID: return launch_step_sparse<RP_MODEL_LI_SIGMOID>(P, st);
        case RP_MODEL_QIF: return launch_step_sparse<RP_MODEL_QIF>(P, st);
        case RP_MODEL_QIF_SFA: return launch_step_sparse<RP_MODEL_QIF_SFA>(P, st);
        case RP_MODEL_LIF: return launch_step_sparse<RP_MODEL_LIF>(P, st);
        case RP_MODEL_IK: return launch_step_sparse<RP_MODEL_IK>(P, st);
        case RP_MODEL_IKU: return launch_step_sparse<RP_MODEL_IKU>(P, st);
        case kModelEdgeSum: return launch_step_sparse<kModelEdgeSum>(P, st);
        default: return launch_step_sparse<RP_MODEL_IK_BIEXP>(P, st);
    }
}

// Population means of the pre-update v and of the spike input (iku_op / ik_biexp_op, ik.yaml:34-46): one block,
// 1024 strided partial sums combined by a halving tree (mirrored by the oracle).
__global__ __launch_bounds__(1024) void mean_kernel(const StepParams P) {
    __shared__ float red_v[1024];
    __shared__ float red_s[1024];
    const int tid = threadIdx.x;
    float av = 0.f, as = 0.f;
    for (int j = tid; j < P.rows; j += 1024) {
        const float v = P.y[RP_Y_V][j];
        av = av + v;
        as = as + ((v - P.theta >= 0.0f) ? 1.0f : 0.0f) * P.inv_dt;
    }
    red_v[tid] = av;
    red_s[tid] = as;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if (tid < off) {
            red_v[tid] = red_v[tid] + red_v[tid + off];
            red_s[tid] = red_s[tid] + red_s[tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        P.scratch[0] = red_v[0] / (float)P.rows;
        P.scratch[1] = red_s[0] / (float)P.rows;
    }
}

template <int MODEL>
__global__ void refresh_kernel(const StepParams P) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.rows) P.src_next[P.row0 + i] = activation<MODEL>(rate_regs_load<MODEL>(P, i), P.y[RP_Y_V][i]);
}

__global__ void stream_probe_kernel(const float* __restrict__ buf, int64_t n4, float* out) {
    rp_f4 a = {0.f, 0.f, 0.f, 0.f};
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        rp_f4 v = rp_ld4_stream(buf + 4 * i);
        a += v;
    }
    float s = rp_wave_allreduce(rp_lane_sum4(a));
    if ((threadIdx.x & 63) == 0 && s == 12345.678f) out[0] = s;  // keep the loads live
}

// ---- host side ----------------------------------------------------------------

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int n_params_required(int model, int* idx, int cap) {
    int c = 0;
    idx[c++] = RP_P_TAU;
    idx[c++] = RP_P_K;
    idx[c++] = RP_P_ETA;
    if (model == RP_MODEL_LI_SIGMOID) {
        idx[c++] = RP_P_RMAX;
        idx[c++] = RP_P_SLOPE;
        idx[c++] = RP_P_V0;
    } else if (model == RP_MODEL_QIF || model == RP_MODEL_LIF) {
        idx[c++] = RP_P_TAU_S;
    } else if (model == RP_MODEL_QIF_SFA) {
        idx[c++] = RP_P_TAU_S;
        idx[c++] = RP_P_ALPHA;
        idx[c++] = RP_P_TAU_X;
    } else if (model >= RP_MODEL_IK) {
        const int ik[] = {RP_P_TAU_S, RP_P_V_R, RP_P_V_THETA, RP_P_G, RP_P_E_R, RP_P_B, RP_P_TAU_U, RP_P_KAPPA};
        for (int q : ik) idx[c++] = q;
        if (model == RP_MODEL_IK_BIEXP) idx[c++] = RP_P_TAU_R;
    }
    (void)cap;
    return c;
}

int validate_net(const rp_net* net) {
    if (!net) RP_FAIL(RP_ERR_ARG, "net is null");
    if (net->model < RP_MODEL_LI_TANH || net->model >= RP_MODEL_COUNT)
        RP_FAIL(RP_ERR_ARG, "unknown model id %d", net->model);
    if (net->model == RP_MODEL_INSTANT) {  // an activation node: no matrix, no state, no parameters -- its output ping-pong only
        if (net->n <= 0 || net->rows != net->n || net->row0 != 0) RP_FAIL(RP_ERR_SHAPE, "an activation-node member has rows == n, row0 == 0 (n=%d rows=%d row0=%d)", net->n, net->rows, net->row0);
        if (net->act != RP_ACT_IDENTITY && net->act != RP_ACT_TANH && net->act != RP_ACT_SIGMOID) RP_FAIL(RP_ERR_ARG, "bad rp_net.act %d", net->act);
        if (!net->src[0] || !net->src[1] || net->src[0] == net->src[1]) RP_FAIL(RP_ERR_ARG, "src[0], src[1] must be two distinct buffers");
        if (net->cur != 0 && net->cur != 1) RP_FAIL(RP_ERR_ARG, "cur must be 0 or 1");
        if (net->sp_vals) RP_FAIL(RP_ERR_ARG, "an activation-node member has no matrix (sp_vals)");
        return RP_OK;
    }
    if ((net->model == RP_MODEL_QIF_SFA || net->model >= RP_MODEL_IK) && !net->y[RP_Y_AUX])
        RP_FAIL(RP_ERR_ARG, "model %d needs the second state vector y[1]", net->model);
    if (net->model == RP_MODEL_IK_BIEXP && !net->y[RP_Y_AUX2])
        RP_FAIL(RP_ERR_ARG, "model %d needs the third state vector y[2]", net->model);
    if (net->model == RP_MODEL_IKU || net->model == RP_MODEL_IK_BIEXP) {
        if (!net->scratch) RP_FAIL(RP_ERR_ARG, "model %d needs rp_net.scratch", net->model);
        if (net->rows != net->n)
            RP_FAIL(RP_ERR_ARG, "population means are not available on a row shard (model %d)", net->model);
    }
    if (net->n <= 0 || net->rows <= 0 || net->row0 < 0 || (int64_t)net->row0 + net->rows > net->n)
        RP_FAIL(RP_ERR_SHAPE, "bad shard: n=%d rows=%d row0=%d", net->n, net->rows, net->row0);
    if (net->ldw < net->n) RP_FAIL(RP_ERR_SHAPE, "ldw=%lld < n=%d", (long long)net->ldw, net->n);
    if (net->row0 % 8 != 0) RP_FAIL(RP_ERR_ARG, "row0=%d must be a multiple of 8 (rows are processed in aligned groups)", net->row0);
    if (!net->W || !net->y[RP_Y_V] || !net->src[0] || !net->src[1])
        RP_FAIL(RP_ERR_ARG, "W, y[0], src[0], src[1] must be non-null");
    if (net->src[0] == net->src[1]) RP_FAIL(RP_ERR_ARG, "src[0] and src[1] must be distinct buffers");
    if (net->cur != 0 && net->cur != 1) RP_FAIL(RP_ERR_ARG, "cur must be 0 or 1");
    int idx[RP_MAX_PARAMS];
    int c = n_params_required(net->model, idx, RP_MAX_PARAMS);
    for (int q = 0; q < c; ++q)
        if (!net->p[idx[q]]) RP_FAIL(RP_ERR_ARG, "parameter slot %d is null for model %d", idx[q], net->model);
    if (!(net->dt > 0.f)) RP_FAIL(RP_ERR_ARG, "dt must be > 0");
    return RP_OK;
}

bool use_vec4(const rp_net* net) { return (net->n % 4 == 0) && (net->ldw % 4 == 0); }

void fill_static(StepParams& P, const rp_net* net) {
    P.W = net->W;
    P.ldw = net->ldw;
    P.n = net->n;
    P.rows = net->rows;
    P.row0 = net->row0;
    for (int q = 0; q < RP_MAX_PARAMS; ++q) P.p[q] = net->p[q];
    for (int q = 0; q < RP_MAX_STATE; ++q) P.y[q] = net->y[q];
    P.dt = net->dt;
    P.inv_dt = 1.0f / net->dt;
    P.theta = net->theta;
    P.v_reset = net->v_reset;
    P.scratch = net->scratch;
    P.sp_vals = net->sp_vals;
    P.sp_cidx = net->sp_cidx;
    P.sp_rowoff = net->sp_rowoff;
    P.sp_row_ids = net->sp_row_ids;
    P.sp_batch_class = net->sp_batch_class;
    P.sp_phase_window = net->sp_phase_window;
    P.sp_block_rows = net->sp_block_rows;
    P.sp_phases = net->sp_phases;
    P.sp_batches = net->sp_batches;
    P.model = net->model;
    P.vec4 = use_vec4(net) ? 1 : 0;
    P.act = net->act;
}

#ifdef RP_MEASUREMENT
// measurement build only (rp_set_tuning): 0 = use the built-in choice.
std::atomic<int> g_tune_R{0}, g_tune_U{0}, g_tune_nt{-1}, g_tune_blocks{0};
#endif

template <int MODEL, int R, int U, bool VEC4, bool NT>
void launch_cfg(const StepParams& P, hipStream_t st, int max_blocks) {
    int groups = (P.rows + R - 1) / R;
    int blocks = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL((step_kernel<MODEL, R, U, VEC4, NT>), dim3(blocks), dim3(kThreads), 0, st, P);
}

// ---- streaming configuration of the dense step kernel ---------------------------------------------------------------
// Rows per wave R, chunks in flight U and the number of resident workgroups change how W is walked, never what is
// summed in which order (rp_dot.h): every configuration gives the same bits.  The choice is (1) the benchmark override
// rp_set_tuning, (2) a configuration measured on this device for this shape by rp_tune_forward, (3) a static rule.
struct StreamCfg {
    int R, U, blocks;  // blocks = 0: one workgroup per 4 row groups
};

struct TunedShape {
    int device, rows, n;
    int64_t ldw;
    StreamCfg cfg;
    float us;
};
std::mutex g_tuned_mu;
std::vector<TunedShape> g_tuned;

bool tuned_lookup(int rows, int n, int64_t ldw, StreamCfg* cfg, float* us) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    for (const TunedShape& t : g_tuned)
        if (t.device == dev && t.rows == rows && t.n == n && t.ldw == ldw) {
            if (cfg) *cfg = t.cfg;
            if (us) *us = t.us;
            return true;
        }
    return false;
}

bool streams_from_hbm(int rows, int n) { return rows >= 8192 && (int64_t)rows * n * 4 > (320ll << 20); }

// measured with scripts/archive/tune_forward.py / tune_shard_shapes.py / sweep_forward_shapes.py (us per step):
//   n = 32768: (4,2) on 2048 workgroups 620, (2,1) on 683 604;  16384^2 (4,2) 166, (2,1) on 683 151;
//   16384 x 32768 314 -> 303;  8192 x 32768 (4,2) 163, (2,1) on 1024 155;  4096 x 32768 (2,2) 79
//   65536^2 and 32768 x 262144: (4,2) on 2048, 7.0 TB/s, nothing better found
//   n = 12288: (4,2) 90 vs (8,1) 99;  n = 8192 (W ~ Infinity-Cache sized): (8,1) 43 vs (4,2) 51;
//   n = 4096: (2,2) 11.8;  n = 2048: (1,4) 4.9 vs (2,2) 6.4
// With rows of at most 128 KiB, two rows x one chunk per wave on about 683 resident workgroups that split the row
// groups into whole passes is the fastest walk found; which grid is best depends on the shape in ways only a
// measurement shows (e.g. 20000^2: (4,2) on 512 230 us, this rule 261), hence rp_tune_forward.
StreamCfg static_cfg(int rows, int n) {
    if (streams_from_hbm(rows, n)) {
        if (n <= 32768 && n % 256 == 0) {
            const int groups = (rows + 1) / 2;
            const int passes = std::max(1, (groups + 683) / 2732);
            return StreamCfg{2, 1, (groups + 4 * passes - 1) / (4 * passes)};
        }
        return StreamCfg{4, 2, 2048};
    }
    if (rows >= 8192) return StreamCfg{8, 1, 2048};
    if (rows > 2048) return StreamCfg{2, 2, 0};
    return StreamCfg{1, 4, 0};
}

StreamCfg choose_cfg(const rp_net* net) {
    StreamCfg cfg;
    if (use_vec4(net) && streams_from_hbm(net->rows, net->n) && tuned_lookup(net->rows, net->n, net->ldw, &cfg, nullptr))
        return cfg;
    return static_cfg(net->rows, net->n);
}

template <int MODEL, bool VEC4>
void launch_step(const StepParams& P, hipStream_t st, const StreamCfg& cfg) {
#ifdef RP_MEASUREMENT
    if constexpr (VEC4 && MODEL == RP_MODEL_LI_TANH) {
        if (g_tune_R > 0) {  // sweep of the streaming configuration
            const int key = g_tune_R * 100 + g_tune_U * 10 + (g_tune_nt != 0 ? 1 : 0);
            const int mb = g_tune_blocks;
            switch (key) {
#define RP_CFG(R_, U_)                                                                  \
    case R_ * 100 + U_ * 10 + 1: launch_cfg<MODEL, R_, U_, true, true>(P, st, mb); return; \
    case R_ * 100 + U_ * 10 + 0: launch_cfg<MODEL, R_, U_, true, false>(P, st, mb); return;
                RP_CFG(1, 1) RP_CFG(1, 2) RP_CFG(1, 4) RP_CFG(1, 8) RP_CFG(2, 8) RP_CFG(3, 1)
                RP_CFG(2, 1) RP_CFG(2, 2) RP_CFG(2, 4)
                RP_CFG(4, 1) RP_CFG(4, 2) RP_CFG(4, 4)
                RP_CFG(8, 1) RP_CFG(8, 2)
#undef RP_CFG
                default: break;
            }
        }
    }
#endif
    if constexpr (VEC4) {
        switch (cfg.R * 10 + cfg.U) {  // the configurations rp_tune_forward may pick, for every model
            case 14: launch_cfg<MODEL, 1, 4, true, true>(P, st, cfg.blocks); return;
            case 21: launch_cfg<MODEL, 2, 1, true, true>(P, st, cfg.blocks); return;
            case 22: launch_cfg<MODEL, 2, 2, true, true>(P, st, cfg.blocks); return;
            case 41: launch_cfg<MODEL, 4, 1, true, true>(P, st, cfg.blocks); return;
            case 81: launch_cfg<MODEL, 8, 1, true, true>(P, st, cfg.blocks); return;
            default: launch_cfg<MODEL, 4, 2, true, true>(P, st, cfg.blocks); return;
        }
    } else {  // 4-byte path: no chunks in flight to choose
        switch (cfg.R) {
            case 1: launch_cfg<MODEL, 1, 1, false, true>(P, st, cfg.blocks); return;
            case 2: launch_cfg<MODEL, 2, 1, false, true>(P, st, cfg.blocks); return;
            case 8: launch_cfg<MODEL, 8, 1, false, true>(P, st, cfg.blocks); return;
            default: launch_cfg<MODEL, 4, 1, false, true>(P, st, cfg.blocks); return;
        }
    }
}

template <bool VEC4>
void launch_step_model(int model, const StepParams& P, hipStream_t st, const StreamCfg& cfg) {
    switch (model) {
        case RP_MODEL_LI_TANH: launch_step<RP_MODEL_LI_TANH, VEC4>(P, st, cfg); break;
        case RP_MODEL_LI_SIGMOID: launch_step<RP_MODEL_LI_SIGMOID, VEC4>(P, st, cfg); break;
        case RP_MODEL_QIF: launch_step<RP_MODEL_QIF, VEC4>(P, st, cfg); break;
        case RP_MODEL_QIF_SFA: launch_step<RP_MODEL_QIF_SFA, VEC4>(P, st, cfg); break;
        case RP_MODEL_LIF: launch_step<RP_MODEL_LIF, VEC4>(P, st, cfg); break;
        case RP_MODEL_IK: launch_step<RP_MODEL_IK, VEC4>(P, st, cfg); break;
        case RP_MODEL_IKU: launch_step<RP_MODEL_IKU, VEC4>(P, st, cfg); break;
        default: launch_step<RP_MODEL_IK_BIEXP, VEC4>(P, st, cfg); break;
    }
}

// ---- populations with dense edges (rp_run.edge_*) ------------------------------------------------------------------
// The walk follows the static rule for the bytes a row group streams: the columns of W plus those of every edge.
template <int MODEL, int R, int U, bool VEC4>
void launch_edges_cfg(const StepParams& P, hipStream_t st, int max_blocks) {
    int groups = (P.rows + R - 1) / R;
    int blocks = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL((step_edges_kernel<MODEL, R, U, VEC4>), dim3(blocks), dim3(kThreads), 0, st, P);
}

template <int MODEL>
void launch_step_edges(const StepParams& P, hipStream_t st, bool vec4) {
    int64_t cols = MODEL == RP_MODEL_INSTANT ? 0 : P.n;
    for (int e = 0; e < P.n_edges; ++e) cols += P.edge[e].W ? P.edge[e].n : 1;
    if (cols < 1) cols = 1;
    StreamCfg cfg = static_cfg(P.rows, (int)std::min<int64_t>(cols, 1 << 30));
#ifdef RP_MEASUREMENT
    if (g_tune_R > 0) cfg = StreamCfg{g_tune_R, g_tune_U, g_tune_blocks};  // sweep (rp_set_tuning)
#endif
    if (!vec4) {  // 4-byte recurrent path (small test networks): one instance
        launch_edges_cfg<MODEL, 4, 1, false>(P, st, cfg.blocks);
        return;
    }
    switch (cfg.R * 10 + cfg.U) {
        case 14: launch_edges_cfg<MODEL, 1, 4, true>(P, st, cfg.blocks); return;
        case 18: launch_edges_cfg<MODEL, 1, 8, true>(P, st, cfg.blocks); return;
        case 21: launch_edges_cfg<MODEL, 2, 1, true>(P, st, cfg.blocks); return;
        case 22: launch_edges_cfg<MODEL, 2, 2, true>(P, st, cfg.blocks); return;
        case 81: launch_edges_cfg<MODEL, 8, 1, true>(P, st, cfg.blocks); return;
        default: launch_edges_cfg<MODEL, 4, 2, true>(P, st, cfg.blocks); return;
    }
}

void launch_step_edges_model(int model, const StepParams& P, hipStream_t st, bool vec4) {
    switch (model) {
        case RP_MODEL_LI_TANH: launch_step_edges<RP_MODEL_LI_TANH>(P, st, vec4); break;
        case RP_MODEL_LI_SIGMOID: launch_step_edges<RP_MODEL_LI_SIGMOID>(P, st, vec4); break;
        case RP_MODEL_QIF: launch_step_edges<RP_MODEL_QIF>(P, st, vec4); break;
        case RP_MODEL_QIF_SFA: launch_step_edges<RP_MODEL_QIF_SFA>(P, st, vec4); break;
        case RP_MODEL_LIF: launch_step_edges<RP_MODEL_LIF>(P, st, vec4); break;
        case RP_MODEL_IK: launch_step_edges<RP_MODEL_IK>(P, st, vec4); break;
        case RP_MODEL_IKU: launch_step_edges<RP_MODEL_IKU>(P, st, vec4); break;
        case RP_MODEL_INSTANT: launch_step_edges<RP_MODEL_INSTANT>(P, st, vec4); break;
        default: launch_step_edges<RP_MODEL_IK_BIEXP>(P, st, vec4); break;
    }
}

}  // namespace

extern "C" {

int rp_abi_version(void) { return RP_ABI_VERSION; }
int rp_async_status(void) { return rp_resident_poll(); }

const char* rp_last_error(void) { return rp_err_buf; }

int rp_sizeof_net(void) { return (int)sizeof(rp_net); }
int rp_sizeof_run(void) { return (int)sizeof(rp_run); }
int rp_sizeof_bwd(void) { return (int)sizeof(rp_bwd); }

int rp_device_check(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) RP_FAIL(RP_ERR_ARCH, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count) RP_FAIL(RP_ERR_ARG, "device %d out of range [0,%d)", device, count);
    hipDeviceProp_t prop;
    RP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        RP_FAIL(RP_ERR_ARCH, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return RP_OK;
}

int rp_device_name(int device, char* buf, int32_t len) {
    if (!buf || len < 2) RP_FAIL(RP_ERR_ARG, "rp_device_name: buffer missing");
    hipDeviceProp_t prop;
    RP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    snprintf(buf, (size_t)len, "%s", prop.name);
    return RP_OK;
}

int rp_refresh_source(const rp_net* net, void* stream) {
    int rc = validate_net(net);
    if (rc != RP_OK) return rc;
    if (net->model >= RP_MODEL_QIF) return RP_OK;
    StepParams P = {};
    fill_static(P, net);
    P.src_next = net->src[net->cur];
    hipStream_t st = static_cast<hipStream_t>(stream);
    int blocks = (net->rows + 255) / 256;
    if (net->model == RP_MODEL_LI_TANH)
        hipLaunchKernelGGL(refresh_kernel<RP_MODEL_LI_TANH>, dim3(blocks), dim3(256), 0, st, P);
    else
        hipLaunchKernelGGL(refresh_kernel<RP_MODEL_LI_SIGMOID>, dim3(blocks), dim3(256), 0, st, P);
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

}  // extern "C"

namespace {

typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
constexpr int kNcclFloat = 7;  // ncclFloat32 (rccl.h)

// RP_COMM_DIRECT: copy this rank's slice of the next source vector into the same rows of every peer's buffer (peer
// memory mapped through hipIpc; blockIdx.y = peer), 16 bytes per lane.
struct PushParams {
    const float* src;
    float* dst[63];
    int quads, peers;
};
__global__ __launch_bounds__(256) void push_slice_kernel(const PushParams Q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < Q.quads) reinterpret_cast<rp_f4*>(Q.dst[blockIdx.y])[i] = reinterpret_cast<const rp_f4*>(Q.src)[i];
}

// ---- one window = a host loop over the steps ------------------------------------------------------------------------
// Argument checks of one (net, run) pair; `coupled`: the pair is a member of rp_forward_coupled (edge sources come from
// the other members, not from run->edge_src).
int check_run(const rp_net* net, const rp_run* run, bool coupled) {
    if (!run) RP_FAIL(RP_ERR_ARG, "run is null");
    if (run->spike_src && (net->model < RP_MODEL_QIF || net->model == RP_MODEL_IKU || net->model == RP_MODEL_IK_BIEXP))
        RP_FAIL(RP_ERR_ARG, "spike_src is for the spiking models without population means (model %d)", net->model);
    if (run->steps < 0) RP_FAIL(RP_ERR_ARG, "steps < 0");
    if (run->sampling_steps < 1) RP_FAIL(RP_ERR_ARG, "sampling_steps must be >= 1");
    if (run->cutoff < 0 || run->step0 < 0) RP_FAIL(RP_ERR_ARG, "cutoff/step0 must be >= 0");
    if (run->out_var != RP_OUT_STATE0 && run->out_var != RP_OUT_SOURCE)
        RP_FAIL(RP_ERR_ARG, "bad out_var %d", run->out_var);
    if (run->x_target != RP_XT_MEMBRANE && !(run->x_target == RP_XT_SYNAPSE && net->model == RP_MODEL_LIF))
        RP_FAIL(RP_ERR_ARG, "x_target %d is not available for model %d", run->x_target, net->model);
    switch (run->x_mode) {
        case RP_X_NONE: break;
        case RP_X_FULL:
            if (!run->x || run->ldx < net->rows) RP_FAIL(RP_ERR_SHAPE, "RP_X_FULL needs x with ldx >= rows");
            break;
        case RP_X_BCAST:
            if (!run->x || run->ldx < 1) RP_FAIL(RP_ERR_SHAPE, "RP_X_BCAST needs x with ldx >= 1");
            break;
        case RP_X_PROJ:
            if (!run->x || !run->w_in || run->m < 1 || run->ldx < run->m || run->ldwin < run->m)
                RP_FAIL(RP_ERR_SHAPE, "RP_X_PROJ needs x (ldx >= m), w_in (ldwin >= m), m >= 1");
            break;
        default: RP_FAIL(RP_ERR_ARG, "bad x_mode %d", run->x_mode);
    }
    if (run->traj && run->ldtraj < net->rows) RP_FAIL(RP_ERR_SHAPE, "ldtraj < rows");
    bool any_save = run->save_src || run->save_acc || run->save_in;
    for (int q = 0; q < RP_MAX_STATE; ++q) any_save = any_save || run->save_state[q];
    if (any_save && run->ldsave < net->rows) RP_FAIL(RP_ERR_SHAPE, "ldsave < rows");
    bool any_rec = false;
    for (int q = 0; q < RP_MAX_STATE; ++q) any_rec = any_rec || run->rec[q];
    if ((run->samples || any_rec)) {
        if (run->ldsamples < net->rows) RP_FAIL(RP_ERR_SHAPE, "ldsamples < rows");
        if (!run->acc) RP_FAIL(RP_ERR_ARG, "acc buffer required when sampling");
    }
    if (run->save_src_full && run->ldfull < net->n) RP_FAIL(RP_ERR_SHAPE, "ldfull < n");
    if (run->spike_count && (!run->spike_events || run->spike_capacity < 0))
        RP_FAIL(RP_ERR_ARG, "spike_count given without spike_events");
    if (use_vec4(net) && (!aligned16(net->W) || !aligned16(net->src[0]) || !aligned16(net->src[1])))
        RP_FAIL(RP_ERR_ARG, "W and src buffers must be 16-byte aligned when n and ldw are multiples of 4");
    if (net->sp_vals) {
        if (!net->sp_cidx || !net->sp_rowoff) RP_FAIL(RP_ERR_ARG, "sp_vals given without sp_cidx / sp_rowoff");
        if (net->n % 4 != 0) RP_FAIL(RP_ERR_SHAPE, "the packed form of W needs n %% 4 == 0 (n=%d)", net->n);
        if (!aligned16(net->sp_vals) || (reinterpret_cast<uintptr_t>(net->sp_cidx) & 7u) || !aligned16(net->src[0]) ||
            !aligned16(net->src[1]))
            RP_FAIL(RP_ERR_ARG, "sp_vals / src must be 16-byte and sp_cidx 8-byte aligned");
        if (net->sp_row_ids && (!net->sp_batch_class || !net->sp_phase_window || net->sp_phases < 3 || net->sp_batches < 0))
            RP_FAIL(RP_ERR_ARG, "windowed packed form needs sp_batch_class, sp_phase_window, sp_phases >= 3, sp_batches");
        if (net->sp_block_rows < 0 || net->sp_block_rows > 64 ||
            (net->sp_block_rows > 0 && (net->sp_row_ids || net->n > kSparseLdsMaxN)))
            RP_FAIL(RP_ERR_ARG, "sp_block_rows must be in [0, 64], without the windowed form and with n <= %d", kSparseLdsMaxN);
    }
    if (run->n_edges < 0 || run->n_edges > RP_MAX_EDGES) RP_FAIL(RP_ERR_ARG, "n_edges must be in [0, %d]", RP_MAX_EDGES);
    if (net->model == RP_MODEL_INSTANT && (run->x_mode == RP_X_PROJ || run->hist[0] || run->save_src || run->save_acc ||
                                           run->save_state[0] || run->spike_count || any_rec))
        RP_FAIL(RP_ERR_ARG, "an activation-node member takes x (full / broadcast), edges, traj, samples and save_in only");
    for (int e = 0; e < run->n_edges; ++e) {
        if (!run->edge_w[e]) {  // the identity edge
            if (run->edge_n[e] != net->rows) RP_FAIL(RP_ERR_SHAPE, "edge %d: an identity edge (edge_w == NULL) needs edge_n == rows", e);
            if (run->edge_sp[e].vals) RP_FAIL(RP_ERR_ARG, "edge %d: an identity edge has no packed form", e);
        } else if (run->edge_n[e] < 1 || run->edge_ld[e] < run->edge_n[e])
            RP_FAIL(RP_ERR_SHAPE, "edge %d needs edge_n >= 1 and edge_ld >= edge_n", e);
        const bool v4 = run->edge_w[e] && run->edge_n[e] % 4 == 0 && run->edge_ld[e] % 4 == 0;
        if (v4 && !aligned16(run->edge_w[e])) RP_FAIL(RP_ERR_ARG, "edge %d: edge_w must be 16-byte aligned", e);
        if (!coupled) {
            if (!run->edge_src[e]) RP_FAIL(RP_ERR_ARG, "edge %d has no source vector", e);
            if (v4 && (!aligned16(run->edge_src[e]) || run->edge_src_ld[e] % 4 != 0))
                RP_FAIL(RP_ERR_ARG, "edge %d: edge_src / edge_src_ld must be 16-byte aligned", e);
        }
        const rp_edge_mem& M = run->edge_mem[e];
        if (M.kind != RP_EMEM_NONE) {
            const int rc = rp_internal_edge_mem_check(&M);
            if (rc != RP_OK) return rc;
            if (M.n != run->edge_n[e]) RP_FAIL(RP_ERR_SHAPE, "edge %d: edge_mem.n = %d, edge_n = %d", e, M.n, run->edge_n[e]);
            if (M.kind != RP_EMEM_FILTER && !M.u) RP_FAIL(RP_ERR_ARG, "edge %d: edge_mem.u is required", e);
        }
    }
    if ((run->hist[0] == nullptr) != (run->hist[1] == nullptr) || (run->hist[0] == nullptr) != (run->hist[2] == nullptr))
        RP_FAIL(RP_ERR_ARG, "hist: all three buffers or none");
    if (run->hist[0] && (run->hist[0] == run->hist[1] || run->hist[0] == run->hist[2] || run->hist[1] == run->hist[2]))
        RP_FAIL(RP_ERR_ARG, "the hist buffers must be distinct");

    // count the samples this call produces before touching anything
    int64_t produced = 0;
    for (int64_t t = run->step0; t < run->step0 + run->steps; ++t)
        if (t >= run->cutoff && t % run->sampling_steps == 0) ++produced;
    if ((run->samples || any_rec) && run->sample_idx + produced > run->max_samples)
        RP_FAIL(RP_ERR_SHAPE, "sample buffers hold %d rows, call needs %lld", run->max_samples,
                (long long)(run->sample_idx + produced));
    return RP_OK;
}

// the parts of the kernel arguments that stay the same for the whole call
void init_params(StepParams& P, const rp_net* net, const rp_run* run) {
    fill_static(P, net);
    P.x_mode = run->x_mode;
    P.m = run->m;
    P.w_in = run->w_in;
    P.ldwin = run->ldwin;
    P.out_var = run->out_var;
    P.x_target = run->x_target;
    P.acc = run->acc;
    P.spike_events = run->spike_events;
    P.spike_capacity = run->spike_capacity;
    P.spike_count = run->spike_count;
    P.n_edges = run->n_edges;
    for (int e = 0; e < run->n_edges; ++e) {
        P.edge[e].W = run->edge_w[e];
        P.edge[e].ld = run->edge_ld[e];
        P.edge[e].n = run->edge_n[e];
        P.edge[e].vec4 = (run->edge_w[e] && run->edge_n[e] % 4 == 0 && run->edge_ld[e] % 4 == 0) ? 1 : 0;
        P.edge[e].src = nullptr;
    }
}

// the arguments of step t of the call (row pointers, sampling flags); advances the run's sampling bookkeeping
void set_step(StepParams& P, const rp_net* net, rp_run* run, int t) {
    const int64_t t_abs = run->step0 + t;
    P.src_cur = net->src[net->cur];
    P.src_next = net->src[net->cur ^ 1];
    P.x_t = run->x ? run->x + (int64_t)t * run->ldx : nullptr;
    P.traj_t = run->traj ? run->traj + (int64_t)t * run->ldtraj : nullptr;
    P.hist_t = run->hist[0] ? run->hist[t_abs % 3] : nullptr;
    P.hist_next = nullptr;
    for (int q = 0; q < RP_MAX_STATE; ++q)
        P.save_state_t[q] = run->save_state[q] ? run->save_state[q] + (int64_t)t * run->ldsave : nullptr;
    P.save_src_t = run->save_src ? run->save_src + (int64_t)t * run->ldsave : nullptr;
    P.save_acc_t = run->save_acc ? run->save_acc + (int64_t)t * run->ldsave : nullptr;
    P.save_in_t = run->save_in ? run->save_in + (int64_t)t * run->ldsave : nullptr;
    P.t_abs = (int)t_abs;
    P.spike_src = t == 0 ? run->spike_src : nullptr;
    P.flags = 0;
    P.sample_row = nullptr;
    for (int q = 0; q < RP_MAX_STATE; ++q) P.rec_row[q] = nullptr;
    if (run->acc && t_abs >= run->cutoff) {
        P.flags |= FLAG_BUFFER;
        if (run->acc_count == 0) P.flags |= FLAG_FIRST;  // an empty buffer: no reset pass over acc is ever needed
        run->acc_count += 1;
        if (t_abs % run->sampling_steps == 0) {
            P.flags |= FLAG_SAMPLE;
            P.count = (float)run->acc_count;
            const int64_t off = (int64_t)run->sample_idx * run->ldsamples;
            P.sample_row = run->samples ? run->samples + off : nullptr;
            for (int q = 0; q < RP_MAX_STATE; ++q) P.rec_row[q] = run->rec[q] ? run->rec[q] + off : nullptr;
            run->sample_idx += 1;
            run->acc_count = 0;
        }
    }
}

// the launches of one step of one population (P complete)
int launch_population_step(const rp_net* net, const StepParams& P, hipStream_t st, const StreamCfg& cfg, bool vec4) {
    if (net->model == RP_MODEL_IKU || net->model == RP_MODEL_IK_BIEXP)
        hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, st, P);
    if (P.n_edges > 0 || net->model == RP_MODEL_INSTANT) {
        launch_step_edges_model(net->model, P, st, vec4);
    } else if (net->sp_vals) {
        const int src_rc = launch_step_sparse_model(net->model, P, st);
        if (src_rc != RP_OK) RP_FAIL(src_rc, "sparse step launch failed (device properties / LDS attribute)");
    } else if (vec4) {
        launch_step_model<true>(net->model, P, st, cfg);
    } else {
        launch_step_model<false>(net->model, P, st, cfg);
    }
    return RP_OK;
}

// ---- the resident launch (resident_kernel): eligibility and the call ------------------------------------------------------
template <int MODEL>
void launch_resident_model(const StepParams& P, const ResidentArgs& A, bool v4, hipStream_t st) {
    const dim3 grid(rp_resident_grid(A.C));
#define RP_LAUNCH_(L_) hipLaunchKernelGGL((resident_kernel<MODEL, RP_UNPAREN L_>), grid, dim3(kThreads), 0, st, P, A)
    RP_RESIDENT_DISPATCH_LAYOUT_WIDE(P.n, v4, RP_LAUNCH_)
#undef RP_LAUNCH_
}

constexpr int kResidentMinSteps = 4;
bool resident_ok(const rp_net* net, const rp_run* run) {
    if (!rp_resident_enabled()) return false;
    if (net->model == RP_MODEL_IKU || net->model == RP_MODEL_IK_BIEXP || net->model == RP_MODEL_INSTANT || net->model < 0 ||
        net->model > RP_MODEL_IK_BIEXP)
        return false;
    if (net->sp_vals || run->n_edges > 0 || run->save_src_full || run->steps < kResidentMinSteps) return false;
    if (net->rows != net->n || net->row0 != 0) return false;
    if (use_vec4(net))   // the 16-byte path of the per-step kernels, or their 4-byte path: the same rule, the same bits
        return rp_resident_shape_wide(net->n, net->ldw) && aligned16(net->W) && aligned16(net->src[0]) && aligned16(net->src[1]);
    return rp_resident_shape_any(net->n);
}

void resident_args_of(const rp_net* net, const rp_run* run, ResidentArgs& A);

int forward_resident(rp_net* net, rp_run* run, StepParams& P, hipStream_t st) {
    ResidentArgs A = {};
    resident_args_of(net, run, A);
    void* ws = nullptr;
    const bool v4 = use_vec4(net);
    if (rp_resident_begin(net->n, st, &A.C, &ws, v4) != RP_OK) {   // (no stream-ordered allocator: one launch per step)
        (void)hipGetLastError();
        return kResidentUnavailable;
    }
    set_step(P, net, run, 0);   // (step 0's arguments -- the kernel forms every step's itself; the bookkeeping is advanced below)
    switch (net->model) {
        case RP_MODEL_LI_TANH: launch_resident_model<RP_MODEL_LI_TANH>(P, A, v4, st); break;
        case RP_MODEL_LI_SIGMOID: launch_resident_model<RP_MODEL_LI_SIGMOID>(P, A, v4, st); break;
        case RP_MODEL_QIF: launch_resident_model<RP_MODEL_QIF>(P, A, v4, st); break;
        case RP_MODEL_QIF_SFA: launch_resident_model<RP_MODEL_QIF_SFA>(P, A, v4, st); break;
        case RP_MODEL_LIF: launch_resident_model<RP_MODEL_LIF>(P, A, v4, st); break;
        default: launch_resident_model<RP_MODEL_IK>(P, A, v4, st); break;
    }
    const int rc = rp_resident_end(A.C, ws, st, "the resident step kernel");
    if (rc != RP_OK) return rc;
    StepParams Q = P;
    net->cur ^= 1;
    for (int t = 1; t < run->steps; ++t) {   // the sampling bookkeeping and the buffer parity of the steps the kernel ran
        set_step(Q, net, run, t);
        net->cur ^= 1;
    }
    return RP_OK;
}

void resident_args_of(const rp_net* net, const rp_run* run, ResidentArgs& A) {
    A.steps = run->steps;
    A.cur0 = net->cur;
    A.src[0] = net->src[0];
    A.src[1] = net->src[1];
    A.x = run->x;
    A.ldx = run->ldx;
    A.traj = run->traj;
    A.ldtraj = run->ldtraj;
    for (int q = 0; q < 3; ++q) A.hist[q] = run->hist[q];
    for (int q = 0; q < RP_MAX_STATE; ++q) {
        A.save_state[q] = run->save_state[q];
        A.rec[q] = run->rec[q];
    }
    A.save_src = run->save_src;
    A.save_acc = run->save_acc;
    A.save_in = run->save_in;
    A.ldsave = run->ldsave;
    A.step0 = run->step0;
    A.cutoff = run->cutoff;
    A.sampling_steps = run->sampling_steps;
    A.acc_count0 = run->acc_count;
    A.sample_idx0 = run->sample_idx;
    A.has_acc = run->acc ? 1 : 0;
    A.samples = run->samples;
    A.ldsamples = run->ldsamples;
}

// rp_forward / rp_forward_sharded: `comm` != nullptr adds the per-step all-gather of the source slices.
int forward_impl(rp_net* net, rp_run* run, const rp_comm* comm, void* stream) {
    int rc = rp_resident_poll();
    if (rc != RP_OK) return rc;
    rc = validate_net(net);
    if (rc != RP_OK) return rc;
    if (!run) RP_FAIL(RP_ERR_ARG, "run is null");
    const bool direct = comm && comm->mode == RP_COMM_DIRECT;
    if (comm) {
        if (comm->mode != RP_COMM_RCCL && comm->mode != RP_COMM_DIRECT) RP_FAIL(RP_ERR_ARG, "bad rp_comm.mode %d", comm->mode);
        if (!direct && (!comm->comm || !comm->all_gather)) RP_FAIL(RP_ERR_ARG, "rp_comm needs comm and all_gather");
        if (direct) {
            if (!comm->peer_src || !comm->peer_flags || !comm->flags) RP_FAIL(RP_ERR_ARG, "RP_COMM_DIRECT needs peer_src, peer_flags and flags");
            if (comm->world > 64) RP_FAIL(RP_ERR_ARG, "RP_COMM_DIRECT: at most 64 ranks");
            for (int p = 0; p < comm->world; ++p) {
                if (!comm->peer_flags[p] || !comm->peer_src[2 * p] || !comm->peer_src[2 * p + 1])
                    RP_FAIL(RP_ERR_ARG, "RP_COMM_DIRECT: peer %d has a null buffer", p);
            }
            if (comm->peer_src[2 * comm->rank] != net->src[0] || comm->peer_src[2 * comm->rank + 1] != net->src[1] ||
                comm->peer_flags[comm->rank] != comm->flags)
                RP_FAIL(RP_ERR_ARG, "RP_COMM_DIRECT: this rank's entries of peer_src / peer_flags must be its own buffers");
            if (net->rows % 4 != 0) RP_FAIL(RP_ERR_SHAPE, "RP_COMM_DIRECT pushes 16-byte pieces: rows %% 4 == 0 required");
        }
        if (comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world)
            RP_FAIL(RP_ERR_ARG, "bad rank %d / world %d", comm->rank, comm->world);
        if ((int64_t)net->rows * comm->world != net->n || net->row0 != comm->rank * net->rows)
            RP_FAIL(RP_ERR_SHAPE, "shard rows=%d row0=%d does not match rank %d of %d over n=%d", net->rows, net->row0,
                    comm->rank, comm->world, net->n);
        if (net->model == RP_MODEL_IKU || net->model == RP_MODEL_IK_BIEXP)
            RP_FAIL(RP_ERR_ARG, "population-mean models cannot be row-sharded");
    }
    rc = check_run(net, run, false);
    if (rc != RP_OK) return rc;

    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec4 = use_vec4(net);
    const StreamCfg stream_cfg = choose_cfg(net);
    StepParams P = {};
    init_params(P, net, run);
    if (!comm && resident_ok(net, run)) {
        rc = forward_resident(net, run, P, st);
        if (rc != kResidentUnavailable) return rc;
    }

    for (int t = 0; t < run->steps; ++t) {
        const int64_t t_abs = run->step0 + t;
        set_step(P, net, run, t);
        for (int e = 0; e < run->n_edges; ++e) {
            P.edge[e].src = run->edge_src[e] + (int64_t)t * run->edge_src_ld[e];
            if (run->edge_mem[e].kind != RP_EMEM_NONE) {  // the edge's memory first: x -> u
                rc = rp_internal_edge_mem_step(&run->edge_mem[e], P.edge[e].src, &P.edge[e].src, st, t);
                if (rc != RP_OK) return rc;
            }
        }
        if (direct)  // every peer has pushed step seq (and is done with the buffer this step pushes into)
            for (int p = 0; p < comm->world; ++p)
                if (p != comm->rank)
                    RP_HIP_CHECK(hipStreamWaitValue32(st, comm->flags + p, comm->seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
        if (run->save_src_full)
            RP_HIP_CHECK(hipMemcpyAsync(run->save_src_full + (int64_t)t * run->ldfull, P.src_cur,
                                        sizeof(float) * (size_t)net->n, hipMemcpyDeviceToDevice, st));
        rc = launch_population_step(net, P, st, stream_cfg, vec4);
        if (rc != RP_OK) return rc;
        if (t == 0) RP_HIP_CHECK(hipGetLastError());  // a rejected launch configuration shows on the first step
        if (direct) {
            // push this rank's slice into every peer's next buffer, then tell them from the stream; the wait for the
            // peers' previous step sits in front of the step kernel (below, for the next step; here for this one)
            rp_comm* c = const_cast<rp_comm*>(comm);
            PushParams Q = {};
            Q.src = P.src_next + net->row0;
            Q.quads = net->rows / 4;
            Q.peers = 0;
            for (int p = 0; p < comm->world; ++p)
                if (p != comm->rank) Q.dst[Q.peers++] = comm->peer_src[2 * p + (net->cur ^ 1)] + net->row0;
            if (Q.peers > 0) {
                const int per = (Q.quads + 255) / 256;
                hipLaunchKernelGGL(push_slice_kernel, dim3(per, Q.peers), dim3(256), 0, st, Q);
            }
            c->seq += 1;
            for (int p = 0; p < comm->world; ++p)
                if (p != comm->rank)
                    RP_HIP_CHECK(hipStreamWriteValue32(st, comm->peer_flags[p] + comm->rank, c->seq, 0));
        } else if (comm) {  // every rank wrote rows [row0, row0 + rows) of src_next: exchange in place, same stream
            const int nrc = reinterpret_cast<all_gather_fn>(comm->all_gather)(
                P.src_next + net->row0, P.src_next, (size_t)net->rows, kNcclFloat, comm->comm, st);
            if (nrc != 0) RP_FAIL(RP_ERR_COMM, "ncclAllGather returned %d at step %lld", nrc, (long long)t_abs);
        }
        net->cur ^= 1;
    }
    if (direct && run->steps > 0)
        // Later work on this stream (a copy or clone of the source buffers, a restore into them, the host's read after a
        // synchronize) must come after the peers' LAST pushes into this rank's buffers: wait for their final sequence number.
        for (int p = 0; p < comm->world; ++p)
            if (p != comm->rank)
                RP_HIP_CHECK(hipStreamWaitValue32(st, comm->flags + p, comm->seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

// ---- the resident launch of a coupled circuit (resident_coupled_kernel): eligibility and the call --------------------------------
// Circuits the shared launch takes (coupled_impl: `merged`) whose matrices are all on the 16-byte path with at most 1024 columns,
// (1 + in-edges) x chunks <= 8 per member (chunks: of the circuit's widest matrix).  kResidentUnavailable: nothing launched, the caller goes on with a launch per step.
int coupled_resident(rp_member* mem, int count, StepParams* P, const bool* vec4, hipStream_t st) {
    const int steps = mem[0].run->steps;
    if (!rp_resident_enabled() || steps < kResidentMinSteps) return kResidentUnavailable;
    int nseg = 2, widest = 0;
    for (int m = 0; m < count; ++m) {
        const rp_net* net = mem[m].net;
        const rp_run* run = mem[m].run;
        if (!vec4[m] || net->sp_vals || net->n > kRcCap || !rp_resident_shape(net->n, net->ldw) || !aligned16(net->W) ||
            !aligned16(net->src[0]) || !aligned16(net->src[1]) || run->n_edges + 1 > kRcSlots)
            return kResidentUnavailable;
        widest = std::max(widest, net->n);
        nseg = std::max(nseg, run->n_edges + 1);
        for (int e = 0; e < run->n_edges; ++e)
            if (!P[m].edge[e].vec4 || !aligned16(run->edge_w[e]) || run->edge_n[e] > kRcCap) return kResidentUnavailable;
    }
    const int nc = (widest + 255) / 256;   // chunks per slot, by the widest matrix of the circuit
    if (nseg * nc > 8 || nc > 4) return kResidentUnavailable;
    const int rw = nseg * nc <= 2 ? 16 : (nseg * nc <= 4 ? 8 : 4);

    static_assert(sizeof(RcParams) <= 8192, "the members' argument blocks: a few KiB behind the granules");
    RcParams H = {};
    H.count = count;
    int blocks = 0;
    const int64_t a0 = mem[0].run->step0;
    for (int m = 0; m < count; ++m) {
        rp_net* net = mem[m].net;
        rp_run* run = mem[m].run;
        RcMember& M = H.M[m];
        M.P0 = P[m];
        resident_args_of(net, run, M.A);
        M.first_block = blocks;
        blocks += ((net->rows + rw - 1) / rw + kWavesPerBlock - 1) / kWavesPerBlock;
        M.nslots = 1 + run->n_edges;
        M.model = net->model;
        M.slot_src[0] = m;
        M.slot_n[0] = net->n;
        M.slot_w[0] = net->W;
        M.slot_ld[0] = net->ldw;
        M.slot_vec0[0] = net->src[net->cur];
        for (int e = 0; e < run->n_edges; ++e) {
            const int k = e + 1, q = mem[m].edge_source[e];
            const rp_net* qn = mem[q].net;
            const rp_run* qr = mem[q].run;
            const bool state_out = qr->out_var == RP_OUT_STATE0;
            M.slot_src[k] = q;
            M.slot_state[k] = state_out ? 1 : 0;
            M.slot_lag[k] = mem[m].edge_mode[e] == RP_EDGE_LAG ? 1 : 0;
            M.slot_n[k] = run->edge_n[e];
            M.slot_w[k] = run->edge_w[e];
            M.slot_ld[k] = run->edge_ld[e];
            // (the state vector of this step from the ring -- seeded by the caller --, never from y: the step overwrites that)
            M.slot_vec0[k] = state_out ? qr->hist[a0 % 3] : qn->src[qn->cur];
            M.slot_lag0[k] = M.slot_lag[k] ? qr->hist[(a0 + 2) % 3] : nullptr;
            if (state_out) H.M[q].publish_v = 1;
        }
        for (int q = 0; q < count; ++q) M.all_rows[q] = mem[q].net->rows;
    }
    rp_resident_ctl C = {};
    void* ws = nullptr;
    void* dev_params = nullptr;
    if (rp_resident_begin_streams(2 * count, kRcCap, blocks, sizeof(RcParams), st, &C, &ws, &dev_params) != RP_OK) {
        (void)hipGetLastError();
        return kResidentUnavailable;
    }
    for (int m = 0; m < count; ++m) {   // step 0's arguments (the kernel forms every step's itself)
        set_step(P[m], mem[m].net, mem[m].run, 0);
        H.M[m].P0 = P[m];
    }
    if (hipMemcpyAsync(dev_params, &H, sizeof(RcParams), hipMemcpyHostToDevice, st) != hipSuccess) {
        rp_resident_release(st);
        RP_FAIL(RP_ERR_HIP, "rp_forward_coupled: copying the members' argument blocks failed");
    }
    const dim3 grid(rp_resident_grid(C));
    const RcParams* D = static_cast<const RcParams*>(dev_params);
#define RP_RC_(S_, C_, R_) hipLaunchKernelGGL((resident_coupled_kernel<S_, C_, R_>), grid, dim3(kThreads), 0, st, C, D)
    switch (nseg * 10 + nc) {   // (slots, chunks per slot) -> rows per wave: at most 128 registers of matrix rows
        case 21: RP_RC_(2, 1, 16); break;
        case 31: RP_RC_(3, 1, 8); break;
        case 41: RP_RC_(4, 1, 8); break;
        case 22: RP_RC_(2, 2, 8); break;
        case 32: RP_RC_(3, 2, 4); break;
        case 42: RP_RC_(4, 2, 4); break;
        case 23: RP_RC_(2, 3, 4); break;
        default: RP_RC_(2, 4, 4); break;   // (24)
    }
#undef RP_RC_
    const int rc = rp_resident_end(C, ws, st, "the resident step kernel of a coupled circuit");
    if (rc != RP_OK) return rc;
    for (int m = 0; m < count; ++m) {   // the sampling bookkeeping and the buffer parity of the steps the kernel ran
        StepParams Q = P[m];
        mem[m].net->cur ^= 1;
        for (int t = 1; t < steps; ++t) {
            set_step(Q, mem[m].net, mem[m].run, t);
            mem[m].net->cur ^= 1;
        }
    }
    return RP_OK;
}

// rp_forward_coupled: the members' steps interleaved, one launch per member and Euler step
int coupled_impl(rp_member* mem, int32_t count, void* stream) {
    constexpr int kMaxMembers = RP_MAX_MEMBERS;
    if (!mem || count < 1 || count > kMaxMembers) RP_FAIL(RP_ERR_ARG, "rp_forward_coupled: 1..%d members", kMaxMembers);
    if (const int late = rp_resident_poll(); late != RP_OK) return late;
    for (int m = 0; m < count; ++m) {
        if (!mem[m].net || !mem[m].run) RP_FAIL(RP_ERR_ARG, "member %d: net / run is null", m);
        int rc = validate_net(mem[m].net);
        if (rc != RP_OK) return rc;
        rc = check_run(mem[m].net, mem[m].run, true);
        if (rc != RP_OK) return rc;
        const rp_net* net = mem[m].net;
        const rp_run* run = mem[m].run;
        if (net->rows != net->n || net->row0 != 0) RP_FAIL(RP_ERR_ARG, "member %d: row shards cannot be coupled", m);
        if (run->spike_src || run->save_src_full) RP_FAIL(RP_ERR_ARG, "member %d: spike_src / save_src_full are not available here", m);
        if (run->steps != mem[0].run->steps || run->step0 != mem[0].run->step0)
            RP_FAIL(RP_ERR_ARG, "member %d: steps / step0 differ from member 0", m);
        for (int q = 0; q < m; ++q)
            if (mem[q].net == net || mem[q].run == run) RP_FAIL(RP_ERR_ARG, "members %d and %d share a net / run", q, m);
    }
    for (int m = 0; m < count; ++m) {
        const rp_run* run = mem[m].run;
        for (int e = 0; e < run->n_edges; ++e) {
            const int q = mem[m].edge_source[e], mode = mem[m].edge_mode[e];
            if (q < 0 || q >= count) RP_FAIL(RP_ERR_ARG, "member %d edge %d: source %d out of range", m, e, q);
            if (mode != RP_EDGE_LIVE && mode != RP_EDGE_OUT && mode != RP_EDGE_LAG)
                RP_FAIL(RP_ERR_ARG, "member %d edge %d: bad mode %d", m, e, mode);
            const rp_run* qrun = mem[q].run;
            if (run->edge_n[e] != mem[q].net->n)
                RP_FAIL(RP_ERR_SHAPE, "member %d edge %d has %d columns, its source holds %d neurons", m, e, run->edge_n[e], mem[q].net->n);
            if (mode == RP_EDGE_OUT && q >= m)
                RP_FAIL(RP_ERR_ARG, "member %d edge %d: RP_EDGE_OUT needs a source stepped earlier (source %d)", m, e, q);
            const bool q_instant = mem[q].net->model == RP_MODEL_INSTANT;  // (read through src[cur], whatever the mode)
            if (mode == RP_EDGE_LIVE && q == m && qrun->out_var == RP_OUT_STATE0 && !q_instant)
                RP_FAIL(RP_ERR_ARG, "member %d edge %d: a population cannot read its own state while it is stepped", m, e);
            if (!q_instant && (mode == RP_EDGE_LAG || (mode == RP_EDGE_OUT && qrun->out_var == RP_OUT_STATE0)) && !qrun->hist[0])
                RP_FAIL(RP_ERR_ARG, "member %d edge %d: source %d needs rp_run.hist", m, e, q);
            // every vector this edge can be pointed at
            const bool state_out = qrun->out_var == RP_OUT_STATE0;
            bool ok16 = true;
            if (qrun->hist[0]) ok16 = aligned16(qrun->hist[0]) && aligned16(qrun->hist[1]) && aligned16(qrun->hist[2]);
            ok16 = ok16 && aligned16(mem[q].net->y[RP_Y_V]) && aligned16(mem[q].net->src[0]) && aligned16(mem[q].net->src[1]);
            (void)state_out;
            if (run->edge_mem[e].kind != RP_EMEM_NONE && run->edge_mem[e].n % 4 == 0 && !ok16)
                RP_FAIL(RP_ERR_ARG, "member %d edge %d: the source vectors must be 16-byte aligned", m, e);
            if (run->edge_w[e] && run->edge_n[e] % 4 == 0 && run->edge_ld[e] % 4 == 0 && !ok16)
                RP_FAIL(RP_ERR_ARG, "member %d edge %d: the source vectors must be 16-byte aligned", m, e);
        }
    }

    hipStream_t st = static_cast<hipStream_t>(stream);
    StepParams P[kMaxMembers];
    StreamCfg cfg[kMaxMembers];
    bool vec4[kMaxMembers];
    for (int m = 0; m < count; ++m) {
        P[m] = StepParams{};
        init_params(P[m], mem[m].net, mem[m].run);
        cfg[m] = choose_cfg(mem[m].net);
        vec4[m] = use_vec4(mem[m].net);
    }
    const int steps = mem[0].run->steps;

    // One launch for all members per step?  Only values from BEFORE the step may be read: no RP_EDGE_LIVE edge from a
    // member stepped earlier; state-variable outputs come from the hist ring (written one step ahead), so their owners
    // need one; launch-bound sizes only (larger members want their own walk and lose nothing to the extra launches).
    bool merged = count >= 2 && count <= kMergeMax && steps > 0;
    int max_rows = 0;
    for (int m = 0; m < count && merged; ++m) {
        const rp_net* net = mem[m].net;
        max_rows = std::max(max_rows, net->rows);
        if (net->rows > 4096 || net->model == RP_MODEL_IKU || net->model == RP_MODEL_IK_BIEXP || net->model == RP_MODEL_INSTANT ||
            mem[m].run->n_edges > kMergeMaxEdges)
            merged = false;
        for (int e = 0; e < mem[m].run->n_edges && merged; ++e) {
            const int q = mem[m].edge_source[e], mode = mem[m].edge_mode[e];
            if (!mem[m].run->edge_w[e] || mem[m].run->edge_mem[e].kind != RP_EMEM_NONE) merged = false;
            if (mode == RP_EDGE_LIVE && q < m) merged = false;
            if (mode != RP_EDGE_LAG && mem[q].run->out_var == RP_OUT_STATE0 && !mem[q].run->hist[0]) merged = false;
        }
    }
    // members on the packed path (rp_run.edge_sp): every edge matrix in packed form (W itself packed or dense)
    bool packed[kMaxMembers];
    for (int m = 0; m < count; ++m) {
        const rp_run* run = mem[m].run;
        packed[m] = run->n_edges > 0 && run->edge_sum;
        for (int e = 0; e < run->n_edges && packed[m]; ++e) packed[m] = run->edge_sp[e].vals != nullptr;
        if (packed[m]) {
            if (!aligned16(run->edge_sum)) RP_FAIL(RP_ERR_ARG, "member %d: edge_sum must be 16-byte aligned", m);
            for (int e = 0; e < run->n_edges; ++e)
                if (run->edge_n[e] % 4 != 0 || !run->edge_sp[e].cidx || !run->edge_sp[e].rowoff)
                    RP_FAIL(RP_ERR_ARG, "member %d edge %d: a packed edge needs edge_n %% 4 == 0, cidx and rowoff", m, e);
            merged = false;
        }
    }
    if (const char* off = getenv("RECTIPY_AMD_COUPLED_LAUNCHES"); off && off[0] == 'm') merged = false;  // "member": one launch per member
    if (merged) {
        const int64_t a0 = mem[0].run->step0;
        for (int m = 0; m < count; ++m) {  // ring entry of the first step: the member's current output
            const rp_net* net = mem[m].net;
            const rp_run* run = mem[m].run;
            if (!run->hist[0]) continue;
            const float* cur_out = run->out_var == RP_OUT_STATE0 ? net->y[RP_Y_V] : net->src[net->cur];
            RP_HIP_CHECK(hipMemcpyAsync(run->hist[a0 % 3], cur_out, sizeof(float) * (size_t)net->rows, hipMemcpyDeviceToDevice, st));
        }
    }
    if (merged) {   // launch-bound sizes: all steps in ONE launch (the resident form of the shared launch)
        const int rc = coupled_resident(mem, count, P, vec4, st);
        if (rc != kResidentUnavailable) return rc;
    }
    const int mR = max_rows <= 2048 ? 1 : 2;  // rows per wave (chunks in flight: 8, the whole row up to 2048 columns)

    for (int t = 0; t < steps; ++t) {
        const int64_t t_abs = mem[0].run->step0 + t;
        for (int m = 0; m < count; ++m) {
            rp_net* net = mem[m].net;
            rp_run* run = mem[m].run;
            set_step(P[m], net, run, t);
            if (merged) {  // the ring is written one step ahead (post-update output); nothing reads this step's entry late
                P[m].hist_t = nullptr;
                P[m].hist_next = run->hist[0] ? run->hist[(t_abs + 1) % 3] : nullptr;
            }
            for (int e = 0; e < run->n_edges; ++e) {
                const int q = mem[m].edge_source[e];
                const rp_net* qn = mem[q].net;
                const rp_run* qr = mem[q].run;
                const bool state_out = qr->out_var == RP_OUT_STATE0;
                if (qn->model == RP_MODEL_INSTANT) {  // an activation node's output as it stands: this step's if q was stepped, else the last one
                    P[m].edge[e].src = qn->src[qn->cur];
                } else
                switch (mem[m].edge_mode[e]) {
                    case RP_EDGE_LIVE:  // qn->cur names q's CURRENT source buffer: flipped already if q < m (never when merged)
                        P[m].edge[e].src = state_out ? (merged ? qr->hist[t_abs % 3] : qn->y[RP_Y_V]) : qn->src[qn->cur];
                        break;
                    case RP_EDGE_OUT:   // q < m: its buffer of this step is the one it flipped away from (merged: not yet flipped)
                        P[m].edge[e].src = state_out ? qr->hist[t_abs % 3] : qn->src[merged ? qn->cur : qn->cur ^ 1];
                        break;
                    default:            // RP_EDGE_LAG
                        P[m].edge[e].src = qr->hist[(t_abs + 2) % 3];
                        break;
                }
                if (run->edge_mem[e].kind != RP_EMEM_NONE) {  // the edge's memory first: x -> u (never in the shared launch)
                    const int rc = rp_internal_edge_mem_step(&run->edge_mem[e], P[m].edge[e].src, &P[m].edge[e].src, st, t);
                    if (rc != RP_OK) return rc;
                }
            }
            if (!merged && packed[m]) {
                // one packed mat-vec per edge, chained into edge_sum (the last adds the member's input), then the member's own
                // step -- packed or dense, as its W is stored -- with edge_sum as its full-width input
                for (int e = 0; e < run->n_edges; ++e) {
                    StepParams E = {};
                    const rp_packed& S = run->edge_sp[e];
                    E.n = run->edge_n[e];
                    E.rows = net->rows;
                    E.src_cur = P[m].edge[e].src;
                    E.sp_vals = S.vals;
                    E.sp_cidx = S.cidx;
                    E.sp_rowoff = S.rowoff;
                    E.sp_row_ids = S.row_ids;
                    E.sp_batch_class = S.batch_class;
                    E.sp_phase_window = S.phase_window;
                    E.sp_phases = S.phases;
                    E.sp_batches = S.batches;
                    E.sp_block_rows = S.block_rows;
                    E.esum_out = run->edge_sum;
                    E.esum_prev = e > 0 ? run->edge_sum : nullptr;
                    E.esum_final = e == run->n_edges - 1;
                    E.x_mode = P[m].x_mode;
                    E.x_t = P[m].x_t;
                    E.m = P[m].m;
                    E.w_in = P[m].w_in;
                    E.ldwin = P[m].ldwin;
                    const int rc = launch_step_sparse_model(kModelEdgeSum, E, st);
                    if (rc != RP_OK) RP_FAIL(rc, "packed edge launch failed (device properties / LDS attribute)");
                }
                StepParams Q = P[m];
                Q.n_edges = 0;
                Q.x_mode = RP_X_FULL;
                Q.x_t = run->edge_sum;
                const int rc = launch_population_step(net, Q, st, cfg[m], vec4[m]);
                if (rc != RP_OK) return rc;
                if (t == 0) RP_HIP_CHECK(hipGetLastError());
                net->cur ^= 1;
            } else if (!merged) {
                const int rc = launch_population_step(net, P[m], st, cfg[m], vec4[m]);
                if (rc != RP_OK) return rc;
                if (t == 0) RP_HIP_CHECK(hipGetLastError());
                net->cur ^= 1;
            }
        }
        if (merged) {
            MergedParams M = {};
            M.count = count;
            int blocks = 0;
            for (int m = 0; m < count; ++m) {
                M.first_block[m] = blocks;
                M.P[m] = P[m];
                const int groups = (mem[m].net->rows + mR - 1) / mR;
                blocks += (groups + kWavesPerBlock - 1) / kWavesPerBlock;
            }
            M.first_block[count] = blocks;
            if (mR == 1)
                hipLaunchKernelGGL((step_merged_kernel<1, 8>), dim3(blocks), dim3(kThreads), 0, st, M);
            else
                hipLaunchKernelGGL((step_merged_kernel<2, 8>), dim3(blocks), dim3(kThreads), 0, st, M);
            if (t == 0) RP_HIP_CHECK(hipGetLastError());
            for (int m = 0; m < count; ++m) mem[m].net->cur ^= 1;
        }
    }
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

}  // namespace

// (declared in rp_common.h: the packed mat-vec of rp_forward_coupled's edges, also used by rp_backward_coupled for fixed sparse
// edges -- there with the transposed matrix)
int rp_internal_packed_matvec(const rp_packed* M, int n, int rows, const float* src, const float* prev, float* out, hipStream_t st) {
    StepParams E = {};
    E.n = n;
    E.rows = rows;
    E.src_cur = src;
    E.sp_vals = M->vals;
    E.sp_cidx = M->cidx;
    E.sp_rowoff = M->rowoff;
    E.sp_row_ids = M->row_ids;
    E.sp_batch_class = M->batch_class;
    E.sp_phase_window = M->phase_window;
    E.sp_phases = M->phases;
    E.sp_batches = M->batches;
    E.sp_block_rows = M->block_rows;
    E.esum_out = out;
    E.esum_prev = prev;
    E.esum_final = 0;
    E.x_mode = RP_X_NONE;
    return launch_step_sparse_model(kModelEdgeSum, E, st);
}

// (declared in rp_common.h) dst[i] = (F src)_i (+ x[i]) with the recurrent row dot's summation tree and streaming walk: the
// recursive filter of an edge with memory (rnn_edges.hip)
int rp_internal_filter_matvec(const float* F, int64_t ldf, int n, const float* src, const float* x, float* dst, hipStream_t st) {
    StepParams P = {};
    P.W = F;
    P.ldw = ldf;
    P.n = n;
    P.rows = n;
    P.src_cur = src;
    P.src_next = dst;
    P.x_mode = x ? RP_X_FULL : RP_X_NONE;
    P.x_t = x;
    const StreamCfg cfg = static_cfg(n, n);
    if (n % 4 == 0 && ldf % 4 == 0) {  // (which path is taken is part of the summation order: never decided by an address)
        if (!aligned16(F) || !aligned16(src)) RP_FAIL(RP_ERR_ARG, "edge filter: F and the state vectors must be 16-byte aligned");
        launch_step<kModelFilter, true>(P, st, cfg);
    } else
        launch_step<kModelFilter, false>(P, st, cfg);
    return RP_OK;
}

extern "C" {

int rp_forward(rp_net* net, rp_run* run, void* stream) { return forward_impl(net, run, nullptr, stream); }

int rp_forward_sharded(rp_net* net, rp_run* run, const rp_comm* comm, void* stream) {
    if (!comm) RP_FAIL(RP_ERR_ARG, "comm is null");
    return forward_impl(net, run, comm, stream);
}

int rp_forward_coupled(rp_member* members, int32_t count, void* stream) { return coupled_impl(members, count, stream); }

int rp_sizeof_member(void) { return (int)sizeof(rp_member); }

int rp_sizeof_comm(void) { return (int)sizeof(rp_comm); }

int rp_ipc_alloc(int64_t bytes, void** ptr, unsigned char handle[64]) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!ptr || !handle || bytes <= 0) RP_FAIL(RP_ERR_ARG, "rp_ipc_alloc: bad arguments");
    void* p = nullptr;
    RP_HIP_CHECK(hipMalloc(&p, (size_t)bytes));
    hipIpcMemHandle_t h;
    if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipIpcGetMemHandle(&h, p) != hipSuccess) {
        const hipError_t e = hipGetLastError();
        (void)hipFree(p);
        RP_FAIL(RP_ERR_HIP, "rp_ipc_alloc: %s", hipGetErrorString(e));
    }
    memcpy(handle, &h, 64);
    *ptr = p;
    return RP_OK;
}

int rp_ipc_open(const unsigned char handle[64], void** ptr) {
    if (!ptr || !handle) RP_FAIL(RP_ERR_ARG, "rp_ipc_open: bad arguments");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    RP_HIP_CHECK(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
    return RP_OK;
}

int rp_ipc_close(void* ptr) {
    if (ptr) RP_HIP_CHECK(hipIpcCloseMemHandle(ptr));
    return RP_OK;
}

int rp_ipc_free(void* ptr) {
    if (ptr) RP_HIP_CHECK(hipFree(ptr));
    return RP_OK;
}

#ifdef RP_MEASUREMENT
int rp_set_blocked_tuning(int waves, int groups_in_flight) {
    g_blocked_waves = waves;
    g_blocked_U = groups_in_flight;
    return RP_OK;
}

int rp_set_sparse_tuning(int rows_per_wave, int groups_in_flight) {
    g_sparse_R = rows_per_wave;
    g_sparse_U = groups_in_flight;
    return RP_OK;
}
#endif

int rp_tune_forward(const rp_net* net, void* stream) {
    if (!net || !net->W) RP_FAIL(RP_ERR_ARG, "rp_tune_forward: net and net->W must be non-null");
    if (net->n <= 0 || net->rows <= 0 || net->row0 < 0 || (int64_t)net->row0 + net->rows > net->n || net->ldw < net->n ||
        net->row0 % 8 != 0)
        RP_FAIL(RP_ERR_SHAPE, "rp_tune_forward: bad shard n=%d rows=%d row0=%d ldw=%lld", net->n, net->rows, net->row0,
                (long long)net->ldw);
    // only the HBM-streaming 16-byte path has anything to choose; smaller shapes keep the static rule
    if (!use_vec4(net) || !aligned16(net->W) || !streams_from_hbm(net->rows, net->n)) return RP_OK;
    if (tuned_lookup(net->rows, net->n, net->ldw, nullptr, nullptr)) return RP_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return RP_OK;  // timing needs host synchronisation: not inside a graph capture
    }
    int dev = 0;
    RP_HIP_CHECK(hipGetDevice(&dev));

    // a scratch tanh population on the caller's W (read-only): v = 0, tau = 1, k = eta = 0, source = 0
    const size_t rows = (size_t)net->rows, n = (size_t)net->n;
    float* buf = nullptr;
    if (hipMalloc(&buf, sizeof(float) * (4 * rows + 2 * n)) != hipSuccess) {
        (void)hipGetLastError();
        return RP_OK;  // no room for the scratch vectors: keep the static rule (the measurement is an optimisation)
    }
    float *v = buf, *tau = buf + rows, *k = buf + 2 * rows, *eta = buf + 3 * rows, *src[2] = {buf + 4 * rows, buf + 4 * rows + n};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipFree(buf);
    };
    if (hipMemsetAsync(buf, 0, sizeof(float) * (4 * rows + 2 * n), st) != hipSuccess ||
        hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(tau), 0x3f800000, rows, st) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        cleanup();
        RP_FAIL(RP_ERR_HIP, "rp_tune_forward: scratch set-up failed (%s)", hipGetErrorString(hipGetLastError()));
    }
    StepParams P = {};
    P.W = net->W;
    P.ldw = net->ldw;
    P.n = net->n;
    P.rows = net->rows;
    P.row0 = net->row0;
    P.p[RP_P_TAU] = tau;
    P.p[RP_P_K] = k;
    P.p[RP_P_ETA] = eta;
    P.y[RP_Y_V] = v;
    P.dt = 1e-3f;
    P.inv_dt = 1e3f;
    P.x_mode = RP_X_NONE;
    P.out_var = RP_OUT_STATE0;
    int cur = 0;
    bool failed = false;
    auto launch_probe = [&](const StreamCfg& cfg) {
        const int groups = (P.rows + cfg.R - 1) / cfg.R;
        int blocks = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
        if (cfg.blocks > 0 && blocks > cfg.blocks) blocks = cfg.blocks;
        switch (cfg.R * 10 + cfg.U) {  // (the set launch_step dispatches on)
#define RP_PROBE(R_, U_) \
    case R_ * 10 + U_: hipLaunchKernelGGL((step_kernel_tuning_probe<R_, U_>), dim3(blocks), dim3(kThreads), 0, st, P); break;
            RP_PROBE(1, 4) RP_PROBE(2, 1) RP_PROBE(2, 2) RP_PROBE(4, 1) RP_PROBE(8, 1)
            default: hipLaunchKernelGGL((step_kernel_tuning_probe<4, 2>), dim3(blocks), dim3(kThreads), 0, st, P); break;
#undef RP_PROBE
        }
    };
    auto time_cfg = [&](const StreamCfg& cfg, int steps) -> float {  // us per step
        for (int t = -1; t < steps; ++t) {  // one untimed step first
            if (t == 0) failed = failed || hipEventRecord(e0, st) != hipSuccess;
            P.src_cur = src[cur];
            P.src_next = src[cur ^ 1];
            cur ^= 1;
            launch_probe(cfg);
        }
        float ms = 0.f;
        failed = failed || hipGetLastError() != hipSuccess ||  // (a rejected launch must not look like a fast one)
                 hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                 hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f);
        return ms * 1e3f / (float)steps;
    };

    struct Timed {
        StreamCfg cfg;
        float us;
    };
    std::vector<Timed> cand;
    const StreamCfg fallback = static_cfg(net->rows, net->n);
    cand.push_back(Timed{fallback, 0.f});
    const int shapes[][2] = {{2, 1}, {4, 2}, {8, 1}, {4, 1}, {2, 2}};
    for (const auto& ru : shapes) {
        const int groups = (net->rows + ru[0] - 1) / ru[0];
        int grids[16];
        int ng = 0;
        for (int g : {256, 512, 683, 768, 1024, 2048}) grids[ng++] = g;
        for (int passes : {2, 3, 4, 6}) grids[ng++] = (groups + 4 * passes - 1) / (4 * passes);  // whole passes
        for (int q = 0; q < ng; ++q) {
            const int g = grids[q];
            if (g < 256 || (int64_t)g * 4 > groups) continue;
            bool seen = false;
            for (const Timed& c : cand) seen = seen || (c.cfg.R == ru[0] && c.cfg.U == ru[1] && c.cfg.blocks == g);
            if (!seen) cand.push_back(Timed{StreamCfg{ru[0], ru[1], g}, 0.f});
        }
    }
    for (Timed& c : cand) c.us = time_cfg(c.cfg, 3);
    std::sort(cand.begin() + 1, cand.end(), [](const Timed& a, const Timed& b) { return a.us < b.us; });
    // second look at the static rule (cand[0]) and the best few: three interleaved rounds of 8 steps, medians compared,
    // so that a drifting clock or a slow stretch of the box hits every finalist alike
    constexpr int kRounds = 3;
    const size_t finalists = std::min<size_t>(cand.size(), 5);
    float seen[5][kRounds];
    for (int r = 0; r < kRounds && !failed; ++r)
        for (size_t q = 0; q < finalists; ++q) seen[q][r] = time_cfg(cand[q].cfg, 8);
    failed = failed || hipStreamSynchronize(st) != hipSuccess;
    if (!failed)
        for (size_t q = 0; q < finalists; ++q) {
            std::sort(seen[q], seen[q] + kRounds);
            cand[q].us = seen[q][kRounds / 2];
        }
    if (const char* log = getenv("RECTIPY_AMD_TUNE_LOG"); log && log[0] == '1')
        for (size_t q = 0; q < cand.size(); ++q)
            fprintf(stderr, "[rp_tune_forward %dx%d] %s R=%d U=%d workgroups=%d: %.2f us\n", net->rows, net->n,
                    q < finalists ? "final" : "first", cand[q].cfg.R, cand[q].cfg.U, cand[q].cfg.blocks, cand[q].us);
    cleanup();
    if (failed) RP_FAIL(RP_ERR_HIP, "rp_tune_forward: timing failed (%s)", hipGetErrorString(hipGetLastError()));
    size_t best = 0;
    for (size_t q = 1; q < finalists; ++q)
        if (cand[q].us < cand[best].us * (best == 0 ? 0.99f : 1.0f)) best = q;  // the static rule keeps near-ties
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    g_tuned.push_back(TunedShape{dev, net->rows, net->n, net->ldw, cand[best].cfg, cand[best].us});
    return RP_OK;
}

int rp_forward_tuning(const rp_net* net, int32_t* rows_per_wave, int32_t* chunks_in_flight, int32_t* workgroups,
                      float* us_per_step, int32_t* measured) {
    if (!net) RP_FAIL(RP_ERR_ARG, "net is null");
    StreamCfg cfg = static_cfg(net->rows, net->n);
    float us = 0.f;
    const bool tuned = use_vec4(net) && streams_from_hbm(net->rows, net->n) &&
                       tuned_lookup(net->rows, net->n, net->ldw, &cfg, &us);
    if (rows_per_wave) *rows_per_wave = cfg.R;
    if (chunks_in_flight) *chunks_in_flight = cfg.U;
    if (workgroups) *workgroups = cfg.blocks;
    if (us_per_step) *us_per_step = us;
    if (measured) *measured = tuned ? 1 : 0;
    return RP_OK;
}

int rp_install_forward_tuning(const rp_net* net, int32_t rows_per_wave, int32_t chunks_in_flight, int32_t workgroups,
                               float us_per_step) {
    if (!net) RP_FAIL(RP_ERR_ARG, "net is null");
    if (net->n <= 0 || net->rows <= 0 || net->ldw < net->n)
        RP_FAIL(RP_ERR_SHAPE, "rp_install_forward_tuning: bad shape n=%d rows=%d ldw=%lld", net->n, net->rows,
                (long long)net->ldw);
    bool known = false;  // the (R, U) instances launch_step dispatches on
    for (int ru : {14, 21, 22, 41, 42, 81}) known = known || rows_per_wave * 10 + chunks_in_flight == ru;
    if (!known || workgroups < 0 || workgroups > (1 << 20))
        RP_FAIL(RP_ERR_ARG, "rp_install_forward_tuning: no step kernel instance for R=%d U=%d workgroups=%d",
                rows_per_wave, chunks_in_flight, workgroups);
    if (!use_vec4(net) || !streams_from_hbm(net->rows, net->n)) return RP_OK;  // such shapes keep the static rule
    int dev = 0;
    RP_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    for (TunedShape& t : g_tuned)
        if (t.device == dev && t.rows == net->rows && t.n == net->n && t.ldw == net->ldw) {
            t.cfg = StreamCfg{rows_per_wave, chunks_in_flight, workgroups};
            t.us = us_per_step;
            return RP_OK;
        }
    g_tuned.push_back(TunedShape{dev, net->rows, net->n, net->ldw, StreamCfg{rows_per_wave, chunks_in_flight, workgroups},
                                 us_per_step});
    return RP_OK;
}

int rp_clear_forward_tuning(void) {
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    g_tuned.clear();
    return RP_OK;
}

#ifdef RP_MEASUREMENT
int rp_set_tuning(int rows_per_wave, int unroll, int nontemporal, int max_blocks) {
    g_tune_R = rows_per_wave;
    g_tune_U = unroll;
    g_tune_nt = nontemporal;
    g_tune_blocks = max_blocks;
    return RP_OK;
}
#endif

int rp_stream_probe(const float* buf, int64_t n_floats, float* out, void* stream) {
    if (!buf || !out || n_floats < 4 || !aligned16(buf)) RP_FAIL(RP_ERR_ARG, "bad probe arguments");
    hipLaunchKernelGGL(stream_probe_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), buf,
                       n_floats / 4, out);
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

}  // extern "C"
