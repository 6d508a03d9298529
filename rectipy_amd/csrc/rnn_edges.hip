// rnn_edges.hip -- the MEMORY of the reference's stateful edge classes as device stages (rp_edge_mem, rp_edge_window).
//
//   LinearMemory        rectipy/edges.py:68-94     buffer = roll(buffer, -1, 1); buffer[:, delays] = x; u = buffer[:, 0]
//   LinearFilter        rectipy/edges.py:97-120    y = F y + x; u = y
//   LinearMemoryFilter  rectipy/edges.py:123-147   B1 = F roll(B); B1[:, delays] = x;  then the parent's forward on top (91-94):
//                                                  B2 = roll(B1); B2[:, delays] = B1[:, 0]; u = B2[:, 0]
//
// The projection `weights @ u` is not part of the stage: inside a coupled circuit it is a dense edge of the target's step
// kernel (rnn_forward.hip) reading u, on an acyclic edge one GEMM over the window's rows of U.
//
// Wide edges (n > RP_EMEM_NARROW_MAX: an edge between two populations) stream F with the recurrent matrix's row-dot kernel
// (rp_internal_filter_matvec: F IS a second recurrent matrix) -- one pass per step, HBM-bound like the step itself.  Narrow
// edges (an input node's few channels) are latency-bound: ONE workgroup keeps F and the state in LDS and runs all steps of a
// window in one launch; F v then is a j-ascending fmaf chain per row (include/rectipy_hip.h states both orders,
// oracle/rnn_oracle.c mirrors them).

#include "rp_common.h"

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

bool is_narrow(const rp_edge_mem* M) {
    if (M->n > RP_EMEM_NARROW_MAX) return false;
    return M->kind != RP_EMEM_DELAY_FILTER || (int64_t)M->depth * M->n <= 8192;
}

// ---- DELAY ---------------------------------------------------------------------------------------------------------------
// One row b[0 .. depth) of the buffer.  Per step: b'[c] = col_chan[c] >= 0 ? x[col_chan[c]] : b[(c + 1) % depth];  u = b'[0].
struct DelayArgs {
    const int32_t* col_chan;
    const float* ring_old;
    float* ring_new;
    const float* x;
    float* u;
    int n, depth;
};

// per-step form (inside rp_forward / rp_forward_coupled): every thread derives c0 itself, block 0 also advances the row
__global__ __launch_bounds__(256) void delay_step_kernel(const DelayArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ch0 = A.col_chan[0];
    const float c0 = ch0 >= 0 ? A.x[ch0] : A.ring_old[1 % A.depth];
    if (i < A.n) A.u[i] = c0;
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < A.depth; c += 256) {
            const int ch = A.col_chan[c];
            A.ring_new[c] = ch >= 0 ? A.x[ch] : A.ring_old[(c + 1) % A.depth];
        }
}

// window form, pass 1 (one workgroup): the row through all steps; c0[t] of every step.  depth <= 1024 columns live in LDS.
struct DelayWindowArgs {
    const int32_t* col_chan;
    float* ring;   // in: the row before the window; out: after it
    const float* X;
    int64_t ldx;
    float* U;      // c0 of step t is parked at U[t][0]; pass 2 spreads it along the row
    int64_t ldu;
    int depth, steps;
};
constexpr int kDelayMaxDepth = 4096;
__global__ __launch_bounds__(256) void delay_window_kernel(const DelayWindowArgs A) {
    __shared__ float row[2][kDelayMaxDepth];
    for (int c = threadIdx.x; c < A.depth; c += 256) row[0][c] = A.ring[c];
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < A.steps; ++t) {
        const float* x = A.X + (int64_t)t * A.ldx;
        for (int c = threadIdx.x; c < A.depth; c += 256) {
            const int ch = A.col_chan[c];
            const float v = ch >= 0 ? x[ch] : row[cur][(c + 1) % A.depth];
            row[cur ^ 1][c] = v;
            if (c == 0) A.U[(int64_t)t * A.ldu] = v;
        }
        cur ^= 1;
        __syncthreads();
    }
    for (int c = threadIdx.x; c < A.depth; c += 256) A.ring[c] = row[cur][c];
}

// window form, pass 2: U[t][i] = U[t][0] for i >= 1
__global__ __launch_bounds__(256) void delay_fill_kernel(float* U, int64_t ldu, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 1 && i < n) U[(int64_t)blockIdx.y * ldu + i] = U[(int64_t)blockIdx.y * ldu];
}

// ---- narrow FILTER / DELAY_FILTER: one workgroup, the whole window ------------------------------------------------------------
struct NarrowArgs {
    const float* F;
    int64_t ldf;
    const int32_t* col_chan;
    const float* state_in;   // FILTER: y [n]; DELAY_FILTER: B [depth x n]
    float* state_out;
    const float* X;
    int64_t ldx;
    float* U;
    int64_t ldu;
    int n, depth, steps;
};

// y' = F y + x with F v as a j-ascending fmaf chain from 0 per row; dynamic LDS: F [n x (n + 1)], y [2 x n]
__global__ __launch_bounds__(64) void filter_narrow_kernel(const NarrowArgs A) {
    extern __shared__ float lds[];
    const int n = A.n, i = threadIdx.x, pitch = n + 1;
    float* Fl = lds;
    float* y = lds + n * pitch;
    for (int q = threadIdx.x; q < n * n; q += 64) Fl[(q / n) * pitch + q % n] = A.F[(int64_t)(q / n) * A.ldf + q % n];
    if (i < n) y[i] = A.state_in[i];
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < A.steps; ++t) {
        if (i < n) {
            const float x = A.X[(int64_t)t * A.ldx + i];
            float acc = 0.f;
            for (int j = 0; j < n; ++j) acc = fmaf(Fl[i * pitch + j], y[cur * n + j], acc);
            const float v = acc + x;
            y[(cur ^ 1) * n + i] = v;
            A.U[(int64_t)t * A.ldu + i] = v;
        }
        cur ^= 1;
        __syncthreads();
    }
    if (i < n) A.state_out[i] = y[cur * n + i];
}

// LinearMemoryFilter, both phases per step; dynamic LDS: F [n x (n + 1)], B [2 x depth x n]
__global__ __launch_bounds__(256) void delay_filter_narrow_kernel(const NarrowArgs A) {
    extern __shared__ float lds[];
    const int n = A.n, depth = A.depth, pitch = n + 1, cells = depth * n;
    float* Fl = lds;
    float* B0 = lds + n * pitch;   // the buffer between steps
    float* B1 = B0 + cells;        // after phase 1
    for (int q = threadIdx.x; q < n * n; q += 256) Fl[(q / n) * pitch + q % n] = A.F[(int64_t)(q / n) * A.ldf + q % n];
    for (int q = threadIdx.x; q < cells; q += 256) B0[q] = A.state_in[q];
    __syncthreads();
    for (int t = 0; t < A.steps; ++t) {
        const float* x = A.X + (int64_t)t * A.ldx;
        for (int q = threadIdx.x; q < cells; q += 256) {  // phase 1: B1 = F roll(B0); B1[:, delays] = x
            const int c = q / n, i = q % n;
            const int ch = A.col_chan[c];
            float v;
            if (ch >= 0) {
                v = x[ch];
            } else {
                const float* col = B0 + ((c + 1) % depth) * n;
                v = 0.f;
                for (int j = 0; j < n; ++j) v = fmaf(Fl[i * pitch + j], col[j], v);
            }
            B1[q] = v;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < cells; q += 256) {  // phase 2: B0 = roll(B1); B0[:, delays] = B1[:, 0]
            const int c = q / n, i = q % n;
            const int ch = A.col_chan[c];
            const float v = ch >= 0 ? B1[ch] : B1[((c + 1) % depth) * n + i];
            B0[q] = v;
            if (c == 0) A.U[(int64_t)t * A.ldu + i] = v;
        }
        __syncthreads();
    }
    for (int q = threadIdx.x; q < cells; q += 256) A.state_out[q] = B0[q];
}

// ---- wide DELAY_FILTER: the element-wise parts around the mat-vec launches ----------------------------------------------------
struct FixArgs {
    const int32_t* col_chan;
    float* Bn;        // phase 1's result [depth x n]: the filtered columns are there already
    float* Bo;        // receives phase 2's result
    const float* x;
    float* u;
    int n, depth;
};
// B1[:, d] = x[chan(d)] for the delay columns d
__global__ __launch_bounds__(256) void delay_filter_fill_kernel(const FixArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    const int ch = A.col_chan[c];
    if (i < A.n && ch >= 0) A.Bn[(int64_t)c * A.n + i] = A.x[ch];
}
// B2 = roll(B1); B2[:, d] = B1[chan(d), 0]; u = B2[:, 0]
__global__ __launch_bounds__(256) void delay_filter_shift_kernel(const FixArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (i >= A.n) return;
    const int ch = A.col_chan[c];
    const float v = ch >= 0 ? A.Bn[ch] : A.Bn[(int64_t)((c + 1) % A.depth) * A.n + i];
    A.Bo[(int64_t)c * A.n + i] = v;
    if (c == 0) A.u[i] = v;
}

int narrow_lds_bytes(const rp_edge_mem* M) {
    const int n = M->n;
    return (int)sizeof(float) * (n * (n + 1) + 2 * (M->kind == RP_EMEM_FILTER ? n : M->depth * n));
}

// `steps` steps of a narrow FILTER / DELAY_FILTER stage in ONE launch
int narrow_window(rp_edge_mem* M, int steps, const float* X, int64_t ldx, float* U, int64_t ldu, hipStream_t st) {
    NarrowArgs A = {M->F, M->ldf, M->col_chan, M->state[M->cur], M->state[M->cur ^ 1], X, ldx, U, ldu, M->n, M->depth, steps};
    const int lds = narrow_lds_bytes(M);
    if (M->kind == RP_EMEM_FILTER) {
        hipLaunchKernelGGL(filter_narrow_kernel, dim3(1), dim3(64), lds, st, A);
    } else {
        if (rp_allow_dynamic_lds<&delay_filter_narrow_kernel>(sizeof(float) * (64 * 65 + 2 * 8192)) != RP_OK)
            RP_FAIL(RP_ERR_HIP, "dynamic LDS attribute of the narrow delay-filter stage");
        hipLaunchKernelGGL(delay_filter_narrow_kernel, dim3(1), dim3(256), lds, st, A);
    }
    M->cur ^= 1;
    return RP_OK;
}

// one step of a wide DELAY_FILTER stage: the filtered columns (one streaming mat-vec each), the delay columns, the second phase
int wide_delay_filter_step(rp_edge_mem* M, const float* x, float* u, hipStream_t st) {
    const int32_t* col_chan_host = M->col_chan_host;
    float* Bo = M->state[M->cur];
    float* Bn = M->state[M->cur ^ 1];
    const int n = M->n, depth = M->depth;
    for (int c = 0; c < depth; ++c)
        if (col_chan_host[c] < 0) {
            const int rc = rp_internal_filter_matvec(M->F, M->ldf, n, Bo + (int64_t)((c + 1) % depth) * n, nullptr, Bn + (int64_t)c * n, st);
            if (rc != RP_OK) return rc;
        }
    FixArgs A = {M->col_chan, Bn, Bo, x, u, n, depth};
    const dim3 grid((n + 255) / 256, depth);
    hipLaunchKernelGGL(delay_filter_fill_kernel, grid, dim3(256), 0, st, A);
    hipLaunchKernelGGL(delay_filter_shift_kernel, grid, dim3(256), 0, st, A);
    return RP_OK;  // (phase 2 wrote the buffer back into state[cur]: no flip)
}

}  // namespace

int rp_internal_edge_mem_check(const rp_edge_mem* M) {
    if (!M) RP_FAIL(RP_ERR_ARG, "edge_mem is null");
    if (M->kind != RP_EMEM_DELAY && M->kind != RP_EMEM_FILTER && M->kind != RP_EMEM_DELAY_FILTER)
        RP_FAIL(RP_ERR_ARG, "bad rp_edge_mem.kind %d", M->kind);
    if (M->n < 1) RP_FAIL(RP_ERR_SHAPE, "rp_edge_mem.n = %d", M->n);
    if (!M->state[0] || !M->state[1] || M->state[0] == M->state[1]) RP_FAIL(RP_ERR_ARG, "rp_edge_mem.state: two distinct buffers");
    if (!aligned16(M->state[0]) || !aligned16(M->state[1]) || (M->u && !aligned16(M->u)))
        RP_FAIL(RP_ERR_ARG, "rp_edge_mem.state / u must be 16-byte aligned");
    if (M->cur != 0 && M->cur != 1) RP_FAIL(RP_ERR_ARG, "rp_edge_mem.cur must be 0 or 1");
    if (M->save_u && M->ld_save < M->n) RP_FAIL(RP_ERR_SHAPE, "rp_edge_mem.ld_save < n");
    if (M->kind != RP_EMEM_FILTER) {
        if (M->depth < 1 || !M->col_chan || !M->col_chan_host) RP_FAIL(RP_ERR_ARG, "a delay buffer needs depth >= 1, col_chan and col_chan_host");
        if (M->kind == RP_EMEM_DELAY && M->depth > kDelayMaxDepth) RP_FAIL(RP_ERR_SHAPE, "delay buffers of at most %d columns", kDelayMaxDepth);
    }
    if (M->kind != RP_EMEM_DELAY) {
        if (!M->F || M->ldf < M->n) RP_FAIL(RP_ERR_SHAPE, "a filter needs F with ldf >= n");
        if (M->n % 4 == 0 && M->ldf % 4 == 0 && M->n > RP_EMEM_NARROW_MAX && !aligned16(M->F)) RP_FAIL(RP_ERR_ARG, "F must be 16-byte aligned");
    }
    return RP_OK;
}

static int edge_mem_step(rp_edge_mem* M, const float* x, const float** u, hipStream_t st);

// one step of the stage; `save_row`: the call's step index for rp_edge_mem.save_u (< 0: nothing is saved)
int rp_internal_edge_mem_step(rp_edge_mem* M, const float* x, const float** u, hipStream_t st, int save_row) {
    const int rc = edge_mem_step(M, x, u, st);
    if (rc != RP_OK) return rc;
    if (M->save_u && save_row >= 0)
        RP_HIP_CHECK(hipMemcpyAsync(M->save_u + (int64_t)save_row * M->ld_save, *u, sizeof(float) * (size_t)M->n, hipMemcpyDeviceToDevice, st));
    return RP_OK;
}

static int edge_mem_step(rp_edge_mem* M, const float* x, const float** u, hipStream_t st) {
    switch (M->kind) {
        case RP_EMEM_DELAY: {
            DelayArgs A = {M->col_chan, M->state[M->cur], M->state[M->cur ^ 1], x, M->u, M->n, M->depth};
            hipLaunchKernelGGL(delay_step_kernel, dim3((M->n + 255) / 256), dim3(256), 0, st, A);
            M->cur ^= 1;
            *u = M->u;
            return RP_OK;
        }
        case RP_EMEM_FILTER:
            if (is_narrow(M)) {
                float* out = M->state[M->cur ^ 1];
                const int rc = narrow_window(M, 1, x, 0, out, 0, st);  // (U row 0 = the new state itself)
                *u = out;
                return rc;
            } else {
                float* out = M->state[M->cur ^ 1];
                const int rc = rp_internal_filter_matvec(M->F, M->ldf, M->n, M->state[M->cur], x, out, st);
                M->cur ^= 1;
                *u = out;
                return rc;
            }
        default:
            *u = M->u;
            if (is_narrow(M)) return narrow_window(M, 1, x, 0, M->u, 0, st);
            return wide_delay_filter_step(M, x, M->u, st);
    }
}

extern "C" {

int rp_sizeof_edge_mem(void) { return (int)sizeof(rp_edge_mem); }

int rp_edge_window(rp_edge_mem* M, int32_t steps, const float* X, int64_t ldx, float* U, int64_t ldu, void* stream) {
    int rc = rp_internal_edge_mem_check(M);
    if (rc != RP_OK) return rc;
    if (steps < 0) RP_FAIL(RP_ERR_ARG, "steps < 0");
    if (steps == 0) return RP_OK;
    if (!X || !U || (steps > 1 && (ldx < M->n || ldu < M->n))) RP_FAIL(RP_ERR_SHAPE, "rp_edge_window: X / U with ldx, ldu >= n");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int n = M->n;
    if (M->kind == RP_EMEM_DELAY) {
        // pass 1 (one workgroup): the buffer row through all steps, c0 of step t parked at U[t][0]; pass 2 spreads it along row t
        DelayWindowArgs A = {M->col_chan, M->state[M->cur], X, ldx, U, ldu, M->depth, steps};
        hipLaunchKernelGGL(delay_window_kernel, dim3(1), dim3(256), 0, st, A);
        if (n > 1) hipLaunchKernelGGL(delay_fill_kernel, dim3((n + 255) / 256, steps), dim3(256), 0, st, U, ldu, n);
        RP_HIP_CHECK(hipGetLastError());
        return RP_OK;
    }
    if (is_narrow(M)) {
        rc = narrow_window(M, steps, X, ldx, U, ldu, st);
        if (rc != RP_OK) return rc;
        RP_HIP_CHECK(hipGetLastError());
        return RP_OK;
    }
    if (M->kind == RP_EMEM_FILTER) {  // y_t = F y_{t-1} + x_t, written straight into U's rows; the last row is the new state
        const float* prev = M->state[M->cur];
        for (int t = 0; t < steps; ++t) {
            float* row = U + (int64_t)t * ldu;
            if (n % 4 == 0 && !aligned16(row)) RP_FAIL(RP_ERR_ARG, "rp_edge_window: the rows of U must be 16-byte aligned");
            rc = rp_internal_filter_matvec(M->F, M->ldf, n, prev, X + (int64_t)t * ldx, row, st);
            if (rc != RP_OK) return rc;
            prev = row;
        }
        RP_HIP_CHECK(hipMemcpyAsync(M->state[M->cur ^ 1], prev, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, st));
        M->cur ^= 1;
        RP_HIP_CHECK(hipGetLastError());
        return RP_OK;
    }
    for (int t = 0; t < steps; ++t) {
        rc = wide_delay_filter_step(M, X + (int64_t)t * ldx, U + (int64_t)t * ldu, st);
        if (rc != RP_OK) return rc;
    }
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

}  // extern "C"
