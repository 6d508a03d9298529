/*
 * rectipy_hip.h -- C ABI of the MI355X (gfx950) integration + BPTT engine.
 *
 * This is the drop-in boundary for ONE hot path of RectiPy: the per-step
 * explicit-Euler recurrence of a dense recurrent population and its adjoint.
 * Every pointer that is not marked [host] is a DEVICE pointer into HBM owned by
 * the caller (the Python facade allocates them as torch tensors and passes
 * `tensor.data_ptr()`); the kernels borrow them.  `stream` is a hipStream_t
 * passed as void* (0 = default stream).  No torch types appear here.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   rp_forward        <- RateNet.forward            rectipy/nodes.py:166-170
 *                        SpikeResetNet.forward      rectipy/nodes.py:382-392
 *                        Spike.forward              rectipy/nodes.py:474-476
 *                        the generated vector field neuron_model_templates/
 *                          rate_neurons/leaky_integrator.yaml:10,22,33
 *                          spiking_neurons/qif.yaml:11-12
 *                        the time loop + sampling   rectipy/network.py:588-599
 *                        the input edge             rectipy/edges.py:48-49
 *   rp_forward_coupled <- the per-step graph walk over several populations joined by dense edges, feedback edges
 *                        included                   rectipy/network.py:962-981, 1322-1357 (FeedbackNetwork._backward,
 *                                                   _edge_bwd), rectipy/edges.py:48-49
 *   rp_backward_coupled <- autograd over that walk     rectipy/network.py:1016-1048 with 1322-1357
 *   rp_run.edge_mem, rp_edge_window
 *                     <- the edges with memory: LinearMemory / LinearFilter / LinearMemoryFilter.forward
 *                                                   rectipy/edges.py:68-94, 97-120, 123-147
 *   RP_MODEL_INSTANT  <- InstantNode.forward (an activation node BETWEEN populations)   rectipy/nodes.py:14-51
 *   rp_refresh_source <- (state re-initialisation)  rectipy/nodes.py:198-211,398-401
 *   rp_rate_backward, rp_qif_backward
 *                     <- torch.autograd over the unrolled loop (incl. Spike.backward, nodes.py:479-481),
 *                        Network._bptt_step         rectipy/network.py:1123-1130
 *   rp_outer_accum    <- the per-step outer-product accumulation autograd does
 *                        for `weights @ r`          (same call site)
 *   rp_rls_sweep      <- RLS.update                 rectipy/edges.py:227-234
 *   rp_forward_sharded, rp_backward_sharded, rp_bwd_prepare/partial/apply
 *                     <- no reference counterpart: row shards over the GPUs of a node
 *
 * All functions return RP_OK (0) or a negative RP_ERR_* code;
 * rp_last_error() returns a thread-local description of the last failure.
 * The Python facade maps RP_ERR_ARG -> ValueError, RP_ERR_SHAPE -> RuntimeError
 * (as torch raises on size mismatch, rectipy_tests/test_nodes.py:148-151),
 * RP_ERR_HIP / RP_ERR_COMM -> RuntimeError.
 *
 * Arithmetic: fp32 storage and fp32 arithmetic, one rounding per operation,
 * no fast-math, no contraction.  The row dot product has a FIXED summation
 * tree that does not depend on grid size or on how rows are sharded over GPUs
 * (see DESIGN.md "Summation order"), mirrored by oracle/rnn_oracle.c.
 */
#ifndef RECTIPY_HIP_H
#define RECTIPY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RP_ABI_VERSION 19
#define RP_SPARSE_END 0x2000 /* blocked packed form: last entry of a lane's row segment */

#define RP_OK 0
#define RP_ERR_ARG (-1)      /* null pointer, bad enum, bad flag combination  */
#define RP_ERR_SHAPE (-2)    /* inconsistent sizes / leading dimensions       */
#define RP_ERR_ARCH (-3)     /* no gfx950 device visible                      */
#define RP_ERR_HIP (-4)      /* HIP runtime error (text in rp_last_error)     */
#define RP_ERR_COMM (-5)     /* RCCL collective returned an error              */

/* neuron models (closed table replacing the PyRates code generator for this path) */
#define RP_MODEL_LI_TANH 0    /* leaky_integrator.yaml: li_op + tanh_op      */
#define RP_MODEL_LI_SIGMOID 1 /* leaky_integrator.yaml: li_op + sigmoid_op   */
#define RP_MODEL_QIF 2        /* qif.yaml: qif_op, threshold + reset         */
#define RP_MODEL_QIF_SFA 3    /* qif.yaml: qif_sfa_op (spike-frequency adaptation x) */
#define RP_MODEL_LIF 4        /* lif.yaml: lif_op                            */
#define RP_MODEL_IK 5         /* ik.yaml: ik_op (Izhikevich, recovery variable u) */
#define RP_MODEL_IKU 6        /* ik.yaml: iku_op (recovery driven by the population means of v and spike)   */
#define RP_MODEL_IK_BIEXP 7   /* ik.yaml: ik_biexp_op (iku + bi-exponential synapse, extra state x)          */
/* A memory-less activation node between populations (rectipy/nodes.py:14-51) as a member of rp_forward_coupled: no state, no
   recurrent matrix (rp_net.W / y / p may be NULL, n == rows, row0 == 0); its step is out = act(I) with I the node input of the
   step (external input + edge sums, as for a population) and act = rp_net.act.  The output lives in the source ping-pong:
   the step writes src[cur ^ 1] and flips, so src[cur] is the node's LAST output; an edge FROM such a member reads src[cur]
   as it stands whatever its mode (after the member's step: this step's output; before it: the previous step's).  With the
   identity and edge matrices NULL (rp_run.edge_w) it also serves as the partial sum of a long edge list (more than
   RP_MAX_EDGES edges into one node: E = ((E_0 + E_1) + ...) is continued by the next member). */
#define RP_MODEL_INSTANT 8
#define RP_MODEL_COUNT 9

#define RP_ACT_IDENTITY 0
#define RP_ACT_TANH 1    /* rp_tanhf                                            */
#define RP_ACT_SIGMOID 2 /* 1 / (1 + rp_expf(-x))                               */

#define RP_MAX_PARAMS 12
#define RP_MAX_STATE 4
#define RP_MAX_EDGES 8 /* dense edges from other vectors into one population (rp_run.edge_*) */
#define RP_MAX_MEMBERS 64 /* members of one rp_forward_coupled / rp_backward_coupled call */

/* parameter slots p[] (each a per-neuron vector of length `rows`) */
#define RP_P_TAU 0
#define RP_P_K 1
#define RP_P_ETA 2
#define RP_P_RMAX 3  /* LI_SIGMOID */
#define RP_P_SLOPE 4 /* LI_SIGMOID: s */
#define RP_P_V0 5    /* LI_SIGMOID */
#define RP_P_TAU_S 3 /* QIF, QIF_SFA, LIF, IK */
#define RP_P_ALPHA 4 /* QIF_SFA */
#define RP_P_TAU_X 5 /* QIF_SFA */
/* IK (ik.yaml:8-31): slot 0 holds C, slot 1 k, slot 2 eta, slot 3 tau_s */
#define RP_P_IK_C 0
#define RP_P_V_R 4
#define RP_P_V_THETA 5
#define RP_P_G 6
#define RP_P_E_R 7
#define RP_P_B 8
#define RP_P_TAU_U 9
#define RP_P_KAPPA 10
#define RP_P_TAU_R 11 /* IK_BIEXP: rise time; its decay time tau_d uses slot RP_P_TAU_S */

/* state slots y[] (per-neuron vectors of length `rows`) */
#define RP_Y_V 0
#define RP_Y_AUX 1 /* QIF_SFA: adaptation x; IK family: recovery u */
#define RP_Y_AUX2 2 /* IK_BIEXP: synaptic rise variable x */

/* record slots rec[] */
#define RP_REC_V 0
#define RP_REC_SOURCE 1
#define RP_REC_AUX 2
#define RP_REC_AUX2 3

/* which equation the extrinsic input enters (rp_run.x_target) */
#define RP_XT_MEMBRANE 0 /* I_ext of the v equation (every model)            */
#define RP_XT_SYNAPSE 1  /* s_ext of the s equation (LIF, lif.yaml:16)       */

/* input modes */
#define RP_X_NONE 0  /* I_ext = 0                                             */
#define RP_X_FULL 1  /* I_ext[i] = x[t*ldx + i]            (x is T x rows)    */
#define RP_X_BCAST 2 /* I_ext[i] = x[t*ldx]                (x is T x 1)       */
#define RP_X_PROJ 3  /* I_ext[i] = sum_j w_in[i*ldwin+j]*x[t*ldx+j], j<m      */

/* which variable is the node output ("out" of var_map, nodes.py:67-68) */
#define RP_OUT_STATE0 0 /* y[0]: v                                            */
#define RP_OUT_SOURCE 1 /* the presynaptic source: r=act(v) (LI) or s (QIF)   */

/*
 * A recurrent population (one "diff_eq" node, rectipy/network.py:213-306).
 * Rows of W are target neurons, columns are source neurons
 * (documentation/interfaces/model_definition.py:57).  With row sharding a rank
 * owns rows [row0, row0+rows) of the n x n matrix; per-neuron vectors are LOCAL
 * (length rows), the two source buffers are GLOBAL (length n).
 */
typedef struct rp_net {
    int32_t model;            /* RP_MODEL_*                                   */
    int32_t n;                /* global neuron count = columns of W           */
    int32_t rows;             /* local rows (== n when not sharded)           */
    int32_t row0;             /* global index of local row 0                  */
    int64_t ldw;              /* leading dimension of W in elements (>= n)    */
    const float* W;           /* [rows x ldw] row-major                       */
    const float* p[RP_MAX_PARAMS]; /* per-neuron parameter vectors [rows]     */
    float* y[RP_MAX_STATE];   /* per-neuron state vectors [rows] (in/out)     */
    float* src[2];            /* [n] presynaptic source, ping-pong (in/out)   */
    int32_t cur;              /* [host, in/out] index of the current src      */
    float dt;                 /* Euler step                                   */
    float theta;              /* QIF spike threshold (nodes.py:338)           */
    float v_reset;            /* QIF reset value     (nodes.py:338)           */
    float* scratch;           /* device scratch [>= 4 floats]; required by IKU / IK_BIEXP (population means) */
    /* Optional packed form of a FIXED coupling matrix (the reference multiplies its p = 0.1 random matrices densely,
       rectipy/utility.py:153-178): per row and per lane only the nonzeros of the lane's four columns per chunk, in
       the order the dense kernel visits them, four entries per lane per group, so that rp_forward gives bit-identical
       results while streaming 6 bytes per stored entry (layout: rectipy_amd/sparse.py; n % 4 == 0).  When sp_vals is
       non-null rp_forward reads it, not W. */
    const float* sp_vals;     /* [groups x 256] entry 4k+e of lane l at (group k) * 256 + 4 l + e; 0.0 = padding    */
    const uint16_t* sp_cidx;  /* [groups x 256] code of each entry: (component << 14) | column chunk m                */
    const int32_t* sp_rowoff; /* [rows + 1] first group of each row (windowed form: [slots * sp_phases + 1])        */
    /* Windowed form for n > 32768 (sparse.py: pack_rows_windowed), selected by sp_row_ids != NULL: the source vector
       is staged in LDS one 32768-column window at a time; rows are regrouped into batches of 64 slots that start their
       sweep in the same window, a lane's entries are stored per phase and codes hold window-relative chunks. */
    const int32_t* sp_row_ids;      /* [sp_batches * 64] row of each slot, -1 = padding                               */
    const int32_t* sp_batch_class;  /* [sp_batches] start window of a batch                                           */
    const int32_t* sp_phase_window; /* [windows * sp_phases] window staged in phase ph of class c, -1 = none           */
    int32_t sp_phases;
    int32_t sp_batches;
    /* Blocked form for n <= 32768 (sparse.py: pack_rows_blocked), selected by sp_block_rows > 0: sp_block_rows
       consecutive rows share one padded stream (sp_rowoff = [blocks + 1]); bit 13 of a code (RP_SPARSE_END) closes the
       lane's segment of a row, chunk indices take 13 bits. */
    int32_t sp_block_rows;
    int32_t act;              /* RP_MODEL_INSTANT: RP_ACT_*                                                             */
} rp_net;

/*
 * One call = `steps` Euler steps (the body of the loop at network.py:588-599).
 * Absolute step index of step t is step0 + t; it drives cutoff / sampling
 * exactly as the reference does (record iff step >= cutoff and
 * step % sampling_steps == 0, storing the mean of the buffered outputs).
 */
/* A matrix in the packed sparse form of rectipy_amd/sparse.py (the fields of rp_net.sp_* for a matrix other than W). */
typedef struct rp_packed {
    const float* vals;             /* NULL: no packed form                                                   */
    const uint16_t* cidx;
    const int32_t* rowoff;
    const int32_t* row_ids;        /* windowed form only                                                     */
    const int32_t* batch_class;
    const int32_t* phase_window;
    int32_t phases, batches;
    int32_t block_rows;            /* > 0: blocked form                                                      */
} rp_packed;

/*
 * The MEMORY of an edge (the reference's edge classes with state, rectipy/edges.py:68-147): a stage in front of the edge's
 * projection.  Every step it takes the vector x the edge reads from its source (length n) and yields the vector u the
 * projection `weights @ u` is applied to:
 *   RP_EMEM_DELAY         LinearMemory (edges.py:91-94)        buffer = roll(buffer, -1, 1); buffer[:, delays] = x; u = buffer[:, 0]
 *   RP_EMEM_FILTER        LinearFilter (edges.py:118-120)      y = F y + x;  u = y
 *   RP_EMEM_DELAY_FILTER  LinearMemoryFilter (edges.py:143-146, then its parent's forward 91-94 ON TOP: the class steps its
 *                         buffer twice per call)               B1 = F roll(B); B1[:, delays] = x;
 *                                                              B2 = roll(B1); B2[:, delays] = B1[:, 0]; u = B2[:, 0]
 * `buffer[:, delays] = x` writes x[j] into column delays[j] of EVERY row (and of a repeated delay the last j wins: col_chan).
 * DELAY therefore keeps ONE row of the buffer (its rows are identical from the first `depth` steps on, whatever they held;
 * a buffer whose rows differ is not representable here) and u = c0 * ones(n), c0 = the new column 0.
 * Arithmetic of F v (per output row i): n > RP_EMEM_NARROW_MAX: the row dot of the recurrent matrix (summation tree of the
 * 16-byte path iff n % 4 == 0 and ldf % 4 == 0, start chunk from row i); n <= RP_EMEM_NARROW_MAX (DELAY_FILTER: and
 * depth * n <= 8192): a j-ascending fmaf chain from 0.  FILTER then adds x[i] (one addition: fl(fl(F y) + x)).
 * State: two buffers in turn, `cur` names the current one (host, in/out).  DELAY: [depth] floats each; FILTER: [n];
 * DELAY_FILTER: [depth x n], column c of the reference's buffer at state + c * n.  All 16-byte aligned.
 */
#define RP_EMEM_NONE 0
#define RP_EMEM_DELAY 1
#define RP_EMEM_FILTER 2
#define RP_EMEM_DELAY_FILTER 3
#define RP_EMEM_NARROW_MAX 64

typedef struct rp_edge_mem {
    int32_t kind;             /* RP_EMEM_*                                                                           */
    int32_t n;                /* n_in: length of x and u                                                              */
    int32_t depth;            /* DELAY*: columns of the delay buffer = largest delay + 1                              */
    const int32_t* col_chan;  /* DELAY*: [depth] the input channel written into buffer column c every step, -1 = none  */
    const int32_t* col_chan_host; /* [host] the same array (which columns are filtered decides the launches of a step)  */
    const float* F;           /* FILTER*: [n x ldf] row-major                                                         */
    int64_t ldf;
    float* state[2];
    int32_t cur;              /* [host, in/out]                                                                       */
    float* u;                 /* [n] DELAY / DELAY_FILTER: receives u every step (rp_run.edge_mem; unused by rp_edge_window).
                                 FILTER: unused, u is the new state                                                   */
    float* save_u;            /* optional [steps x ld_save] (rp_run.edge_mem only): row t receives the u of step t -- what the
                                 edge's projection read; the edge's adjoint and weight gradient need it (rp_backward_coupled) */
    int64_t ld_save;
} rp_edge_mem;
int rp_sizeof_edge_mem(void);

/* `steps` consecutive steps of the stage on the rows of X [steps x ldx] into the rows of U [steps x ldu] (an ACYCLIC edge:
   its source's whole window is known, the projection of the window then is one GEMM of the caller's).  The state after the
   call is what `steps` calls of the reference's forward leave behind.  Narrow edges: ONE launch for the whole window. */
int rp_edge_window(rp_edge_mem* mem, int32_t steps, const float* X, int64_t ldx, float* U, int64_t ldu, void* stream);

typedef struct rp_run {
    int32_t steps;
    int64_t step0;            /* absolute index of the first step             */
    int32_t x_mode;           /* RP_X_*                                       */
    int32_t m;                /* RP_X_PROJ: input dimension                   */
    const float* x;           /* inputs, row t at x + t*ldx                   */
    int64_t ldx;
    const float* w_in;        /* RP_X_PROJ: [rows x ldwin]                    */
    int64_t ldwin;
    int32_t out_var;          /* RP_OUT_*                                     */
    int32_t x_target;         /* RP_XT_*                                      */
    int32_t cutoff;
    int32_t sampling_steps;   /* >= 1                                         */
    int32_t max_samples;      /* rows available in samples / rec[]            */
    float* traj;              /* optional [steps x ldtraj]: pre-update output */
    int64_t ldtraj;
    float* save_state[RP_MAX_STATE]; /* optional [steps x ldsave]: pre-update y[j] */
    float* save_src;          /* optional [steps x ldsave]: pre-update source (local rows) */
    float* save_acc;          /* optional [steps x ldsave]: (W @ src)_i       */
    int64_t ldsave;
    float* acc;               /* [rows] running sum of buffered outputs (in/out); required if samples.  Read only
                                 while acc_count > 0: with acc_count == 0 the first buffered output overwrites it */
    int32_t acc_count;        /* [host, in/out] buffered outputs in acc       */
    int32_t sample_idx;       /* [host, in/out] next sample row               */
    float* samples;           /* optional [max_samples x ldsamples]           */
    int64_t ldsamples;
    float* rec[RP_MAX_STATE]; /* optional [max_samples x ldsamples], slots RP_REC_*: recorded state at sample steps
                                 (LI: post-update, network.py:147-148; spiking: pre-update, nodes.py:387-392) */
    int32_t* spike_events;    /* optional QIF: (step, global index) int32 pairs */
    int32_t spike_capacity;   /* number of pairs that fit                     */
    int32_t* spike_count;     /* device counter (keeps counting past capacity) */
    /* row t receives the FULL source vector the step read (all n entries, i.e. r_t / s_t
       before the update) -- the B operand of the shard's rectangular dW GEMM. */
    float* save_src_full;     /* [steps x ldfull] or NULL                     */
    int64_t ldfull;
    /* Spiking models, optional [rows]: the FIRST step of the call detects spikes (and applies the reset mask) on these
       values instead of on v -- MultiSpikeResetNet keeps its own spike-detector copy of the reset variable, which
       starts at zero and is not refreshed by reset() (rectipy/nodes.py:428-431, 451-456).  Later steps of the call
       use v.  Not available for the population-mean models (IKU, IK_BIEXP). */
    const float* spike_src;
    /* Dense edges INTO this population evaluated inside its step (the reference's `edge.forward(source output)` summed
       into the node input, rectipy/network.py:962-977, 1322-1357, rectipy/edges.py:48-49): at step t edge e adds
       E_e[i] = sum_j edge_w[e][i * edge_ld[e] + j] * s_e[j], j < edge_n[e], with s_e = edge_src[e] + t * edge_src_ld[e]
       (edge_src_ld = 0: the same vector every step; rp_forward_coupled sets the vector of every step itself).  Each
       E_e[i] is a row dot product with the summation tree of the recurrent one (16-byte path iff edge_n % 4 == 0 and
       edge_ld % 4 == 0 -- then edge_w, edge_src and edge_src_ld must be 16-byte aligned --, start chunk from the global
       row index row0 + i and edge_n).  The node input of the step then is
           I[i] = x-input[i] + ((E_0[i] + E_1[i]) + ...)      (just the edge sum when x_mode == RP_X_NONE),
       the order in which the front end adds a node's edge inputs.  The matrices are streamed from HBM next to W by the
       same wave that owns the row: one launch per population and step, whatever the number of edges.  Dense W only
       (rp_net.sp_vals is ignored when n_edges > 0; the packed form gives the same bits as the dense one anyway).
       edge_w[e] == NULL: the IDENTITY edge, E_e[i] = s_e[i] (edge_n[e] == rows) -- the input of an activation-node member
       from a member of its own width, or the running sum of a long edge list (RP_MODEL_INSTANT). */
    int32_t n_edges;                       /* 0..RP_MAX_EDGES                                                   */
    const float* edge_w[RP_MAX_EDGES];     /* [rows x edge_ld]                                                  */
    int64_t edge_ld[RP_MAX_EDGES];
    int32_t edge_n[RP_MAX_EDGES];          /* columns of the edge = length of its source vector                 */
    const float* edge_src[RP_MAX_EDGES];   /* source vector of the call's first step (rp_forward only)          */
    int64_t edge_src_ld[RP_MAX_EDGES];     /* added per step (rp_forward only)                                  */
    /* optional, [rows] each, all three or none: a ring holding the population's OUTPUT (rp_run.out_var) of the last steps --
       hist[a % 3] = the pre-update output of the step with absolute index a.  The RP_EDGE_OUT / RP_EDGE_LAG edges of
       rp_forward_coupled read it (and RP_EDGE_LIVE ones when the members' steps share one launch, see there).  Three
       slots so that, in the shared launch, step a can store the output of step a + 1 (its post-update value) while
       other members still read those of steps a and a - 1. */
    float* hist[3];
    /* optional [steps x ldsave]: the node input I_t of every step -- the external input plus the edge sums, as the epilogue
       received it.  The adjoint of a spiking population with edges needs it where it otherwise takes the window's external
       input (rp_bwd.x, RP_X_FULL): rp_backward_coupled. */
    float* save_in;
    /* rp_forward_coupled only: packed forms of the edge matrices (rows = this population's rows, row0 = 0, n = edge_n) and a
       workspace [rows].  A member whose edges ALL have a packed form is stepped as n_edges packed mat-vec launches --
       E = (E_0 + E_1) + ..., then I = x + E into edge_sum -- followed by its own step (packed when rp_net.sp_vals is set,
       else the dense one) reading I as a full-width input: the dense arithmetic bit for bit (the packed layout reproduces
       the dense row dot), at the bytes of the stored nonzeros.  Otherwise (any edge without a packed form, or no edge_sum)
       the dense edge matrices are streamed as described above. */
    rp_packed edge_sp[RP_MAX_EDGES];
    float* edge_sum;
    /* Edges with memory (kind != RP_EMEM_NONE): every step the stage is advanced with the vector the edge reads (edge_src row /
       the source member's vector by edge_mode) right before the population's step, and the edge's product is taken with the
       stage's output u instead; edge_n[e] == edge_mem[e].n.  The stage costs its own launches (FILTER: one streaming pass
       over F, the recurrent matrix of the edge). */
    rp_edge_mem edge_mem[RP_MAX_EDGES];
} rp_run;

/* what an edge of rp_forward_coupled reads from its source population q (q's output variable, rp_run.out_var) */
#define RP_EDGE_LIVE 0 /* as it stands when the target is stepped: y[0] or src[cur] of q.  A feedback edge from a rate
                          population (rectipy/network.py:1354-1357 reads `source["out"]`): the state before q's step of
                          this Euler step if q comes later in the order, after it if q comes earlier */
#define RP_EDGE_OUT 1  /* what q's step of THIS Euler step returned (its pre-update output): a feed-forward edge, or a
                          feedback edge from a spiking population stepped earlier.  q must come earlier in the order;
                          reads q's src buffer of the step (RP_OUT_SOURCE) or q's hist[a % 3] (RP_OUT_STATE0: required) */
#define RP_EDGE_LAG 2  /* q's pre-update output of the PREVIOUS Euler step, hist[(a - 1) % 3] of q (required): a feedback
                          edge from a spiking population that has not been stepped yet -- its exposed state lags one step
                          (rectipy/nodes.py:387-392).  Before the first step of a call the caller stores there what the
                          edge shall read at that step */

/* One population of a coupled circuit: its rp_net, its rp_run (inputs, sampling, trajectories, edge matrices) and, per
   edge of the run, which member the edge reads and how. */
typedef struct rp_member {
    rp_net* net;
    rp_run* run;
    int32_t edge_source[RP_MAX_EDGES]; /* index of the source member                                           */
    int32_t edge_mode[RP_MAX_EDGES];   /* RP_EDGE_*                                                            */
} rp_member;

/* Arguments of the adjoint sweep over a stored window (LI models). */
typedef struct rp_bwd {
    int32_t steps;            /* window length T                              */
    const float* traj_v;      /* [T x ld] pre-update v_t (save_state[0])      */
    const float* traj_r;      /* [T x ld] pre-update r_t (save_src)           */
    const float* traj_acc;    /* optional [T x ld] (W r_t)_i, needed for grad_k */
    int64_t ld;
    int32_t grad_var;         /* RP_OUT_*: variable grad_out refers to        */
    const float* grad_out;    /* [T x ldg] dL/d out_t                         */
    int64_t ldg;
    float* lam;               /* [n] adjoint state lambda (in: lambda_T, out: lambda_0 = dL/dy0) */
    float* u_traj;            /* [T x ld] out: u_{t+1} = k * lambda_{t+1} rows (row t), feeds rp_outer_accum */
    float* partials;          /* workspace [rp_bwd_partial_rows(n) x n]       */
    float* grad_tau;          /* optional [n] accumulated (+=)                */
    float* grad_k;            /* optional [n] accumulated (+=)                */
    float* grad_eta;          /* optional [n] accumulated (+=)                */
    float* grad_x;            /* optional [T x ldgx] dL/dI_ext,t              */
    int64_t ldgx;
    /* spiking (QIF) sweep only; ignored by rp_rate_backward */
    float* lam2;              /* [n] adjoint of s (in: at T, out: at 0); traj_r then holds s_t */
    float* grad_tau_s;        /* optional [n] accumulated (+=)                */
    float slope;              /* surrogate slope: d spike / d x = 1 / (1 + slope |x|)^2, x = v - theta (nodes.py:479-481) */
    int32_t x_mode;           /* RP_X_NONE / RP_X_FULL / RP_X_BCAST: the input of the forward window (needed for grad_tau) */
    const float* x;
    int64_t ldx;
    int32_t x_target;         /* RP_XT_*: which equation the window's input entered (LIF)                     */
    const float* traj_aux;    /* [T x ld] pre-update second state (x of QIF_SFA, u of IK) = save_state[1]     */
    float* lam3;              /* [n] adjoint of the second state (in: at T, out: at 0)                        */
    float* grad_p[RP_MAX_PARAMS]; /* optional per-neuron parameter gradients by slot RP_P_* (+=); the named
                                     fields above take precedence for tau / k / eta / tau_s                   */
    /* Optional workspace for a transposed copy of W (unsharded sweeps).  When it is given and
       n <= RP_BWD_TRANSPOSE_MAX_N, the sweep first writes W^T there (one pass, n^2 floats) and then computes
       z = W^T u with the FORWARD's row-dot kernel on the rows of W^T, fused with the element-wise adjoint update:
       ONE launch per step instead of two -- at these sizes the step is launch/latency-bound, not HBM-bound.
       The summation order of z then is the forward's row order; the oracle mirrors the same rule. */
    float* wt;                /* [n x ldwt] or NULL                                                            */
    int64_t ldwt;
    /* population-mean models (IKU, IK_BIEXP; ik.yaml:34-68) */
    const float* traj_aux2;   /* IK_BIEXP: [T x ld] pre-update x (rise variable of the synapse) = save_state[2] */
    float* lam4;              /* IK_BIEXP: [n] adjoint of x (in: at T, out: at 0)                              */
    float* means;             /* IKU / IK_BIEXP: scratch [4] for the per-step population sums                  */
    /* Optional packed form of W^T for a FIXED sparse matrix (rectipy_amd/sparse.py applied to the transposed matrix,
       unsharded sweeps): when spt_vals is non-null, z = W^T u is the row dot of the packed rows of W^T with u -- the
       summation order of the transposed-copy sweep at any n -- fused with the element-wise update, one launch per step,
       and neither W nor wt is read. */
    const float* spt_vals;
    const uint16_t* spt_cidx;
    const int32_t* spt_rowoff;
    /* windowed form of the packed W^T for n > 32768 (as rp_net.sp_row_ids ...), selected by spt_row_ids != NULL */
    const int32_t* spt_row_ids;
    const int32_t* spt_batch_class;
    const int32_t* spt_phase_window;
    int32_t spt_phases;
    int32_t spt_batches;
    int32_t spt_block_rows;   /* > 0: blocked form of the packed W^T (as rp_net.sp_block_rows) */
} rp_bwd;

#define RP_BWD_TRANSPOSE_MAX_N 4096

int rp_abi_version(void);
const char* rp_last_error(void);

/* (v19) At launch-bound sizes (dense W, no edges, rows == n, n <= 1024 -- n <= 1536 when n % 4 == 0 --, 4 steps or more) rp_forward and -- on the
   transposed-copy sweep -- rp_rate_backward / rp_qif_backward run ALL their steps in ONE launch (rectipy_amd/csrc/rp_resident.h; bit-identical
   to a launch per step; RECTIPY_AMD_RESIDENT=0 in the environment selects the launches).  Such a launch is as asynchronous as the
   launches it replaces; the one thing it can report only afterwards is a wave that timed out waiting for its peers (seconds: the
   GPU never ran the launch's workgroups together).  The library's next rp_forward / rp_rate_backward call returns RP_ERR_HIP
   then; a caller that has synchronised the stream asks here: RP_OK, or RP_ERR_HIP once per such launch (its results are not valid;
   reference: the time loops of rectipy/network.py:542-601 and 1016-1048, which this replaces step for step). */
int rp_async_status(void);

/* sizeof the three argument structs as compiled (lets a binding verify its layout) */
int rp_sizeof_net(void);
int rp_sizeof_run(void);
int rp_sizeof_bwd(void);
int rp_sizeof_member(void);

/* RP_OK iff HIP device `device` exists and is gfx950. */
int rp_device_check(int device);
/* The device's marketing name (hipDeviceProp_t::name) into buf[len], NUL-terminated: the key under which measured walks
   are persisted.  (Asking torch costs a process 105 ms once: torch.cuda.get_device_name initialises amdsmi.) */
int rp_device_name(int device, char* buf, int32_t len);

/* src[cur][row0+i] = act(y[0][i]) for LI models (no-op for QIF, whose source is a state). */
int rp_refresh_source(const rp_net* net, void* stream);

/* `run->steps` fused Euler steps; asynchronous on `stream`. */
int rp_forward(rp_net* net, rp_run* run, void* stream);

/*
 * `count` (1..RP_MAX_MEMBERS) unsharded populations joined by dense edges, cycles allowed: every Euler step steps the members in array order
 * (the front end's topological order of the feed-forward edges), one launch per member, inside ONE host call for the
 * whole window -- what the reference does with one Python graph walk per step (rectipy/network.py:588-599, 962-981,
 * 1322-1357).  All runs must agree on steps and step0; each member samples / records / stores trajectories through its
 * own rp_run exactly as under rp_forward.  Edge e of member m reads member edge_source[e] in mode edge_mode[e];
 * edge_n[e] must be that member's n.  A member may read itself only in RP_EDGE_LAG, or in RP_EDGE_LIVE when its output
 * variable is the source vector.  Row shards, spike_src and save_src_full are not available here.
 *
 * ONE launch per Euler step for all members (instead of one per member) when every edge reads a value from BEFORE the
 * step -- no RP_EDGE_LIVE edge from a member stepped earlier --, at most four members of at most 4096 neurons each (the
 * launch-bound sizes: the reference's own scale), none with population-mean coupling, no RP_MODEL_INSTANT member, no identity
 * edge, no edge with memory, at most four edges into a member, and every member that is read
 * through its state variable has the hist ring: each member then also stores its post-update output as the next step's
 * ring entry, and edges read the ring instead of a state vector that the same launch updates.  Same arithmetic, same
 * bits as the launch per member (N = 2 x 2048: 13 instead of 21 us per step).
 *
 * ALL steps of the call in ONE launch (r05) for such circuits when, in addition, every member's W and every edge matrix is on
 * the 16-byte path with at most 1024 columns, no W is packed, (1 + edges into a member) x (1 .. 4 chunks of 256 columns, by the
 * widest matrix of the circuit) <= 8 for every member, and steps >= 4: the rows of a member's own and in-edge matrices stay in
 * registers, the members' vectors go from wave to wave as tagged granules (the resident launches described at rp_async_status,
 * which also reports their one failure mode).  Same bits (2 x 256 neurons: 3.2 instead of 6.5 us per step);
 * RECTIPY_AMD_RESIDENT=0 selects the launch per step.
 */
int rp_forward_coupled(rp_member* members, int32_t count, void* stream);

/* Rows of the partial-sum workspace needed by rp_rate_backward for n neurons. */
int32_t rp_bwd_partial_rows(int32_t n);

/* Adjoint sweep t = T-1..0 over a stored window (unsharded LI models). */
int rp_rate_backward(const rp_net* net, const rp_bwd* bwd, void* stream);

/*
 * Row-sharded adjoint (leaky-integrator models; spiking models QIF, QIF_SFA, LIF, IK), one step at a time (W^T(k*lambda) from row shards
 * gives full-length partial sums -> reduce-scatter).  Per-neuron vectors of rp_bwd are LOCAL (length net->rows);
 * partials needs rp_bwd_shard_partial_rows(n, rows) x n floats.  Sequence:
 *   rp_bwd_prepare;  for t = T-1 .. 0:  rp_bwd_partial(t, z_full[n]) -> caller reduce-scatters z_full over the ranks
 *   -> rp_bwd_apply(t, z_local[rows]).
 */
int32_t rp_bwd_shard_partial_rows(int32_t n, int32_t rows);
int rp_bwd_prepare(const rp_net* net, const rp_bwd* bwd, void* stream);
int rp_bwd_partial(const rp_net* net, const rp_bwd* bwd, int32_t t, float* z_full, void* stream);
int rp_bwd_apply(const rp_net* net, const rp_bwd* bwd, int32_t t, const float* z_local, void* stream);

/*
 * The adjoint of rp_forward_coupled: one sweep t = T-1 .. 0 over the stored windows of ALL members, the gradient passing
 * through the dense edges between them (what autograd does over the reference's step-by-step graph walk,
 * rectipy/network.py:1322-1357 under Network._bptt, network.py:1016-1048).  Member m's edges are the edges INTO m, as in
 * the forward call.  Per adjoint step, for every edge e (q -> m) the transposed product c = W_e^T gx_m(t) -- gx_m(t) =
 * dL/dI_m(t), row t of m's rp_bwd.grad_x, which the members' own sweeps produce -- is added to row tau of q's
 * rp_bwd.grad_out (the gradient with respect to q's output, rp_bwd.grad_var = q's output variable), tau = t for
 * RP_EDGE_OUT and for RP_EDGE_LIVE from a member LATER in the order, tau = t - 1 for RP_EDGE_LAG (nothing for tau < 0:
 * the window's history is cut there, as by Network.detach); then every member's own step (rp_bwd_partial / rp_bwd_apply:
 * z = W^T u over column strips, the element-wise update).  So rp_bwd.grad_out of a member is READ AND WRITTEN here (the
 * caller passes a private buffer, zeros where the loss has no direct gradient), grad_x is required for every member with
 * edges.  An RP_EDGE_LIVE edge from an EARLIER member q reads q's output AFTER q's step of the same Euler step, i.e. the
 * output of step t + 1: its c = W_e^T gx_m(t) goes into row t + 1 of q's output gradient -- for t + 1 = steps into q's final
 * adjoint lam (times act'(r_final) when the output variable is the source: rp_bwd_member.final_src).  To have gx_m(t) in
 * time the members take every adjoint step (and the preparation) in REVERSE order, each right after the edge products
 * that end in its own output gradient.  c is summed like z: per chunk of RC(n_q, rows_m) rows of the edge matrix ascending,
 * chunks ascending; the edges that end in one member's output gradient are added in member / edge order (mirrored by
 * oracle_backward_coupled).  The weight gradient of an edge is left to the caller:
 * dW_e = sum_t gx_m(t) (x) out_q(tau) is rp_outer_store_rect on the two stored windows.
 */
typedef struct rp_bwd_member {
    const rp_net* net;
    const rp_bwd* bwd;                   /* as for rp_bwd_prepare / partial / apply (unsharded: rows == n); see above for grad_out / grad_x */
    int32_t n_edges;
    const float* edge_w[RP_MAX_EDGES];   /* [rows x edge_ld]: rows = this member's neurons, columns = the source's        */
    int64_t edge_ld[RP_MAX_EDGES];
    int32_t edge_source[RP_MAX_EDGES];
    int32_t edge_mode[RP_MAX_EDGES];     /* RP_EDGE_OUT | RP_EDGE_LIVE (source later in the order) | RP_EDGE_LAG          */
    float* edge_work;                    /* workspace: SUM over this member's edges of rp_bwd_shard_partial_rows(n_source, rows) * n_source floats (16-byte aligned) */
    float* z;                            /* workspace [n]                                                               */
    rp_packed edge_spt[RP_MAX_EDGES];    /* optional: the TRANSPOSED edge matrix [n_source x rows] in packed form (a fixed sparse edge).
                                            Then c = W_e^T gx is formed by packed row dots -- c[j] = the forward's row dot of row j of
                                            W_e^T with gx (its summation tree on the width `rows`, start chunk from j) -- instead of
                                            chunk partials: other rounding than the dense form, mirrored by the oracle            */
    const float* final_src;              /* [n] the member's source vector r at the END of the window; required when an RP_EDGE_LIVE
                                            edge of a LATER member reads this member (a post-update read) and its output variable
                                            is the source; else may be NULL                                               */
    /* v18: the adjoint through an edge's MEMORY (rp_run.edge_mem of the forward call; none for RP_EMEM_DELAY_FILTER).  With
       c(t) = W_e^T gx_m(t) as for a plain edge (row / tau by edge_mode as above; not for the post-update read of an RP_EDGE_LIVE
       edge from an earlier member):
         RP_EMEM_FILTER  y_t = F y_{t-1} + x_t:  ly(t) = c(t) + F^T ly(t + 1)  (ly = 0 behind the window), and ly(t) -- not c(t) --
                         is added to row tau of the source's output gradient.  F^T ly is summed like every transposed product
                         (chunks of RC(n, n) rows of F ascending, chunk partials ascending), ly(t) = (chunk sum of c) + (chunk sum
                         of F^T ly); edge_lam [n_source] carries ly (in: zeros; out: ly of the window's first step)
         RP_EMEM_DELAY   u_t = x_{t - d}[j] * ones(n) with d = edge_delay (the smallest delay) and j = edge_chan (its channel):
                         the TOTAL of c(t) -- per-thread chains over q = tid + 1024 m of the columns' chunk sums, then a
                         1024-leaf halving tree -- is added to element j of row tau - d of the source's output gradient (rows
                         before the window: dropped).
       The weight gradient of such an edge is dW_e = sum_t gx_m(t) (x) u_t on the window the forward call saved
       (rp_edge_mem.save_u); a masked edge streams weights * mask in both directions (its gradient: the plain one times the mask). */
    int32_t edge_mem_kind[RP_MAX_EDGES]; /* RP_EMEM_NONE | RP_EMEM_FILTER | RP_EMEM_DELAY                                   */
    const float* edge_F[RP_MAX_EDGES];   /* FILTER: [n_source x edge_ldf]                                                   */
    int64_t edge_ldf[RP_MAX_EDGES];
    float* edge_lam[RP_MAX_EDGES];       /* FILTER: [n_source], 16-byte aligned                                             */
    float* edge_fwork[RP_MAX_EDGES];     /* FILTER: workspace rp_bwd_shard_partial_rows(n_source, n_source) * n_source floats */
    int32_t edge_delay[RP_MAX_EDGES];    /* DELAY: d                                                                        */
    int32_t edge_chan[RP_MAX_EDGES];     /* DELAY: j                                                                        */
} rp_bwd_member;
/* An RP_MODEL_INSTANT member (an activation node between populations) in rp_backward_coupled: its rp_bwd carries steps,
   traj_r = the node's outputs of the window ([steps x ld], rp_run.traj of the forward call), grad_out (read and written like
   a population's) and grad_x; its adjoint step is gx(t)[i] = grad_out(t)[i] * act'(out(t)[i]) -- act' = 1 (identity),
   1 - o^2 (tanh: computed as fmaf(-o, o, 1)), o * (1 - o) (sigmoid) -- taken, like a population's step, right after the edge
   products that end in its output gradient.  No state, no parameters; lam / u_traj / partials / z are not used. */
int rp_sizeof_bwd_member(void);
int rp_backward_coupled(const rp_bwd_member* members, int32_t count, void* stream);

/* Surrogate-gradient adjoint sweep of a spiking step (QIF, QIF_SFA, LIF, IK): gradient flows through the Heaviside
   via Spike.backward (nodes.py:479-481) but not through the reset mask (nodes.py:384). */
int rp_qif_backward(const rp_net* net, const rp_bwd* bwd, void* stream);

/* C[n x n] (ldc) += alpha * A^T B with A = [T x n] (lda), B = [T x n] (ldb): MFMA f32 GEMM. */
int rp_outer_accum(int32_t n, int32_t T, float alpha, const float* A, int64_t lda, const float* B,
                   int64_t ldb, float* C, int64_t ldc, void* stream);
/* Rectangular form for a row shard: C[m x n] += alpha * A^T B with A = [T x m], B = [T x n]. */
int rp_outer_accum_rect(int32_t m, int32_t n, int32_t T, float alpha, const float* A, int64_t lda, const float* B,
                        int64_t ldb, float* C, int64_t ldc, void* stream);
/* Store form: C[m x n] = alpha * A^T B -- C is written, never read (no zero-fill pass before the GEMM, no read of C in
   its epilogue).  Bits of rp_outer_accum_rect into a zeroed C (each element is stored as 0 + alpha * sum). */
int rp_outer_store_rect(int32_t m, int32_t n, int32_t T, float alpha, const float* A, int64_t lda, const float* B,
                        int64_t ldb, float* C, int64_t ldc, void* stream);

/*
 * Row shards over the GPUs of a node with the per-step exchange INSIDE the library: the
 * collectives are RCCL calls enqueued on the same stream as the step kernels, so a whole run / adjoint sweep is
 * one host call per rank and there is no cross-stream hand-over per step.  The library does not link RCCL: a
 * communicator is only valid inside the RCCL instance that created it (PyTorch bundles its own copy), so the caller
 * passes the addresses of ncclAllGather / ncclReduceScatter of THAT instance next to the ncclComm_t.
 */
typedef struct rp_comm {
    void* comm;            /* ncclComm_t spanning the row shards, rank p owning rows [p*rows, (p+1)*rows)   */
    void* all_gather;      /* ncclResult_t (*)(const void*, void*, size_t, int dtype, ncclComm_t, hipStream_t) */
    void* reduce_scatter;  /* ncclResult_t (*)(const void*, void*, size_t, int dtype, int op, ncclComm_t, hipStream_t) */
    int32_t rank, world;
    float* work;           /* sharded adjoint only: n + rows floats (full-length partial, reduced slice)     */
    /* mode = RP_COMM_DIRECT (forward only; opt-in): no collective launch.  After its step kernel a rank PUSHES its slice of
       the next source vector straight into every peer's source buffer (peer memory mapped through hipIpc, written over
       xGMI by one small copy kernel) and then its STREAM writes the step's sequence number into its slot of every peer's
       flag array (hipStreamWriteValue32); before a step kernel the stream waits until every peer's slot holds the previous
       sequence number (hipStreamWaitValue32, >=).  No CU-resident spin loop anywhere: a rank that is late only delays
       the command processors of its peers.  The wait covers both hazards of the two ping-pong buffers: a peer's slices of
       step s-1 have landed in this rank's current buffer, and the peer is done reading the buffer this rank is about to
       push into.  Summation order untouched: results equal the RCCL / host-loop exchange bit for bit.  `seq` is the
       number of steps exchanged so far on these buffers (in/out; flags must be zero while seq is 0).
       comm / all_gather may be NULL in this mode; rp_backward_sharded needs them unless peer_inbox is given (below). */
    int32_t mode;                 /* RP_COMM_RCCL (0) | RP_COMM_DIRECT (1)                                          */
    uint32_t seq;                 /* [host, in/out]                                                                 */
    float* const* peer_src;       /* [2 * world] (host array): peer p's source buffer k at [2 p + k]; own entries = net->src[k] */
    uint32_t* const* peer_flags;  /* [world] (host array): peer p's flag array (world words), mapped here; own entry = flags */
    uint32_t* flags;              /* this rank's flag array [world], slot q written by rank q                       */
    /* RP_COMM_DIRECT, sharded adjoint (rp_backward_sharded): the cross-rank sum of the full-length partials W_p^T u_p
       WITHOUT a collective and in a FIXED order.  Every rank owns an inbox of 2 x n floats (two parities x one slot of
       `rows` floats per contributing rank), mapped by its peers.  Per adjoint step a rank pushes, for every owner q, the
       slice [q rows, (q + 1) rows) of its partial into slot `rank` of q's inbox of the step's parity, signals with the next
       sequence number (the same counter and flag array as the forward exchange), waits for every peer's signal and adds
       the `world` slots of its own inbox in RANK ORDER 0, 1, ..., world - 1 (its own contribution straight from its
       partial).  The sum is therefore the same from run to run and for every arrival order -- what an all-reduce does
       not promise -- and the oracle mirrors it as `world` row chunks added in ascending order.  Two parities suffice:
       a peer's signal of step s + 1 is enqueued behind its reduction of step s.  NULL: RP_COMM_DIRECT covers the forward
       exchange only. */
    float* const* peer_inbox;     /* [world] (host array): peer p's inbox, mapped here; own entry = inbox               */
    float* inbox;                 /* this rank's inbox [2 x n]                                                        */
} rp_comm;
#define RP_COMM_RCCL 0
#define RP_COMM_DIRECT 1
int rp_sizeof_comm(void);

/* Device memory that other PROCESSES can map (hipMalloc + hipIpcGetMemHandle; 64-byte handle), for the buffers of
   RP_COMM_DIRECT: rp_ipc_alloc zero-fills; rp_ipc_open maps a peer's allocation into this process; rp_ipc_close unmaps;
   rp_ipc_free releases an allocation of this process.  The caller exchanges the handles (torch.distributed). */
int rp_ipc_alloc(int64_t bytes, void** ptr, unsigned char handle[64]);
int rp_ipc_open(const unsigned char handle[64], void** ptr);
int rp_ipc_close(void* ptr);
int rp_ipc_free(void* ptr);

/* rp_forward on a row shard (net->rows = n / world, net->row0 = rank * rows) followed, every step, by the in-place
   all-gather of the ranks' slices of the next source vector.  Same arguments and results as rp_forward. */
int rp_forward_sharded(rp_net* net, rp_run* run, const rp_comm* comm, void* stream);

/* The whole sharded adjoint sweep (rp_bwd_prepare; t = T-1..0: rp_bwd_partial -> reduce-scatter -> rp_bwd_apply), for
   the leaky-integrator models and the spiking models QIF, QIF_SFA, LIF, IK. */
int rp_backward_sharded(const rp_net* net, const rp_bwd* bwd, const rp_comm* comm, void* stream);

/*
 * B independent trajectories through ONE coupling matrix (the "batched-input dense W @ state contraction" of the build's
 * brief; the reference has no batch axis -- it would run the B trajectories one after the other, streaming W B times).
 * Every Euler step streams W once and advances all B state vectors: lane l still owns columns 256 m + 4 l .. +3 of a row,
 * holds one set of four accumulators PER trajectory and feeds them from the B source vectors (L2-resident), so each
 * trajectory sees exactly the fmaf sequence of rp_forward: trajectory b of a batched call is bit-identical to an
 * unbatched call on the same state.  Rate models (LI_TANH, LI_SIGMOID), 16-byte path (n % 4 == 0), B in {2, 4, 8, 16}.
 * State lives in the caller's [B x ld] buffers; rp_net supplies W, the per-neuron parameters, dt and the shard geometry
 * (rp_net.y / src / cur are not used).
 */
typedef struct rp_batch {
    int32_t batch;            /* B                                                              */
    int32_t steps;
    float* v;                 /* [B x ldv] states v_b (in/out)                                   */
    int64_t ldv;
    float* src[2];            /* [B x ldsrc] ping-pong source vectors r_b = act(v_b), global length n (in/out) */
    int64_t ldsrc;
    int32_t cur;              /* [host, in/out] index of the current source buffer               */
    int32_t x_mode;           /* RP_X_NONE / RP_X_FULL / RP_X_BCAST / RP_X_PROJ (m, w_in below)   */
    const float* x;           /* inputs: row (t, b) at x + (t * B + b) * ldx                     */
    int64_t ldx;
    int32_t out_var;          /* RP_OUT_*                                                        */
    float* traj;              /* optional: pre-update output of (t, b) at traj + (t * B + b) * ldtraj */
    int64_t ldtraj;
    /* layout = RP_BATCH_MFMA: the contraction W @ [r_0 .. r_31] runs on the matrix cores (v_mfma_f32_32x32x2_f32, exact
       fp32) with up to 32 trajectories per pass over W.  State is NEURON-major then: v = [rows x 32], src[k] = [n x 32]
       (slot b of neuron i at i * 32 + b; ldv / ldsrc are ignored), batch <= 32 names the slots in use.  Summation order
       per (row i, trajectory b): the column tiles of 32 are taken in the sequence t0, t0 + 1, ... (mod tiles) with
       t0 = (3 * ((row0 + i) >> 5)) mod tiles; position q of that sequence goes to chain q mod 4 (the four waves of a
       workgroup); inside a tile the columns 8 g + {0, 4, 1, 5, 2, 6, 3, 7} for g = 0..3; the four fmaf chains are added
       as ((c0 + c1) + c2) + c3 (mirrored by the oracle).  Equal to the unbatched run up to fp32 rounding (different
       order), not bit for bit. */
    /* layout = RP_BATCH_MFMA16: the contraction runs on v_mfma_f32_16x16x4_f32 (exact fp32) with 1..32 trajectories per
       pass over W (one or two blocks of 16 slots: a batch <= 16 fills whole matrix tiles).  State is BATCH-major exactly
       as for RP_BATCH_VALU (v [B x ldv], src[k] [B x ldsrc]); row0 must be a multiple of 128.  W and the source tiles reach
       the LDS by LDS-DMA (global_load_lds_dwordx4) in a ring of stages shared by the four (eight) waves of a 64-row (128-row) workgroup.
       Summation order per (row i, trajectory b): ONE fmaf chain from 0 over the 16-column groups g0, g0 + 1, ..., G - 1,
       0, ..., g0 - 1 with G = ceil(n / 16), g0 = 16 * ((3 * ((row0 + i) >> 7)) mod ceil(n / 256)); inside a group the
       columns 16 g + 4 kq + c for c = 0..3 (outer) and kq = 0..3 (inner); columns >= n contribute fmaf(0, 0, acc)
       (mirrored by the oracle).  Equal to the unbatched run up to fp32 rounding, not bit for bit. */
    int32_t layout;           /* RP_BATCH_VALU (0) | RP_BATCH_MFMA (1) | RP_BATCH_MFMA16 (2)          */
    /* Sampling per trajectory, exactly as rp_run does it for one (rectipy/network.py:588-599): the step with absolute
       index step0 + t is buffered iff >= cutoff and recorded iff also a multiple of sampling_steps, as the running fp32
       sum of the buffered outputs divided by their count.  All optional (samples == NULL and rec == NULL: none of it).
       Every form keeps these buffers BATCH-major, whatever the layout of its state. */
    int64_t step0;            /* absolute index of the call's first step                           */
    int32_t cutoff;
    int32_t sampling_steps;   /* >= 1 when sampling                                                */
    int32_t max_samples;      /* sample rows per trajectory in samples / rec[]                     */
    float* acc;               /* [B x ldacc] running sums (in/out); required when sampling.  Read only while acc_count > 0 */
    int64_t ldacc;
    int32_t acc_count;        /* [host, in/out] buffered outputs in acc (the same for every trajectory) */
    int32_t sample_idx;       /* [host, in/out] next sample row                                    */
    float* samples;           /* optional: sample j of trajectory b at samples + (b * max_samples + j) * ldsamples */
    int64_t ldsamples;
    float* rec[2];            /* optional, same layout: post-update v (RP_REC_V) / r (RP_REC_SOURCE) at the sample steps */
    /* RP_X_PROJ (the input edge inside the step, rectipy/edges.py:48-49): x row (t, b) holds m inputs,
       I[i] = sum_j w_in[i * ldwin + j] * x[j] as a j-ascending fmaf chain from 0 (as rp_run) */
    int32_t m;
    const float* w_in;        /* [rows x ldwin]                                                    */
    int64_t ldwin;
} rp_batch;

#define RP_BATCH_VALU 0
#define RP_BATCH_MFMA 1
#define RP_BATCH_MFMA16 2
#define RP_BATCH_SLOTS 32

int rp_sizeof_batch(void);
int rp_forward_batched(const rp_net* net, rp_batch* batch, void* stream);

/*
 * Recursive-least-squares training of a readout edge (reference RLS.update, rectipy/edges.py:227-234, driven by
 * Network._rls, rectipy/network.py:1093-1121).  P is the n_in x n_in state-error correlation matrix.
 */
typedef struct rp_rls {
    int32_t n_in, n_out;
    int64_t ldp, ldw;
    float* P;        /* [n_in x ldp] in/out                                                   */
    float* W;        /* [n_out x ldw] readout weights in/out                                   */
    float beta_inv;  /* 1 / forgetting rate (edges.py:220)                                     */
    float* z;        /* scratch [n_in], 16-byte aligned                                        */
    float* zn;       /* scratch [n_in], 16-byte aligned                                        */
    float* scratch;  /* scratch [4 + n_out] ([2], [3]: the scalar kernel's x.z and its arrival counter)  */
    /* optional workspace [n_in x ceil(n_in / 1024)]: selects the TILED pass (16-byte path only).  One-shot workgroups
       sweep P in address order, each takes one 4 KiB tile (row i, 1024 columns) and leaves the tile's partial dot
       product here; z'_i = beta_inv * (sum of row i's partials, ascending).  NULL: one wave per row (small n_in). */
    float* partials;
    /* optional workspace [ceil(n_in / sym_rows) x ldc] (ldc >= n_in, ldc % 4 == 0; needs `partials`
       too): selects the SYMMETRIC pass.  P -= k z z^T keeps P symmetric (the reference's k * outer(z, z) does so exactly),
       so only the 1024 x 1024 blocks on and above the diagonal are streamed -- 4 n_in^2 instead of 8 n_in^2 bytes per
       update -- and every off-diagonal element serves both P_ij x_j (row partials, as in the tiled pass) and P_ij x_i
       (column partials per chunk of sym_rows rows: an i-ascending fmaf chain from 0; the chunks of a block row are added
       in ascending order to the block's column sum).  z'_i = beta_inv * (the column sums of the block rows above i's,
       ascending, then the row's tile partials from its diagonal block on, ascending).  On entry the blocks on and above
       the diagonal are authoritative (whatever the lower ones hold: callers must only hand over a P that is symmetric);
       on return the off-diagonal blocks have been mirrored, so P is full again -- unless defer_mirror is set.  Worth it
       from a few updates per call (the mirror costs about one dense pass). */
    float* colparts;
    int64_t ldc;
    int32_t sym_rows;  /* rows per chunk of the symmetric pass: 16, 32 or 64 (0: 64; blocked sweep: 64, 128 or 256); part of its summation order */
    /* symmetric pass only: 1 = return WITHOUT mirroring -- the blocks on and above the diagonal stay the authoritative
       half of P (the state the next symmetric sweep starts from anyway) and the caller restores the lower blocks with
       rp_rls_mirror before anything else reads P.  Saves one pass over half of P per call (0.78 ms at n_in = 32768). */
    int32_t defer_mirror;
    /* symmetric pass only: block > 1 (2, 4 or 8) selects the BLOCKED sweep -- P is streamed once per `block` updates.
       The inputs of a whole window are known in advance, and P_{t-1} = P_b - sum_{i<t} k_i z_i z_i^T inside a block that
       starts from P_b, so ONE pass applies the block before's rank-1 updates (element by element in update order: the bits
       of that many separate passes) and forms Y_j = beta^-1 P_b x_j for all inputs x_j of the block (each exactly as the
       symmetric pass forms z'); then, per update j of the block, without touching P:
           z_j = Y_j - sum_{i<j} g_ij z_i,  g_ij = beta^-1 (k_i c_ij),  c_ij = z_i . x_j,  k_j = 1 / (1 + z_j . x_j)
       IN DOUBLE PRECISION -- consecutive inputs of a slowly moving population are alike, so z_j is a small difference of
       large vectors: acc = Y_j[q]; acc = fma(-g_ij, z_i[q], acc) for i ascending; z_j[q] = (float)acc; the dot products
       in the order of x.z with double accumulators -- and the prediction, the loss and the weight update from z_j and k_j as
       always.  Mathematically the reference's recursion; measured against its float64 run the blocked sweep is as close as
       the pass per update (G31 / G34).
       r05: unless predictions BETWEEN updates are asked for (stride > 1 with pred) or n_out > 64, the per-update launches are replaced by
       TWO launches per block: every z_j is a combination of the block's Y vectors, z_j = sum_{l<=j} A_jl Y_l (A_jj = 1), so
       all dot products follow from the Gram matrix G_lj = Y_l . x_j (double, the order of x.z) and H_oj = W_o . x_j (float;
       W as the block finds it):  c_ij = sum_{l<=i} A_il G_lj (fma, l ascending from 0), g_ij as above, A_j = e_j - sum_{i<j}
       g_ij A_i (fma, i ascending), k_j = (float)(1 / (1 + c_jj)), yhat_j[o] = (float)(H_oj + sum_{i<j} (k_i err_i[o]) c_ij)
       (double, i ascending) -- all on block x block doubles --; then z_j[q] = (float)(Y_j[q] + sum_{l<j} A_jl Y_l[q]) (double, fma,
       l ascending) and W_o[q] += (k_j err_j[o]) z_j[q] for j ascending.  Every cancellation is carried in double.
       Workspaces grow by the factor `block`: partials [block x n_in x ceil(n_in / 1024)], colparts [block x chunks x ldc];
       blk: 2 * block * n_in floats (Y, the block's z vectors), 2 * block^2 doubles (c_ij or G_lj; A_jl), then
       block * (1 + 2 * n_out) floats (k_i; H; k_j err_j), 16-byte aligned.  0 / 1: one pass per update (above). */
    int32_t block;
    float* blk;
} rp_rls;

/* `steps` consecutive rows of X [steps x ldx] (edge inputs) and Y [steps x ldy] (targets); an update happens at
   rows 0, stride, 2*stride, ...  pred[t] = W x_t with the weights in force at step t (before the update at an update
   step); loss[t] = squared error of the most recent update.  The rank-1 update of P is fused with the mat-vec of
   the next update step (8 n_in^2 bytes of HBM traffic per update). */
int rp_sizeof_rls(void);
int rp_rls_sweep(const rp_rls* rls, int32_t steps, int32_t stride, const float* X, int64_t ldx, const float* Y,
                 int64_t ldy, float* pred, int64_t ldpred, float* loss, void* stream);
/* P[j][i] = P[i][j] for every 1024 x 1024 block below the diagonal (uses n_in, P, ldp only): what a symmetric sweep with
   defer_mirror = 1 left undone.  n_in % 4 == 0. */
int rp_rls_mirror(const rp_rls* rls, void* stream);

/* Streaming configuration of the dense step kernel, measured on the current device.  How W is walked (rows per wave,
   256-column chunks in flight, resident workgroups) never changes what is summed in which order, so every configuration
   gives the same bits; which one is fastest depends on the shape (rows, n, ldw) in ways only a measurement shows
   (DESIGN.md section 3).  rp_tune_forward times a few dozen configurations on a scratch tanh population over the
   caller's W (read-only; nothing of `net` but W, n, rows, row0, ldw is used), on `stream`, synchronising the host, and
   remembers the best per (device, rows, n, ldw) for every later rp_forward / rp_forward_sharded of that shape and any
   model.  Shapes that do not stream from HBM (rows < 8192 or W <= 320 MiB), the 4-byte path, and a capturing stream
   return RP_OK without measuring.  No reference counterpart.
   rp_forward_tuning reports the configuration rp_forward would use now (`measured` = 1 if it came from rp_tune_forward
   or rp_install_forward_tuning, with its time per step; 0: the static rule).
   rp_install_forward_tuning records a configuration measured EARLIER for this shape on this device model (the host
   side persists winners per (device name, rows, n, ldw), so that a second process pays nothing); (R, U) must be one of
   the compiled instances (1,4) (2,1) (2,2) (4,1) (4,2) (8,1).  A measurement is an optimisation, never a
   requirement: the host side measures only on request or when a run is long enough to repay it, and treats any error
   of rp_tune_forward as "keep the static rule". */
int rp_tune_forward(const rp_net* net, void* stream);
int rp_forward_tuning(const rp_net* net, int32_t* rows_per_wave, int32_t* chunks_in_flight, int32_t* workgroups,
                      float* us_per_step, int32_t* measured);
int rp_install_forward_tuning(const rp_net* net, int32_t rows_per_wave, int32_t chunks_in_flight, int32_t workgroups,
                               float us_per_step);
int rp_clear_forward_tuning(void);

/*
 * MEASUREMENT BUILD ONLY (-DRP_MEASUREMENT: rectipy_amd/_C/librectipy_hip_measure.so, `python -m rectipy_amd.build
 * --measurement`).  The product library exports none of these and contains none of the kernel instances they select:
 * process-wide knobs that override how a kernel walks its operands (the scripts under scripts/ sweep them), and the
 * ablation instances of two kernels ("probe": halves of the kernel switched off, results meaningless).  Unless a
 * comment says otherwise a knob changes the walk, never the summation order.  0 restores the built-in choice.
 */
#ifdef RP_MEASUREMENT
int rp_set_batched_tuning(int rows_per_wave, int unroll); /* measurement; 0 = built-in */
int rp_set_batched_mfma_tuning(int tile_columns, int stages, int waves); /* measurement: tile width (64, 128, 256), LDS ring depth and waves per workgroup (4, 8) of the RP_BATCH_MFMA16 step; 0 = built-in.  Tile width, ring depth and waves never change the bits; waves >= 100 select the ablation instances. */
int rp_set_batched_grid(int max_workgroups); /* measurement: cap on the resident workgroups of the vector-ALU batched step; 0 = built-in */
/* Measurement: override the streaming configuration of the tanh step kernel
   (rows per wave, unroll depth, non-temporal W loads, grid cap); rows_per_wave = 0 restores
   the built-in choice.  Results are bit-identical for every configuration. */
int rp_set_tuning(int rows_per_wave, int unroll, int nontemporal, int max_blocks);

/* Measurement: rows per wave and groups in flight per row of the sparse step kernel (tanh model); 0, 0 restores the
   built-in choice.  Results are bit-identical for every configuration. */
int rp_set_sparse_tuning(int rows_per_wave, int groups_in_flight);
int rp_set_blocked_tuning(int waves_per_workgroup, int groups_in_flight); /* blocked sparse step; 0 = built-in */

/* Measurement: rows in flight per wave (unroll) and 1-KiB column groups per lane of the adjoint's column-strip
   kernel; 0, 0 restores the built-in choice.  Results are bit-identical for every configuration. */
int rp_set_bwd_tuning(int unroll, int strip_groups);
int rp_set_bwd_grid(int max_workgroups); /* measurement: cap on the strip kernel's resident workgroups; 0 = built-in */
/* Measurement: number of row chunks of the adjoint (0 = built-in RC(n)).  Unlike the two above this CHANGES the
   summation order, i.e. results then differ from the oracle in the last bits; never set in production. */
int rp_set_bwd_chunks(int max_chunks);

/* Measurement: rows per wave, column chunks in flight and store policy (-1 = default) of the RLS pass. */
int rp_set_rls_tuning(int rows_per_wave, int unroll, int nontemporal_store);
int rp_set_gemm_dma(int on); /* measurement: 0 = rp_outer_accum always uses the register-staged kernel (1: LDS-DMA ring on full tiles); identical bits */
int rp_set_rls_sym(int rows_in_flight); /* measurement: bits 0-7 rows in flight per wave of the symmetric pass (1, 2, 4, 8; 0 = built-in), bits 8-11 resident workgroups per CU (an LDS allocation; 0 = built-in, 15 = no cap); never the bits */
int rp_set_rls_blk(int rows_in_flight, int workgroups_per_cu);  /* blocked sweep's pass: rows in flight (0: built-in), residency cap (-1: built-in) */
int rp_set_rls_tile(int kcols_per_tile, int workgroups_per_cu); /* measurement: tile width (x 1024 columns; changes the summation order) and residency cap of the tiled RLS pass; 0, 0 = built-in */

#endif /* RP_MEASUREMENT */

/* Bandwidth probe used by bench.py: sums `bytes` of HBM with the same load shape as the step kernel. */
int rp_stream_probe(const float* buf, int64_t n_floats, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RECTIPY_HIP_H */
