// rnn_batched.hip -- B independent trajectories through one coupling matrix (rp_forward_batched).
//
// The unbatched step is a mat-vec: 4 N^2 bytes of W for 2 N^2 flop, HBM-bound at 0.5 flop/byte.  B trajectories that
// share W (trials of one network, an ensemble of initial conditions, input sweeps) turn the step into W @ [r_0 .. r_B-1]:
// W is streamed ONCE per step and reused B times from registers, so the cost per trajectory-step drops by up to B until
// the vector ALUs bind (0.5 B flop/byte).  The reference (rectipy/nodes.py:166-170, rectipy/network.py:588-599) has no
// batch axis and would run the trajectories one after the other.
//
// Summation order: exactly that of rp_dot_rows_v4 (rp_dot.h) per trajectory -- lane l keeps four accumulators per
// (row, trajectory) for the columns 256 m + 4 l .. +3, chunks from the staggered start m0 with wrap-around, partial
// chunk last, lane sum (a0+a1)+(a2+a3), fixed butterfly -- so trajectory b of a batched call reproduces an unbatched
// rp_forward on the same state bit for bit (tests/test_batched.py).

#include <atomic>
#include <type_traits>

#include "rp_common.h"
#include "rp_dot.h"

namespace {

constexpr int kThreads = 256;

struct BatchParams {
    const float* W;
    int64_t ldw;
    int n, rows, row0;
    const float* p[RP_MAX_PARAMS];
    float dt;
    float* v;
    int64_t ldv;
    const float* src_cur;
    float* src_next;
    int64_t ldsrc;
    int x_mode;
    const float* x_t;  // row (t, 0)
    int64_t ldx;
    int out_var;
    float* traj_t;     // row (t, 0)
    int64_t ldtraj;
    // RP_X_PROJ: the input edge inside the step (x row (t, b) holds m inputs)
    int m;
    const float* w_in;
    int64_t ldwin;
    // sampling of this step (rp_batch: per trajectory what rp_run does for one; all buffers batch-major)
    int flags;          // FLAG_BUFFER | FLAG_SAMPLE | FLAG_FIRST
    float count;
    float* acc;
    int64_t ldacc;
    float* sample_row;  // row (sample, trajectory 0); trajectory b at + b * sample_stride
    float* rec_row[2];
    int64_t sample_stride;
};

constexpr int FLAG_BUFFER = 1;  // step >= cutoff: output joins the sampling buffer
constexpr int FLAG_SAMPLE = 2;  // record the mean of the buffer at this step
constexpr int FLAG_FIRST = 4;   // first output of a buffer: the accumulator's old contents are not read

__device__ __forceinline__ float batched_input(const BatchParams& P, int i, int b) {
    if (P.x_mode == RP_X_FULL) return P.x_t[(int64_t)b * P.ldx + i];
    if (P.x_mode == RP_X_BCAST) return P.x_t[(int64_t)b * P.ldx];
    if (P.x_mode == RP_X_PROJ) {  // weights @ x (edges.py:48-49), sequential in j as in rp_forward
        float s = 0.f;
        const float* wr = P.w_in + (int64_t)i * P.ldwin;
        const float* xb = P.x_t + (int64_t)b * P.ldx;
        for (int j = 0; j < P.m; ++j) s = fmaf(wr[j], xb[j], s);
        return s;
    }
    return 0.f;
}

// trajectory output, running-buffer mean (network.py:590-597) and post-update recordings (network.py:147-148) of
// neuron i, trajectory b -- the unbatched kernel's sample_output per trajectory
__device__ __forceinline__ void batched_outputs(const BatchParams& P, int i, int b, float out, float v_new, float r_new) {
    if (P.traj_t) P.traj_t[(int64_t)b * P.ldtraj + i] = out;
    if (P.flags & FLAG_BUFFER) {
        float* ap = P.acc + (int64_t)b * P.ldacc + i;
        float a = ((P.flags & FLAG_FIRST) ? 0.0f : *ap) + out;
        if (P.flags & FLAG_SAMPLE) {
            if (P.sample_row) P.sample_row[(int64_t)b * P.sample_stride + i] = a / P.count;
            a = 0.f;
        }
        *ap = a;
    }
    if (P.flags & FLAG_SAMPLE) {
        if (P.rec_row[0]) P.rec_row[0][(int64_t)b * P.sample_stride + i] = v_new;
        if (P.rec_row[1]) P.rec_row[1][(int64_t)b * P.sample_stride + i] = r_new;
    }
}

template <int MODEL>
__device__ __forceinline__ float act(const BatchParams& P, int i, float v) {
    if constexpr (MODEL == RP_MODEL_LI_TANH) {
        return rp_tanhf(v);
    } else {  // leaky_integrator.yaml:22
        const float e = rp_expf(P.p[RP_P_SLOPE][i] * (P.p[RP_P_V0][i] - v));
        return P.p[RP_P_RMAX][i] / (1.0f + e);
    }
}

// RateNet.forward (nodes.py:166-170) with the field of leaky_integrator.yaml:10 for neuron i of trajectory b
template <int MODEL>
__device__ __forceinline__ void epilogue(const BatchParams& P, int i, int b, float wsum) {
    const int gi = P.row0 + i;
    const float I = batched_input(P, i, b);
    float* vb = P.v + (int64_t)b * P.ldv;
    const float v = vb[i];
    const float dv = ((-v / P.p[RP_P_TAU][i] + P.p[RP_P_K][i] * wsum) + I) + P.p[RP_P_ETA][i];
    const float v_new = v + P.dt * dv;
    const float r_new = act<MODEL>(P, i, v_new);
    vb[i] = v_new;
    P.src_next[(int64_t)b * P.ldsrc + gi] = r_new;
    if (P.traj_t || P.flags)
        batched_outputs(P, i, b, P.out_var == RP_OUT_STATE0 ? v : P.src_cur[(int64_t)b * P.ldsrc + gi], v_new, r_new);
}

// one 256-column chunk: R loads of W, B loads of the sources, R * B * 4 fmaf (fully unrolled: the accumulators must
// stay in registers)
template <int R, int B>
__device__ __forceinline__ void batched_chunk(const float* const (&wrow)[R], const float* sp, int64_t ldsrc, int m,
                                              rp_f4 (&a)[R][B]) {
    rp_f4 w[R];
#pragma clang loop unroll(full)
    for (int j = 0; j < R; ++j) w[j] = rp_ld4_stream(wrow[j] + (int64_t)m * 256);
#pragma clang loop unroll(full)
    for (int b = 0; b < B; ++b) {
        const rp_f4 s = rp_ld4(sp + (int64_t)b * ldsrc + (int64_t)m * 256);
#pragma clang loop unroll(full)
        for (int j = 0; j < R; ++j) {
            a[j][b].x = fmaf(w[j].x, s.x, a[j][b].x);
            a[j][b].y = fmaf(w[j].y, s.y, a[j][b].y);
            a[j][b].z = fmaf(w[j].z, s.z, a[j][b].z);
            a[j][b].w = fmaf(w[j].w, s.w, a[j][b].w);
        }
    }
}

// One wave owns R consecutive rows and all B trajectories: R * B accumulator sets (R * B <= 32).
template <int MODEL, int R, int B, int U>
__global__ __launch_bounds__(kThreads) void step_batched_kernel(const BatchParams P) {
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int groups = (P.rows + R - 1) / R;
    const int M = P.n >> 8;
    for (int g = blockIdx.x * (kThreads / RP_WAVE) + wave; g < groups; g += gridDim.x * (kThreads / RP_WAVE)) {
        const int row_base = g * R;
        const float* wrow[R];
#pragma clang loop unroll(full)
        for (int j = 0; j < R; ++j) {
            int rr = row_base + j;
            rr = rr < P.rows ? rr : P.rows - 1;  // clamp: tail rows re-read a valid row, result discarded
            wrow[j] = P.W + (int64_t)rr * P.ldw + 4 * lane;
        }
        const float* sp = P.src_cur + 4 * lane;
        rp_f4 a[R][B];
#pragma clang loop unroll(full)
        for (int j = 0; j < R; ++j)
#pragma clang loop unroll(full)
            for (int b = 0; b < B; ++b) a[j][b] = rp_f4{0.f, 0.f, 0.f, 0.f};
        const int m0 = M > 0 ? (int)(((unsigned)((P.row0 + row_base) >> 3) * kStagger) % (unsigned)M) : 0;
        // chunks m0 .. M-1, then 0 .. m0-1 (two ascending sweeps, U chunks per iteration), the partial chunk last
        for (int pass = 0; pass < 2; ++pass) {
            int m = pass == 0 ? m0 : 0;
            const int mB = pass == 0 ? M : m0;
            for (; m + U <= mB; m += U) {
#pragma clang loop unroll(full)
                for (int u = 0; u < U; ++u) batched_chunk<R, B>(wrow, sp, P.ldsrc, m + u, a);
            }
            for (; m < mB; ++m) batched_chunk<R, B>(wrow, sp, P.ldsrc, m, a);
        }
        if ((M << 8) + 4 * lane < P.n) batched_chunk<R, B>(wrow, sp, P.ldsrc, M, a);  // partial last chunk (n % 4 == 0)
        float mine = 0.f;
#pragma clang loop unroll(full)
        for (int j = 0; j < R; ++j)
#pragma clang loop unroll(full)
            for (int b = 0; b < B; ++b) {
                const float sum = rp_wave_allreduce(rp_lane_sum4(a[j][b]));
                if (lane == j * B + b) mine = sum;
            }
        if (lane < R * B) {
            const int j = lane / B, b = lane - j * B;
            if (row_base + j < P.rows) epilogue<MODEL>(P, row_base + j, b, mine);
        }
    }
}

#ifdef RP_MEASUREMENT
std::atomic<int> g_bt_R{0}, g_bt_U{0}, g_bt_blocks{0};  // measurement build only (rp_set_batched_tuning / _grid)
#endif

template <int MODEL, int R, int B, int U>
void launch(const BatchParams& P, hipStream_t st, int default_blocks = 2048) {
    const int groups = (P.rows + R - 1) / R;
    int blocks = (groups + 3) / 4;
#ifdef RP_MEASUREMENT
    const int cap = g_bt_blocks > 0 ? (int)g_bt_blocks : default_blocks;
#else
    const int cap = default_blocks;
#endif
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((step_batched_kernel<MODEL, R, B, U>), dim3(blocks), dim3(kThreads), 0, st, P);
}

// ---- matrix-core form: up to 32 trajectories per pass over W --------------------------------------------------------
// C[32 rows x 32 slots] per workgroup = W[32 x n] @ R[n x 32] with v_mfma_f32_32x32x2_f32 (exact fp32: fmaf chains).
// The reduction over the columns is split over the workgroup's four waves (wave w takes the column tiles q = w, w + 4, ...
// of the staggered tile sequence), so a 32768-neuron network gives 1024 workgroups x 4 waves of independent work and no
// barrier inside the loop: every wave owns its W tile and its source tile in LDS.  W tiles (32 rows x 32 columns) are
// loaded coalesced (8 rows x 128 B per wave instruction, non-temporal) and parked in registers while the previous tile
// is multiplied; MFMA operands are read back as one 16-byte LDS read (A: row l & 31, columns 8 g + 4 (l >> 5) .. +3) and
// four 4-byte reads (B) per four MFMAs.  The four partial tiles are added in wave order through LDS.  State is
// neuron-major ([n x 32]), so the C fragment (slot = lane & 31) reads and writes it coalesced.
// Measured split of the 0.89 ms step at N = 32768 (probe flags of this kernel, r02): loads + LDS stores alone 0.74 ms
// (W at 5.8 TB/s: 128-byte pieces of 8 rows per instruction are a poorer DRAM pattern than the 1 KiB pieces of the
// unbatched kernel), LDS reads + MFMA alone 0.58 ms (118 TFLOP/s); the two overlap only partly.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMT = 32;        // columns per tile
constexpr int kWs = kMT + 4;   // LDS row stride of a W tile (floats): rows 16-byte aligned, quarter-wave reads conflict-free
constexpr int kLpr = kMT / 4;  // lanes per tile row of a W load (16 bytes each)
constexpr int kRpi = 64 / kLpr;  // tile rows per W load instruction
constexpr int kWld = 32 / kRpi;  // W load instructions per tile
constexpr int kRld = kMT * 32 / 4 / 64;  // source load instructions per tile (float4 per lane)
constexpr int kRs = 40;        // LDS row stride of a source tile: the two half-waves of a B read hit disjoint banks
constexpr int kSlots = RP_BATCH_SLOTS;
constexpr int kRT = 1;         // 32-row MFMA tiles per wave (2: every source operand read feeds two MFMAs; +3 % at N = 32768,
                               // but half the workgroups: slower below N = 16384)
constexpr int kRows = 32 * kRT;  // rows per workgroup
constexpr int kWaveLds = kRows * kWs + kMT * kRs;  // floats per wave: its W tile and its source tile

template <int MODEL>
__global__ __launch_bounds__(kThreads) void step_batched_mfma_kernel(const BatchParams P, int batch) {
    __shared__ __attribute__((aligned(16))) float lds[4 * kWaveLds];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int row_blk = blockIdx.x * kRows;         // first local row of this workgroup
    const int tiles = (P.n + kMT - 1) / kMT;
    const int t0 = (int)((3u * (unsigned)((P.row0 + row_blk) / kRows)) % (unsigned)tiles);
    float* myW = lds + wave * kWaveLds;
    float* myR = myW + kRows * kWs;

    rp_f4 wreg[kRT * kWld], rreg[kRld];
    auto load_tile = [&](int tile) {  // global -> registers: kRows x kMT of W (kRpi rows per instruction), kMT x 32 of R
        const int c = tile * kMT + 4 * (lane % kLpr);
#pragma clang loop unroll(full)
        for (int q = 0; q < kRT * kWld; ++q) {
            int row = row_blk + kRpi * q + (lane / kLpr);
            row = row < P.rows ? row : P.rows - 1;  // clamp: rows past the end re-read a valid row, results discarded
            wreg[q] = c < P.n ? rp_ld4_stream(P.W + (int64_t)row * P.ldw + c) : rp_f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma clang loop unroll(full)
        for (int q = 0; q < kRld; ++q) {  // kMT * 8 float4: lane l takes float4 l, l + 64, ...
            const int f = lane + 64 * q;
            const int k = tile * kMT + (f >> 3);
            rreg[q] = k < P.n ? rp_ld4(P.src_cur + (int64_t)k * kSlots + 4 * (f & 7)) : rp_f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&]() {
#pragma clang loop unroll(full)
        for (int q = 0; q < kRT * kWld; ++q)
            *reinterpret_cast<rp_f4*>(myW + (kRpi * q + (lane / kLpr)) * kWs + 4 * (lane % kLpr)) = wreg[q];
#pragma clang loop unroll(full)
        for (int q = 0; q < kRld; ++q) {
            const int f = lane + 64 * q;
            *reinterpret_cast<rp_f4*>(myR + (f >> 3) * kRs + 4 * (f & 7)) = rreg[q];
        }
    };

    f32x16 acc[kRT];
#pragma clang loop unroll(full)
    for (int j = 0; j < kRT; ++j)
#pragma clang loop unroll(full)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    if (wave < tiles) load_tile((t0 + wave) % tiles);
    for (int q = wave; q < tiles; q += 4) {
        store_tile();  // (the wave's own LDS tiles: program order is all the synchronisation needed)
        if (q + 4 < tiles) load_tile((t0 + q + 4) % tiles);  // next tile in flight behind the MFMAs
        const float* ap = myW + lr * kWs + 4 * lh;
        const float* bp = myR + (4 * lh) * kRs + lr;
#pragma clang loop unroll(full)
        for (int g = 0; g < kMT / 8; ++g) {
            rp_f4 a[kRT];
#pragma clang loop unroll(full)
            for (int j = 0; j < kRT; ++j) a[j] = *reinterpret_cast<const rp_f4*>(ap + j * 32 * kWs + 8 * g);
            const float b0 = bp[(8 * g + 0) * kRs], b1 = bp[(8 * g + 1) * kRs];
            const float b2 = bp[(8 * g + 2) * kRs], b3 = bp[(8 * g + 3) * kRs];
#pragma clang loop unroll(full)
            for (int j = 0; j < kRT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b0, acc[j], 0, 0, 0);
#pragma clang loop unroll(full)
            for (int j = 0; j < kRT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b1, acc[j], 0, 0, 0);
#pragma clang loop unroll(full)
            for (int j = 0; j < kRT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b2, acc[j], 0, 0, 0);
#pragma clang loop unroll(full)
            for (int j = 0; j < kRT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, b3, acc[j], 0, 0, 0);
        }
    }
    // partial tiles of waves 1..3 -> LDS (their own regions are free now); wave 0 adds them in wave order
    __syncthreads();
    if (wave > 0) {
#pragma clang loop unroll(full)
        for (int j = 0; j < kRT; ++j)
#pragma clang loop unroll(full)
            for (int r = 0; r < 16; ++r) myW[(j * 16 + r) * 64 + lane] = acc[j][r];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma clang loop unroll(full)
    for (int j = 0; j < kRT; ++j)
#pragma clang loop unroll(full)
        for (int r = 0; r < 16; ++r) {
            const int o = (j * 16 + r) * 64 + lane;
            acc[j][r] = ((acc[j][r] + lds[1 * kWaveLds + o]) + lds[2 * kWaveLds + o]) + lds[3 * kWaveLds + o];
        }
    // C/D layout of the 32x32 MFMA: slot = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (lr < batch) {
#pragma clang loop unroll(full)
        for (int j = 0; j < kRT; ++j)
#pragma clang loop unroll(full)
            for (int r = 0; r < 16; ++r) {
                const int i = row_blk + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (i < P.rows) {
                    const int gi = P.row0 + i, b = lr;
                    const float I = batched_input(P, i, b);
                    float* vp = P.v + (int64_t)i * kSlots + b;
                    const float v = *vp;
                    const float dv = ((-v / P.p[RP_P_TAU][i] + P.p[RP_P_K][i] * acc[j][r]) + I) + P.p[RP_P_ETA][i];
                    const float v_new = v + P.dt * dv;
                    const float r_new = act<MODEL>(P, i, v_new);
                    *vp = v_new;
                    P.src_next[(int64_t)gi * kSlots + b] = r_new;
                    if (P.traj_t || P.flags)
                        batched_outputs(P, i, b, P.out_var == RP_OUT_STATE0 ? v : P.src_cur[(int64_t)gi * kSlots + b], v_new, r_new);
                }
            }
    }
}

// ---- matrix-core form on v_mfma_f32_16x16x4_f32 (RP_BATCH_MFMA16): 1..32 trajectories, batch-major state ------------
// C[16 rows x 16 slots] per MFMA, K = 4 per instruction at 32 cycles: the same FLOP rate as the 32x32x2 form, but a batch
// of <= 16 fills whole tiles (no padded half) and the state keeps the vector-ALU form's [B x n] layout.  A workgroup of
// WAVES (four or eight) waves owns 16 rows per wave and walks the column tiles of TC columns TOGETHER: the 16 NB x TC
// source tile is loaded once per workgroup (each wave brings its share) and shared, each wave brings its own 16 x TC block
// of W.  All
// loads are LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, W non-temporal, no VGPR staging, no ds_write)
// into a ring of S stages, with a counted s_waitcnt vmcnt + ONE raw s_barrier per tile, software-pipelined:
//     [fragments of tile i in registers] wait own DMAs of tile i+1 -> barrier (everyone's landed; everyone holds tile i's
//     fragments, so its stage is free) -> between the MFMAs of tile i: LDS reads of tile i+1's fragments into the other
//     register set and the DMA pieces of tile i+S into the freed stage.
// LDS image: row-major rows of TC floats, the 16-byte quads of a row XOR-swizzled with (row & 15) -- applied on the DMA's
// per-lane SOURCE address (the LDS side of a DMA is lane-linear) and on the read address: lane (r = l & 15, kq = l >> 4)
// reads quad (4 j + kq) ^ r of row r for A and of slot-row r for B with ds_read_b128, conflict-free, and feeds component c
// to MFMA c of the group (k index kq <-> column 16 j + 4 kq + c for both operands).
// Summation order per (row, slot): ONE fmaf chain from 0 over the 16-column groups g0, g0 + 1, ..., G - 1, 0, ..., g0 - 1
// (G = ceil(n / 16), g0 = 16 * ((3 * (global row >> 7)) mod ceil(n / 256))); inside a group the columns
// 16 g + 4 kq + c for c = 0..3 (outer), kq = 0..3 (inner: the k order of one MFMA); columns >= n contribute
// fmaf(0, 0, acc).  Independent of TC and S.  Mirrored by oracle_forward_batched_mfma16.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* rp_gptr;
typedef __attribute__((address_space(3))) void* rp_lptr;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PROBE (measurement builds only, results meaningless): 1 = no MFMAs, 2 = no DMA in the steady state
template <int MODEL, int NB, int TC, int S, int WAVES, int PROBE = 0>
__global__ __launch_bounds__(64 * WAVES) void step_batched_mfma16_kernel(const BatchParams P, int batch) {
    extern __shared__ __attribute__((aligned(16))) float lds16[];
    constexpr int ROWS = 16 * WAVES;         // rows per workgroup (16 per wave)
    constexpr int QPR = TC / 4;              // 16-byte quads per tile row
    constexpr int W_INSTR = TC / 16;         // DMA instructions (1 KiB) for a wave's 16 x TC block of W
    constexpr int R_INSTR = NB * TC / (16 * WAVES);  // ... for this wave's share of the 16 NB x TC source tile
    constexpr int DPT = W_INSTR + R_INSTR;   // DMA instructions per tile and wave
    constexpr int STAGE = ROWS * TC + 16 * NB * TC;  // floats per stage: the workgroup's rows of W, 16 NB slot-rows of the source
    static_assert((NB * TC) % (16 * WAVES) == 0, "every wave brings the same number of source pieces");
    static_assert(DPT * (S - 1) < 64 && S >= 2 && S <= 4, "ring depth");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row_blk = blockIdx.x * ROWS;
    const int tiles = (P.n + TC - 1) / TC;
    const int supers = (P.n + 255) / 256;
    const int t0 = (int)((3u * (unsigned)((P.row0 + row_blk) >> 7)) % (unsigned)supers) * (256 / TC);
    const bool ragged = (P.n % TC) != 0;     // the last tile has columns >= n: filled through registers, zeros past n

    // this lane's part of every DMA instruction: LDS quad Q = 64 i + lane of the block -> (row, swizzled quad)
    const float* wsrc[W_INSTR];
    int wq[W_INSTR];
#pragma clang loop unroll(full)
    for (int i = 0; i < W_INSTR; ++i) {
        const int Q = 64 * i + lane, row = Q / QPR, qp = Q % QPR;
        int grow = row_blk + 16 * wave + row;
        grow = grow < P.rows ? grow : P.rows - 1;  // clamp: rows past the end re-read a valid row, results discarded
        wq[i] = 4 * (qp ^ (row & 15));
        wsrc[i] = P.W + (int64_t)grow * P.ldw + wq[i];
    }
    const float* rsrc[R_INSTR];
    int rq[R_INSTR];
#pragma clang loop unroll(full)
    for (int i = 0; i < R_INSTR; ++i) {
        const int Q = 64 * (wave * R_INSTR + i) + lane, srow = Q / QPR, qp = Q % QPR;
        const int slot = srow < batch ? srow : batch - 1;  // unused slots re-read the last trajectory (never stored)
        rq[i] = 4 * (qp ^ (srow & 15));
        rsrc[i] = P.src_cur + (int64_t)slot * P.ldsrc + rq[i];
    }

    // One piece (k = 0 .. DPT-1: the wave's W pieces first, then its source pieces) of a tile's DMA; a ragged tile (columns
    // >= n) goes through registers instead, zeros past n, all pieces at once with k == 0.
    auto issue_piece = [&](int tile, int slot, int k) {
        float* stage = lds16 + slot * STAGE;
        float* myW = stage + wave * (16 * TC);
        float* myR = stage + ROWS * TC + wave * (R_INSTR * 256);
        const int c0 = tile * TC;
        if (ragged && tile == tiles - 1) {
            if (k != 0) return;
#pragma clang loop unroll(full)
            for (int i = 0; i < W_INSTR; ++i) {
                const rp_f4 w = c0 + wq[i] < P.n ? rp_ld4_stream(wsrc[i] + c0) : rp_f4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<rp_f4*>(myW + 256 * i + 4 * lane) = w;
            }
#pragma clang loop unroll(full)
            for (int i = 0; i < R_INSTR; ++i) {
                const rp_f4 rr = c0 + rq[i] < P.n ? rp_ld4(rsrc[i] + c0) : rp_f4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<rp_f4*>(myR + 256 * i + 4 * lane) = rr;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // a raw barrier does not wait for LDS stores
            return;
        }
#pragma clang loop unroll(full)
        for (int i = 0; i < W_INSTR; ++i)
            if (i == k) __builtin_amdgcn_global_load_lds((rp_gptr)(wsrc[i] + c0), (rp_lptr)(myW + 256 * i), 16, 0, 2 /* nt */);
#pragma clang loop unroll(full)
        for (int i = 0; i < R_INSTR; ++i)
            if (W_INSTR + i == k) __builtin_amdgcn_global_load_lds((rp_gptr)(rsrc[i] + c0), (rp_lptr)(myR + 256 * i), 16, 0, 0);
    };

    constexpr int G = TC / 16;  // 16-column groups per tile
    struct Frags {
        f32x4 a[G], b[NB][G];
    };
    f32x4 acc[NB];
#pragma clang loop unroll(full)
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, kq = lane >> 4;
    auto read_group = [&](Frags& f, int slot, int j) {
        const float* stage = lds16 + slot * STAGE;
        const int off = r * TC + 4 * ((4 * j + kq) ^ r);
        f.a[j] = *reinterpret_cast<const f32x4*>(stage + wave * (16 * TC) + off);
#pragma clang loop unroll(full)
        for (int nb = 0; nb < NB; ++nb) f.b[nb][j] = *reinterpret_cast<const f32x4*>(stage + ROWS * TC + nb * 16 * TC + off);
    };

    // Software pipeline over the tile sequence (position i, tile (t0 + i) mod tiles, stage i mod S).  At the top of
    // iteration i the fragments of tile i are in registers; the iteration waits for tile i+1's DMAs, passes the barrier,
    // and then issues, BETWEEN the MFMAs of tile i, the LDS reads of tile i+1's fragments (other register set) and the DMA
    // pieces of tile i+S (into the stage tile i occupied): the matrix pipe never waits for an LDS read or a DMA issue.
    int issued = 0;  // tiles issued so far (positions 0 .. issued-1)
    int tile_issue = t0, slot_issue = 0;
    for (; issued < S && issued < tiles; ++issued) {
#pragma clang loop unroll(full)
        for (int k = 0; k < DPT; ++k) issue_piece(tile_issue, slot_issue, k);
        tile_issue = tile_issue + 1 == tiles ? 0 : tile_issue + 1;
        slot_issue = slot_issue + 1 == S ? 0 : slot_issue + 1;
    }
    {   // tile 0 has landed once at most the tiles issued after it are outstanding
        const int later = issued - 1;
        if (later >= 3) wait_vmcnt<(S >= 4 ? 3 * DPT : 0)>();
        else if (later == 2) wait_vmcnt<(S >= 3 ? 2 * DPT : 0)>();
        else if (later == 1) wait_vmcnt<DPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    Frags f0, f1;
#pragma clang loop unroll(full)
    for (int j = 0; j < G; ++j) read_group(f0, 0, j);

    int tile = t0, slot = 0;  // tile and stage of position i
    // FAST: the steady state -- a next tile exists, a tile is issued, neither this tile nor the issued one is the ragged
    // one: no conditionals between the barrier and the last MFMA.
    auto iteration = [&](auto fast_tag, int i, Frags& cur, Frags& nxt) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const bool more = FAST || i + 1 < tiles;
        const int slot1 = slot + 1 == S ? 0 : slot + 1;
        // this tile's fragments must be in registers before the barrier (its stage is overwritten after it): naming them
        // as inputs makes the compiler wait for exactly their LDS reads here, and know below that they have arrived
#pragma clang loop unroll(full)
        for (int j = 0; j < G; ++j) {
            asm volatile("" ::"v"(cur.a[j]));
#pragma clang loop unroll(full)
            for (int nb = 0; nb < NB; ++nb) asm volatile("" ::"v"(cur.b[nb][j]));
        }
        if (more) {
            const int later = issued - 2 - i;  // tiles issued after tile i+1: 0 .. S-2
            if (FAST) wait_vmcnt<(S - 2) * (PROBE == 3 ? W_INSTR : DPT)>();
            else if (later >= 2) wait_vmcnt<(S >= 4 ? 2 * DPT : 0)>();
            else if (later == 1) wait_vmcnt<(S >= 3 ? DPT : 0)>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool issue_now = FAST || issued < tiles;
        const int groups_here = FAST ? G : (P.n - tile * TC + 15) >> 4;  // 16-column groups of this tile that start below n
#pragma clang loop unroll(full)
        for (int j = 0; j < G; ++j) {
            if (more) read_group(nxt, slot1, j);
            if (issue_now) {
#pragma clang loop unroll(full)
                for (int k = j; k < DPT; k += G) {
                    if constexpr (FAST && PROBE == 2) {
                    } else if constexpr (FAST) {
                        float* stage = lds16 + slot * STAGE;
                        const int c0 = tile_issue * TC;
                        if (k < W_INSTR)
                            __builtin_amdgcn_global_load_lds((rp_gptr)(wsrc[k < W_INSTR ? k : 0] + c0),
                                                             (rp_lptr)(stage + wave * (16 * TC) + 256 * k), 16, 0, 2 /* nt */);
                        else if (PROBE != 3)
                            __builtin_amdgcn_global_load_lds(
                                (rp_gptr)(rsrc[k >= W_INSTR ? k - W_INSTR : 0] + c0),
                                (rp_lptr)(stage + ROWS * TC + wave * (R_INSTR * 256) + 256 * (k - W_INSTR)), 16, 0, 0);
                    } else {
                        issue_piece(tile_issue, slot, k);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PROBE == 1 || PROBE == 3) {
                asm volatile("" ::"v"(cur.a[j]));
#pragma clang loop unroll(full)
                for (int nb = 0; nb < NB; ++nb) asm volatile("" ::"v"(cur.b[nb][j]));
            } else if (j < groups_here) {
#pragma clang loop unroll(full)
                for (int c = 0; c < 4; ++c)
#pragma clang loop unroll(full)
                    for (int nb = 0; nb < NB; ++nb)
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[j][c], cur.b[nb][j][c], acc[nb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (issue_now) {
            tile_issue = tile_issue + 1 == tiles ? 0 : tile_issue + 1;
            ++issued;
        }
        tile = tile + 1 == tiles ? 0 : tile + 1;
        slot = slot1;
    };
    auto step = [&](int i, Frags& cur, Frags& nxt) {
        const bool fast = issued < tiles && !(ragged && (tile == tiles - 1 || tile_issue == tiles - 1));
        if (fast) iteration(std::true_type{}, i, cur, nxt);
        else iteration(std::false_type{}, i, cur, nxt);
    };
    for (int i = 0; i < tiles; i += 2) {
        step(i, f0, f1);
        if (i + 1 < tiles) step(i + 1, f1, f0);
    }

    // C/D layout of the 16x16 MFMA: slot = lane & 15, row = 4 * (lane >> 4) + reg
    const int i0 = row_blk + 16 * wave + 4 * kq;
#pragma clang loop unroll(full)
    for (int nb = 0; nb < NB; ++nb) {
        const int b = 16 * nb + r;
        if (b >= batch) continue;
#pragma clang loop unroll(full)
        for (int e = 0; e < 4; ++e)
            if (i0 + e < P.rows) epilogue<MODEL>(P, i0 + e, b, acc[nb][e]);
    }
}

// ---- the same contraction with the waves of a workgroup split into roles (17..32 trajectories) -------------------------
// With two slot blocks the matrix pipe is as long a pole as the HBM stream (N = 32768: 0.58 ms of MFMAs, 0.62 ms of DMA),
// and a wave that issues its own DMA pieces (60-180 cycles each, in order) stops feeding the pipe meanwhile.  Here a
// 512-thread workgroup owns 128 rows: waves 0..3 only multiply (32 rows x 32 slots each: 2 x 2 accumulators, every A and
// every B fragment feeds two MFMAs, 64 MFMAs per 64-column tile = all of the SIMD's matrix cycles), waves 4..7 -- their
// SIMD partners -- only load (each a quarter of the 128 x TC block of W and of the 32 x TC source tile per tile).  Same
// ring, same ONE barrier per tile, same LDS image, and the SAME summation order as step_batched_mfma16_kernel.
template <int MODEL, int TC, int S, int PROBE = 0>
__global__ __launch_bounds__(512) void step_batched_mfma16_split_kernel(const BatchParams P, int batch) {
    extern __shared__ __attribute__((aligned(16))) float lds16[];
    constexpr int ROWS = 128, NB = 2, RB = 2;
    constexpr int QPR = TC / 4;
    constexpr int G = TC / 16;
    constexpr int W_PIECES = ROWS * TC / 256 / 4;     // 1 KiB DMA pieces of W per loader wave and tile
    constexpr int R_PIECES = 16 * NB * TC / 256 / 4;  // ... of the source tile
    constexpr int DPT = W_PIECES + R_PIECES;
    constexpr int STAGE = ROWS * TC + 16 * NB * TC;
    static_assert(DPT * (S - 1) < 64 && S >= 2 && S <= 4, "ring depth");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row_blk = blockIdx.x * ROWS;
    const int tiles = (P.n + TC - 1) / TC;
    const int supers = (P.n + 255) / 256;
    int t0 = (int)((3u * (unsigned)((P.row0 + row_blk) >> 7)) % (unsigned)supers) * (256 / TC);
    if (PROBE == 20) t0 = 0;
    if (PROBE == 21) t0 = (int)((((unsigned)((P.row0 + row_blk) >> 7)) & 7u) * (unsigned)(supers / 8)) * (256 / TC);
    const bool ragged = (P.n % TC) != 0;

    // shared bookkeeping of the tile sequence (identical in both roles)
    int issued = 0, tile_issue = t0, tile = t0, slot = 0;
    auto advance_issue = [&]() {
        tile_issue = tile_issue + 1 == tiles ? 0 : tile_issue + 1;
        ++issued;
    };
    auto is_fast = [&]() { return issued < tiles && !(ragged && (tile == tiles - 1 || tile_issue == tiles - 1)); };

    if (wave >= 4) {
        // ---------------- loader ----------------
        const int lw = wave - 4;
        if (PROBE >= 10 && PROBE < 30) __builtin_amdgcn_s_setprio(3);  // (measured: see DESIGN.md)
        // cache policy of the two DMA streams (measurement build: PROBE 30..37; bits: 1 = sc0, 2 = nt, 16 = sc1)
        static constexpr int W_AUX = PROBE == 30 ? 18 : PROBE == 31 ? 19 : PROBE == 32 ? 17 : PROBE == 33 ? 0 : PROBE == 34 ? 16 : PROBE == 37 ? 3 : 2;
        static constexpr int R_AUX = PROBE == 35 ? 1 : PROBE == 36 ? 16 : 0;
        const float* src[DPT];  // per-lane source of piece k (W pieces first), without the tile's column offset
        int quad[DPT];
#pragma clang loop unroll(full)
        for (int k = 0; k < W_PIECES; ++k) {
            const int Q = 64 * (lw * W_PIECES + k) + lane, row = Q / QPR, qp = Q % QPR;
            int grow = row_blk + row;
            grow = grow < P.rows ? grow : P.rows - 1;
            quad[k] = 4 * (qp ^ (row & 15));
            src[k] = P.W + (int64_t)grow * P.ldw + quad[k];
        }
#pragma clang loop unroll(full)
        for (int k = 0; k < R_PIECES; ++k) {
            const int Q = 64 * (lw * R_PIECES + k) + lane, srow = Q / QPR, qp = Q % QPR;
            const int sl = srow < batch ? srow : batch - 1;
            quad[W_PIECES + k] = 4 * (qp ^ (srow & 15));
            src[W_PIECES + k] = P.src_cur + (int64_t)sl * P.ldsrc + quad[W_PIECES + k];
        }
        auto piece_lds = [&](int st, int k) -> float* {
            float* stage = lds16 + st * STAGE;
            return k < W_PIECES ? stage + 256 * (lw * W_PIECES + k) : stage + ROWS * TC + 256 * (lw * R_PIECES + (k - W_PIECES));
        };
        auto issue_tile = [&](auto fast_tag, int tl, int st) {
            const int c0 = tl * TC;
            if (!decltype(fast_tag)::value && ragged && tl == tiles - 1) {
#pragma clang loop unroll(full)
                for (int k = 0; k < DPT; ++k) {
                    rp_f4 w = rp_f4{0.f, 0.f, 0.f, 0.f};
                    if (c0 + quad[k] < P.n) w = k < W_PIECES ? rp_ld4_stream(src[k] + c0) : rp_ld4(src[k] + c0);
                    *reinterpret_cast<rp_f4*>(piece_lds(st, k) + 4 * lane) = w;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                return;
            }
#pragma clang loop unroll(full)
            for (int k = 0; k < DPT; ++k) {
                if (k < W_PIECES)
                    __builtin_amdgcn_global_load_lds((rp_gptr)(src[k] + c0), (rp_lptr)piece_lds(st, k), 16, 0, W_AUX /* nt */);
                else
                    __builtin_amdgcn_global_load_lds((rp_gptr)(src[k] + c0), (rp_lptr)piece_lds(st, k), 16, 0, R_AUX);
            }
        };
        int slot_issue = 0;
        for (; issued < S && issued < tiles; advance_issue()) {
            issue_tile(std::false_type{}, tile_issue, slot_issue);
            slot_issue = slot_issue + 1 == S ? 0 : slot_issue + 1;
        }
        {
            const int later = issued - 1;
            if (later >= 3) wait_vmcnt<(S >= 4 ? 3 * DPT : 0)>();
            else if (later == 2) wait_vmcnt<(S >= 3 ? 2 * DPT : 0)>();
            else if (later == 1) wait_vmcnt<DPT>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        for (int i = 0; i < tiles; ++i) {
            const bool fast = is_fast();
            if (fast) {
                wait_vmcnt<(S - 2) * DPT>();  // tile i+1 has landed; the S-2 tiles issued after it may be in flight
                __builtin_amdgcn_s_barrier();
                if (PROBE != 2) issue_tile(std::true_type{}, tile_issue, slot);
                advance_issue();
            } else {
                if (i + 1 < tiles) {
                    const int later = issued - 2 - i;
                    if (later >= 2) wait_vmcnt<(S >= 4 ? 2 * DPT : 0)>();
                    else if (later == 1) wait_vmcnt<(S >= 3 ? DPT : 0)>();
                    else wait_vmcnt<0>();
                }
                __builtin_amdgcn_s_barrier();
                if (issued < tiles) {
                    issue_tile(std::false_type{}, tile_issue, slot);
                    advance_issue();
                }
            }
            tile = tile + 1 == tiles ? 0 : tile + 1;
            slot = slot + 1 == S ? 0 : slot + 1;
        }
        return;
    }

    // ---------------- multiplier ----------------
    struct Frags {
        f32x4 a[RB][G], b[NB][G];
    };
    f32x4 acc[RB][NB];
#pragma clang loop unroll(full)
    for (int rb = 0; rb < RB; ++rb)
#pragma clang loop unroll(full)
        for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, kq = lane >> 4;
    auto read_group = [&](Frags& f, int st, int j) {
        const float* stage = lds16 + st * STAGE;
        const int off = r * TC + 4 * ((4 * j + kq) ^ r);
#pragma clang loop unroll(full)
        for (int rb = 0; rb < RB; ++rb) f.a[rb][j] = *reinterpret_cast<const f32x4*>(stage + (32 * wave + 16 * rb) * TC + off);
#pragma clang loop unroll(full)
        for (int nb = 0; nb < NB; ++nb) f.b[nb][j] = *reinterpret_cast<const f32x4*>(stage + ROWS * TC + nb * 16 * TC + off);
    };
    for (; issued < S && issued < tiles;) advance_issue();  // (mirror of the loaders' prologue)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    Frags f0, f1;
#pragma clang loop unroll(full)
    for (int j = 0; j < G; ++j) read_group(f0, 0, j);

    auto iteration = [&](auto fast_tag, int i, Frags& cur, Frags& nxt) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const bool more = FAST || i + 1 < tiles;
        const int slot1 = slot + 1 == S ? 0 : slot + 1;
#pragma clang loop unroll(full)
        for (int j = 0; j < G; ++j) {  // this tile's fragments have arrived (their stage is overwritten after the barrier)
#pragma clang loop unroll(full)
            for (int rb = 0; rb < RB; ++rb) asm volatile("" ::"v"(cur.a[rb][j]));
#pragma clang loop unroll(full)
            for (int nb = 0; nb < NB; ++nb) asm volatile("" ::"v"(cur.b[nb][j]));
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int groups_here = FAST ? G : (P.n - tile * TC + 15) >> 4;
#pragma clang loop unroll(full)
        for (int j = 0; j < G; ++j) {
            if (more) read_group(nxt, slot1, j);
            __builtin_amdgcn_sched_barrier(0);
            if (PROBE == 1) {
#pragma clang loop unroll(full)
                for (int rb = 0; rb < RB; ++rb) asm volatile("" ::"v"(cur.a[rb][j]));
#pragma clang loop unroll(full)
                for (int nb = 0; nb < NB; ++nb) asm volatile("" ::"v"(cur.b[nb][j]));
            } else if (j < groups_here) {
#pragma clang loop unroll(full)
                for (int c = 0; c < 4; ++c)
#pragma clang loop unroll(full)
                    for (int rb = 0; rb < RB; ++rb)
#pragma clang loop unroll(full)
                        for (int nb = 0; nb < NB; ++nb)
                            acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[rb][j][c], cur.b[nb][j][c], acc[rb][nb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FAST || issued < tiles) advance_issue();
        tile = tile + 1 == tiles ? 0 : tile + 1;
        slot = slot1;
    };
    auto step = [&](int i, Frags& cur, Frags& nxt) {
        if (is_fast()) iteration(std::true_type{}, i, cur, nxt);
        else iteration(std::false_type{}, i, cur, nxt);
    };
    for (int i = 0; i < tiles; i += 2) {
        step(i, f0, f1);
        if (i + 1 < tiles) step(i + 1, f1, f0);
    }
#pragma clang loop unroll(full)
    for (int rb = 0; rb < RB; ++rb)
#pragma clang loop unroll(full)
        for (int nb = 0; nb < NB; ++nb) {
            const int b = 16 * nb + r;
            if (b >= batch) continue;
            const int i0 = row_blk + 32 * wave + 16 * rb + 4 * kq;
#pragma clang loop unroll(full)
            for (int e = 0; e < 4; ++e)
                if (i0 + e < P.rows) epilogue<MODEL>(P, i0 + e, b, acc[rb][nb][e]);
        }
}

template <int MODEL, int TC, int S, int PROBE = 0>
int launch_mfma16_split(const BatchParams& P, int B, hipStream_t st) {
    constexpr size_t lds = sizeof(float) * S * (128 * TC + 32 * TC);
    static_assert(lds <= 160 * 1024, "LDS");
    if (rp_allow_dynamic_lds<&step_batched_mfma16_split_kernel<MODEL, TC, S, PROBE>>(lds) != RP_OK) {
        (void)hipGetLastError();
        return RP_ERR_HIP;
    }
    const int blocks = (P.rows + 127) / 128;
    hipLaunchKernelGGL((step_batched_mfma16_split_kernel<MODEL, TC, S, PROBE>), dim3(blocks), dim3(512), lds, st, P, B);
    return RP_OK;
}

template <int MODEL, int NB, int TC, int S, int WAVES, int PROBE = 0>
int launch_mfma16(const BatchParams& P, int B, hipStream_t st) {
    constexpr size_t lds = sizeof(float) * S * (16 * WAVES * TC + 16 * NB * TC);
    if constexpr (lds > 160 * 1024) {
        return RP_ERR_SHAPE;  // this instance does not fit the 160 KiB of LDS
    } else {
        if (rp_allow_dynamic_lds<&step_batched_mfma16_kernel<MODEL, NB, TC, S, WAVES, PROBE>>(lds) != RP_OK) {
            (void)hipGetLastError();
            return RP_ERR_HIP;
        }
        const int blocks = (P.rows + 16 * WAVES - 1) / (16 * WAVES);
        hipLaunchKernelGGL((step_batched_mfma16_kernel<MODEL, NB, TC, S, WAVES, PROBE>), dim3(blocks), dim3(64 * WAVES), lds, st, P, B);
        return RP_OK;
    }
}

#ifdef RP_MEASUREMENT
// measurement build only (rp_set_batched_mfma_tuning): tile columns, ring stages, waves per workgroup
std::atomic<int> g_m16_tc{0}, g_m16_stages{0}, g_m16_waves{0};
#endif

template <int MODEL, int NB>
int launch_mfma16_cfg(const BatchParams& P, int B, hipStream_t st) {
    // defaults (scripts/archive/sweep_mfma16.py, r03; ms per step, (tile columns, stages, waves), waves = 16: the role-split kernel):
    //   one slot block (B <= 16)   N = 8192: (64,3,4) 0.067, (64,2,4) 0.107;  16384: 0.156 / 0.231;  32768: 0.621 / 0.618;
    //                              65536: 2.69 / 2.69  -> three stages everywhere (small N has one workgroup per CU)
    //   two slot blocks (B <= 32)  N = 16384: (64,3,4) 0.206, (64,3,8) 0.357, split 0.307;  24576: 0.524 / 0.531 / 0.464;
    //                              32768: 0.82 / 0.81 / 0.745;  65536: 3.28 / 3.21 / 2.99  -> the split kernel from 192 workgroups
    int tc = 64, stages = 3, waves = 4;
    if (NB == 2 && P.rows >= 128 * 192) waves = 16;
#ifndef RP_MEASUREMENT
    (void)tc;
    (void)stages;
    if constexpr (NB == 2)
        if (waves == 16) return launch_mfma16_split<MODEL, 64, 3>(P, B, st);
    return launch_mfma16<MODEL, NB, 64, 3, 4>(P, B, st);
#else
    if (g_m16_tc > 0) tc = g_m16_tc;
    if (g_m16_stages > 0) stages = g_m16_stages;
    if (g_m16_waves > 0) waves = g_m16_waves;
    switch (waves * 10000 + tc * 10 + stages) {
        case 40642: return launch_mfma16<MODEL, NB, 64, 2, 4>(P, B, st);
        case 40643: return launch_mfma16<MODEL, NB, 64, 3, 4>(P, B, st);
        case 40644: return launch_mfma16<MODEL, NB, 64, 4, 4>(P, B, st);
        case 41282: return launch_mfma16<MODEL, NB, 128, 2, 4>(P, B, st);
        case 41283: return launch_mfma16<MODEL, NB, 128, 3, 4>(P, B, st);
        case 42562: return launch_mfma16<MODEL, NB, 256, 2, 4>(P, B, st);
        default: break;
    }
    if constexpr (NB == 2 && MODEL == RP_MODEL_LI_TANH) {  // measurement builds: halves of the kernel switched off
        switch (waves * 10000 + tc * 10 + stages) {
            case 140643: return launch_mfma16<MODEL, NB, 64, 3, 4, 1>(P, B, st);
            case 240643: return launch_mfma16<MODEL, NB, 64, 3, 4, 2>(P, B, st);
            case 180643: return launch_mfma16<MODEL, NB, 64, 3, 8, 1>(P, B, st);
            case 280643: return launch_mfma16<MODEL, NB, 64, 3, 8, 2>(P, B, st);
            case 340643: return launch_mfma16<MODEL, NB, 64, 3, 4, 3>(P, B, st);
            case 380643: return launch_mfma16<MODEL, NB, 64, 3, 8, 3>(P, B, st);
            default: break;
        }
    }
    if constexpr (NB == 2) {  // waves = 16: the role-split kernel (4 multiplier + 4 loader waves, 128 rows)
        switch (waves * 10000 + tc * 10 + stages) {
            case 160643: return launch_mfma16_split<MODEL, 64, 3>(P, B, st);
            case 160644: return launch_mfma16_split<MODEL, 64, 4>(P, B, st);
            case 160642: return launch_mfma16_split<MODEL, 64, 2>(P, B, st);
            default: break;
        }
        if constexpr (MODEL == RP_MODEL_LI_TANH) {
            switch (waves * 10000 + tc * 10 + stages) {  // measurement builds
                case 1160643: return launch_mfma16_split<MODEL, 64, 3, 1>(P, B, st);
                case 2160643: return launch_mfma16_split<MODEL, 64, 3, 2>(P, B, st);
                case 1160644: return launch_mfma16_split<MODEL, 64, 4, 1>(P, B, st);
                case 2160644: return launch_mfma16_split<MODEL, 64, 4, 2>(P, B, st);
                case 10160643: return launch_mfma16_split<MODEL, 64, 3, 10>(P, B, st);
                case 20160643: return launch_mfma16_split<MODEL, 64, 3, 20>(P, B, st);
                case 21160643: return launch_mfma16_split<MODEL, 64, 3, 21>(P, B, st);
                case 30160643: return launch_mfma16_split<MODEL, 64, 3, 30>(P, B, st);
                case 31160643: return launch_mfma16_split<MODEL, 64, 3, 31>(P, B, st);
                case 32160643: return launch_mfma16_split<MODEL, 64, 3, 32>(P, B, st);
                case 33160643: return launch_mfma16_split<MODEL, 64, 3, 33>(P, B, st);
                case 34160643: return launch_mfma16_split<MODEL, 64, 3, 34>(P, B, st);
                case 35160643: return launch_mfma16_split<MODEL, 64, 3, 35>(P, B, st);
                case 36160643: return launch_mfma16_split<MODEL, 64, 3, 36>(P, B, st);
                case 37160643: return launch_mfma16_split<MODEL, 64, 3, 37>(P, B, st);
                default: break;
            }
        }
    }
    if constexpr (NB == 2) {  // eight waves share one source tile: every wave brings one 1 KiB piece of it per 64 columns
        switch (waves * 10000 + tc * 10 + stages) {
            case 80642: return launch_mfma16<MODEL, NB, 64, 2, 8>(P, B, st);
            case 80643: return launch_mfma16<MODEL, NB, 64, 3, 8>(P, B, st);
            case 81282: return launch_mfma16<MODEL, NB, 128, 2, 8>(P, B, st);
            default: break;
        }
    }
    return RP_ERR_ARG;
#endif
}

template <int MODEL>
int launch_model(const BatchParams& P, int B, hipStream_t st) {
    // R * B accumulator sets of four floats per lane; W chunks in flight per wave: R * U KiB
#ifdef RP_MEASUREMENT
    if constexpr (MODEL == RP_MODEL_LI_TANH) {  // sweep (rp_set_batched_tuning)
        const int key = B * 100 + g_bt_R * 10 + g_bt_U;
        switch (key) {
            case 221: launch<MODEL, 2, 2, 1>(P, st); return RP_OK;
            case 222: launch<MODEL, 2, 2, 2>(P, st); return RP_OK;
            case 241: launch<MODEL, 4, 2, 1>(P, st); return RP_OK;
            case 242: launch<MODEL, 4, 2, 2>(P, st); return RP_OK;
            case 421: launch<MODEL, 2, 4, 1>(P, st); return RP_OK;
            case 422: launch<MODEL, 2, 4, 2>(P, st); return RP_OK;
            case 821: launch<MODEL, 2, 8, 1>(P, st); return RP_OK;
            case 841: launch<MODEL, 4, 8, 1>(P, st); return RP_OK;
            case 811: launch<MODEL, 1, 8, 1>(P, st); return RP_OK;
            case 812: launch<MODEL, 1, 8, 2>(P, st); return RP_OK;
            case 481: launch<MODEL, 8, 4, 1>(P, st); return RP_OK;
            case 482: launch<MODEL, 8, 4, 2>(P, st); return RP_OK;
            case 441: launch<MODEL, 4, 4, 1>(P, st); return RP_OK;
            case 822: launch<MODEL, 2, 8, 2>(P, st); return RP_OK;
            case 824: launch<MODEL, 2, 8, 4>(P, st); return RP_OK;
            case 842: launch<MODEL, 4, 8, 2>(P, st); return RP_OK;
            case 1621: launch<MODEL, 2, 16, 1>(P, st); return RP_OK;
            case 1614: launch<MODEL, 1, 16, 4>(P, st); return RP_OK;
            case 1612: launch<MODEL, 1, 16, 2>(P, st); return RP_OK;
            default: break;
        }
    }
#endif
    switch (B) {
        // r02 (scripts/archive/sweep_batched.py, N = 32768, ms per step): fewer resident workgroups, as for the unbatched walk --
        // B=2 (4,2) on 2048 0.627, (4,1) on 768 0.618;  B=4 (8,2) on 2048 0.639, on 512 0.629;  B=8 flat (0.86)
        case 2: launch<MODEL, 4, 2, 1>(P, st, 768); return RP_OK;
        // measured at N = 32768 (scripts/bench_batched.py): B=2 1.98x, B=4 3.91x, B=8 5.74x, B=16 5.92x the throughput of
        // B unbatched runs; from B = 8 the L2 -> CU path (B source loads per R rows of W) and the register budget bind
        case 4: launch<MODEL, 8, 4, 2>(P, st, 512); return RP_OK;
        case 8: launch<MODEL, 4, 8, 1>(P, st); return RP_OK;
        case 16: launch<MODEL, 2, 16, 1>(P, st); return RP_OK;
        default: return RP_ERR_ARG;
    }
}

}  // namespace

extern "C" {

int rp_sizeof_batch(void) { return (int)sizeof(rp_batch); }

#ifdef RP_MEASUREMENT
int rp_set_batched_grid(int max_workgroups) {
    g_bt_blocks = max_workgroups;
    return RP_OK;
}

int rp_set_batched_mfma_tuning(int tile_columns, int stages, int waves) {
    g_m16_tc = tile_columns;
    g_m16_stages = stages;
    g_m16_waves = waves;
    return RP_OK;
}

int rp_set_batched_tuning(int rows_per_wave, int unroll) {
    g_bt_R = rows_per_wave;
    g_bt_U = unroll;
    return RP_OK;
}
#endif

int rp_forward_batched(const rp_net* net, rp_batch* bt, void* stream) {
    if (!net || !bt) RP_FAIL(RP_ERR_ARG, "net / batch is null");
    if (net->model != RP_MODEL_LI_TANH && net->model != RP_MODEL_LI_SIGMOID)
        RP_FAIL(RP_ERR_ARG, "rp_forward_batched covers the rate models (LI_TANH, LI_SIGMOID), got model %d", net->model);
    const int B = bt->batch;
    const bool mfma = bt->layout == RP_BATCH_MFMA;
    const bool mfma16 = bt->layout == RP_BATCH_MFMA16;
    if (bt->layout != RP_BATCH_VALU && !mfma && !mfma16) RP_FAIL(RP_ERR_ARG, "unknown batch layout %d", bt->layout);
    if ((mfma || mfma16) && (B < 1 || B > RP_BATCH_SLOTS))
        RP_FAIL(RP_ERR_ARG, "the matrix-core forms hold 1..%d trajectories, got %d", RP_BATCH_SLOTS, B);
    if (!mfma && !mfma16 && B != 2 && B != 4 && B != 8 && B != 16) RP_FAIL(RP_ERR_ARG, "batch must be 2, 4, 8 or 16, got %d", B);
    if (mfma16 && net->row0 % 128 != 0) RP_FAIL(RP_ERR_ARG, "row0=%d must be a multiple of 128 for the 16x16x4 matrix-core form", net->row0);
    if (net->n <= 0 || net->rows <= 0 || net->row0 < 0 || net->row0 + net->rows > net->n)
        RP_FAIL(RP_ERR_SHAPE, "bad shard geometry n=%d rows=%d row0=%d", net->n, net->rows, net->row0);
    if (net->row0 % 8 != 0) RP_FAIL(RP_ERR_ARG, "row0=%d must be a multiple of 8", net->row0);
    if (net->n % 4 != 0 || net->ldw % 4 != 0 || net->ldw < net->n)
        RP_FAIL(RP_ERR_SHAPE, "the batched step needs n %% 4 == 0 and ldw %% 4 == 0 (n=%d, ldw=%lld)", net->n, (long long)net->ldw);
    if (!net->W || !bt->v || !bt->src[0] || !bt->src[1] || bt->src[0] == bt->src[1])
        RP_FAIL(RP_ERR_ARG, "W, v, src[0], src[1] must be non-null (and the two source buffers distinct)");
    if (!mfma && (bt->ldv < net->rows || bt->ldsrc < net->n || bt->ldsrc % 4 != 0))
        RP_FAIL(RP_ERR_SHAPE, "ldv >= rows, ldsrc >= n and ldsrc %% 4 == 0 required");
    if (mfma && net->row0 % kRows != 0) RP_FAIL(RP_ERR_ARG, "row0=%d must be a multiple of %d for the matrix-core form", net->row0, kRows);
    if ((reinterpret_cast<uintptr_t>(net->W) & 15u) || (reinterpret_cast<uintptr_t>(bt->src[0]) & 15u) ||
        (reinterpret_cast<uintptr_t>(bt->src[1]) & 15u))
        RP_FAIL(RP_ERR_ARG, "W and the source buffers must be 16-byte aligned");
    if (bt->cur != 0 && bt->cur != 1) RP_FAIL(RP_ERR_ARG, "cur must be 0 or 1");
    if (bt->steps < 0) RP_FAIL(RP_ERR_ARG, "steps < 0");
    if (bt->x_mode != RP_X_NONE && bt->x_mode != RP_X_FULL && bt->x_mode != RP_X_BCAST && bt->x_mode != RP_X_PROJ)
        RP_FAIL(RP_ERR_ARG, "x_mode must be RP_X_NONE, RP_X_FULL, RP_X_BCAST or RP_X_PROJ");
    if (bt->x_mode == RP_X_PROJ) {
        if (!bt->x || !bt->w_in || bt->m < 1 || bt->ldx < bt->m || bt->ldwin < bt->m)
            RP_FAIL(RP_ERR_SHAPE, "RP_X_PROJ needs x (ldx >= m), w_in (ldwin >= m), m >= 1");
    } else if (bt->x_mode != RP_X_NONE && (!bt->x || bt->ldx < (bt->x_mode == RP_X_FULL ? net->rows : 1)))
        RP_FAIL(RP_ERR_SHAPE, "inputs missing or rows too short");
    const bool sampling = bt->samples || bt->rec[0] || bt->rec[1];
    if (sampling) {
        if (!bt->acc || bt->ldacc < net->rows) RP_FAIL(RP_ERR_ARG, "sampling needs the acc buffer with ldacc >= rows");
        if (bt->ldsamples < net->rows) RP_FAIL(RP_ERR_SHAPE, "ldsamples < rows");
        if (bt->sampling_steps < 1 || bt->cutoff < 0 || bt->step0 < 0 || bt->acc_count < 0 || bt->sample_idx < 0)
            RP_FAIL(RP_ERR_ARG, "sampling_steps >= 1 and cutoff, step0, acc_count, sample_idx >= 0 required");
        int64_t produced = 0;
        for (int64_t t = bt->step0; t < bt->step0 + bt->steps; ++t)
            if (t >= bt->cutoff && t % bt->sampling_steps == 0) ++produced;
        if (bt->sample_idx + produced > bt->max_samples)
            RP_FAIL(RP_ERR_SHAPE, "sample buffers hold %d rows per trajectory, call needs %lld", bt->max_samples,
                    (long long)(bt->sample_idx + produced));
    }
    if (bt->traj && bt->ldtraj < net->rows) RP_FAIL(RP_ERR_SHAPE, "ldtraj < rows");
    const int need[] = {RP_P_TAU, RP_P_K, RP_P_ETA};
    for (int q = 0; q < 3; ++q)
        if (!net->p[need[q]]) RP_FAIL(RP_ERR_ARG, "parameter slot %d is null", need[q]);
    if (net->model == RP_MODEL_LI_SIGMOID && (!net->p[RP_P_RMAX] || !net->p[RP_P_SLOPE] || !net->p[RP_P_V0]))
        RP_FAIL(RP_ERR_ARG, "sigmoid parameters missing");
    if (!(net->dt > 0.f)) RP_FAIL(RP_ERR_ARG, "dt must be > 0");

    hipStream_t st = static_cast<hipStream_t>(stream);
    BatchParams P = {};
    P.W = net->W;
    P.ldw = net->ldw;
    P.n = net->n;
    P.rows = net->rows;
    P.row0 = net->row0;
    for (int q = 0; q < RP_MAX_PARAMS; ++q) P.p[q] = net->p[q];
    P.dt = net->dt;
    P.v = bt->v;
    P.ldv = bt->ldv;
    P.ldsrc = bt->ldsrc;
    P.x_mode = bt->x_mode;
    P.ldx = bt->ldx;
    P.out_var = bt->out_var;
    P.ldtraj = bt->ldtraj;
    P.m = bt->m;
    P.w_in = bt->w_in;
    P.ldwin = bt->ldwin;
    P.acc = bt->acc;
    P.ldacc = bt->ldacc;
    P.sample_stride = (int64_t)bt->max_samples * bt->ldsamples;
    for (int t = 0; t < bt->steps; ++t) {
        P.src_cur = bt->src[bt->cur];
        P.src_next = bt->src[bt->cur ^ 1];
        P.x_t = bt->x ? bt->x + (int64_t)t * B * bt->ldx : nullptr;
        P.traj_t = bt->traj ? bt->traj + (int64_t)t * B * bt->ldtraj : nullptr;
        P.flags = 0;
        P.sample_row = P.rec_row[0] = P.rec_row[1] = nullptr;
        if (const int64_t t_abs = bt->step0 + t; sampling && t_abs >= bt->cutoff) {  // (network.py:590-597, as rp_forward)
            P.flags |= FLAG_BUFFER;
            if (bt->acc_count == 0) P.flags |= FLAG_FIRST;
            bt->acc_count += 1;
            if (t_abs % bt->sampling_steps == 0) {
                P.flags |= FLAG_SAMPLE;
                P.count = (float)bt->acc_count;
                const int64_t off = (int64_t)bt->sample_idx * bt->ldsamples;
                P.sample_row = bt->samples ? bt->samples + off : nullptr;
                for (int q = 0; q < 2; ++q) P.rec_row[q] = bt->rec[q] ? bt->rec[q] + off : nullptr;
                bt->sample_idx += 1;
                bt->acc_count = 0;
            }
        }
        if (mfma16) {
            int rc;
            if (net->model == RP_MODEL_LI_TANH)
                rc = B <= 16 ? launch_mfma16_cfg<RP_MODEL_LI_TANH, 1>(P, B, st) : launch_mfma16_cfg<RP_MODEL_LI_TANH, 2>(P, B, st);
            else
                rc = B <= 16 ? launch_mfma16_cfg<RP_MODEL_LI_SIGMOID, 1>(P, B, st)
                             : launch_mfma16_cfg<RP_MODEL_LI_SIGMOID, 2>(P, B, st);
            if (rc != RP_OK) RP_FAIL(rc, "no 16x16x4 batched kernel for this configuration (batch %d)", B);
        } else if (mfma) {
            const int blocks = (net->rows + kRows - 1) / kRows;
            if (net->model == RP_MODEL_LI_TANH)
                hipLaunchKernelGGL(step_batched_mfma_kernel<RP_MODEL_LI_TANH>, dim3(blocks), dim3(kThreads), 0, st, P, B);
            else
                hipLaunchKernelGGL(step_batched_mfma_kernel<RP_MODEL_LI_SIGMOID>, dim3(blocks), dim3(kThreads), 0, st, P, B);
        } else {
            const int rc = net->model == RP_MODEL_LI_TANH ? launch_model<RP_MODEL_LI_TANH>(P, B, st)
                                                          : launch_model<RP_MODEL_LI_SIGMOID>(P, B, st);
            if (rc != RP_OK) RP_FAIL(rc, "no batched kernel for batch %d", B);
        }
        if (t == 0) RP_HIP_CHECK(hipGetLastError());
        bt->cur ^= 1;
    }
    RP_HIP_CHECK(hipGetLastError());
    return RP_OK;
}

}  // extern "C"
