// Stand-alone probe (not part of the library): the per-step cost of the resident step kernel's hand-off -- every wave publishes RW
// tagged 8-byte granules and sweeps all n of them each step -- by memory scope and placement:
//   agent scope (sc1: through the fabric), workgroups wherever the dispatcher puts them;
//   agent scope, all workgroups on ONE XCD (only blocks whose HW_REG_XCC_ID matches take part, by ticket);
//   (workgroup scope -- sc0, the XCD's L2 as the meeting point -- was tried: the granules are never seen).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/granule_probe scripts/probes/granule_probe.hip && /tmp/granule_probe
// Every spin is bounded; a timeout sets *fail and every wave leaves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr unsigned kSpinCap = 1u << 20;

template <int SCOPE>  // 0: agent, 1: workgroup
__device__ __forceinline__ unsigned long long gload(const gu64* p) {
    if constexpr (SCOPE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int SCOPE>
__device__ __forceinline__ void gstore(gu64* p, unsigned long long v) {
    if constexpr (SCOPE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// n = 256 NC columns, RW rows per wave, 4 waves per block; `one_xcd`: only blocks on XCD `xcd` take part (ticket = work index)
template <int SCOPE, int NC, int RW>
__global__ __launch_bounds__(256) void k(unsigned long long* gran0, unsigned long long* gran1, unsigned* ctl, int steps, int one_xcd,
                                         int need_blocks, int work, float* sink, unsigned* xcc_seen) {
    __shared__ int ticket_s;
    int block = blockIdx.x;
    if (one_xcd) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
        if (threadIdx.x == 0) {
            atomicAdd(xcc_seen + xcc, 1u);
            ticket_s = xcc == (unsigned)(one_xcd - 1) ? (int)atomicAdd(ctl + 1, 1u) : need_blocks;
        }
        __syncthreads();
        block = ticket_s;
        if (block >= need_blocks) return;
    } else if (block >= need_blocks) return;
    const int lane = threadIdx.x & 63, gw = block * 4 + (threadIdx.x >> 6);
    const int n = 256 * NC;
    float acc = 1.0f + lane;
    gu32* fail = (gu32*)ctl;
    for (int t = 0; t < steps; ++t) {
        float r[4 * NC];
        if (t > 0) {
            const gu64* g = (const gu64*)((t - 1) & 1 ? gran1 : gran0) + 4 * lane;
            bool done = false;
            for (unsigned spins = 0; !done; ++spins) {
                bool ok = true;
#pragma unroll
                for (int m = 0; m < NC; ++m)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned long long q = gload<SCOPE>(g + 256 * m + c);
                        ok = ok && (unsigned)(q >> 32) == (unsigned)t;
                        r[4 * m + c] = __uint_as_float((unsigned)q);
                    }
                done = __all(ok);
                if (!done) {
                    if (spins > kSpinCap || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        if (lane == 0) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4 * NC; ++q) r[q] = 0.5f;
        }
        // stand-in for the row dots, butterflies and the epilogue: a dependent chain of `work` fmas
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4 * NC; ++q) s += r[q];
        for (int q = 0; q < work; ++q) acc = fmaf(acc, 0.999f, s * 1e-6f);
        if ((lane & (64 / RW - 1)) == 0) {
            const int row = gw * RW + lane / (64 / RW);
            gstore<SCOPE>((gu64*)(t & 1 ? gran1 : gran0) + row, ((unsigned long long)(unsigned)(t + 1) << 32) | __float_as_uint(acc));
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int SCOPE, int NC, int RW>
int run(const char* name, int one_xcd, int work) {
    const int n = 256 * NC, need_blocks = n / RW / 4, steps = 4000;
    unsigned long long* g; unsigned* ctl; float* sink; unsigned* seen;
    CHECK(hipMalloc(&g, 2 * n * 8)); CHECK(hipMalloc(&ctl, 64)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&seen, 64));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f, all[12]; unsigned failed = 0; unsigned h_seen[16];
    for (int rep = 0; rep < 12; ++rep) {
        CHECK(hipMemset(g, 0, 2 * n * 8)); CHECK(hipMemset(ctl, 0, 64)); CHECK(hipMemset(seen, 0, 64));
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k<SCOPE, NC, RW>), dim3(one_xcd ? 8 * need_blocks + 64 : need_blocks), dim3(256), 0, 0, g, g + n, ctl, steps, one_xcd, need_blocks,
                           work, sink, seen);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        CHECK(hipMemcpy(&failed, ctl, 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h_seen, seen, 64, hipMemcpyDeviceToHost));
        if (failed) break;
        all[rep] = ms * 1e3f / steps;
        if (ms < best) best = ms;
    }
    for (int i = 0; i < 12; ++i) for (int j = i + 1; j < 12; ++j) if (all[j] < all[i]) { float q = all[i]; all[i] = all[j]; all[j] = q; }
    printf("%-32s n %4d blocks %2d work %3d: %s min %.3f median %.3f max %.3f us per step", name, n, need_blocks, work, failed ? "FAILED" : "ok", all[0], all[6], all[11]);
    if (one_xcd) { printf("   blocks per XCD:"); for (int x = 0; x < 8; ++x) printf(" %u", h_seen[x]); }
    printf("\n"); fflush(stdout);
    CHECK(hipFree(g)); CHECK(hipFree(ctl)); CHECK(hipFree(sink)); CHECK(hipFree(seen));
    return failed ? 1 : 0;
}

int main() {
    // (workgroup scope -- sc0 loads and stores meeting in the XCD's L2 -- never sees the granules: measured, FAILED at the first sweep)
    for (int work : {0, 40}) {
        if (run<0, 2, 16>("agent scope, any placement", 0, work)) return 1;
        if (run<0, 2, 16>("agent scope, one XCD", 1, work)) return 1;
        if (run<0, 1, 16>("agent scope, any placement", 0, work)) return 1;
        if (run<0, 1, 16>("agent scope, one XCD", 1, work)) return 1;
        if (run<0, 3, 16>("agent scope, any placement", 0, work)) return 1;   // n = 768 with 16 rows per wave: 12 workgroups
        if (run<0, 3, 16>("agent scope, one XCD", 1, work)) return 1;
        if (run<0, 3, 8>("agent scope, any placement", 0, work)) return 1;    // ... with 8 rows per wave: 24
        if (run<0, 4, 8>("agent scope, any placement", 0, work)) return 1;    // n = 1024: 32 workgroups
        if (run<0, 4, 16>("agent scope, any placement", 0, work)) return 1;   // (16 rows per wave would not fit the registers: for reference)
        if (run<0, 4, 16>("agent scope, one XCD", 1, work)) return 1;
    }
    return 0;
}
