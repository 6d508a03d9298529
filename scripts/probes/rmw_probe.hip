// Stand-alone probe (not part of the library): in-place read-modify-write streaming of a 4 GiB buffer on gfx950 with the
// access patterns an RLS pass could use.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rmw_probe scripts/probes/rmw_probe.hip && /tmp/rmw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS>
__device__ __forceinline__ void rmw(f4* p, float s) {
    f4 v = NTL ? __builtin_nontemporal_load(p) : *p;
    v.x = v.x - s * v.x; v.y = v.y - s * v.y; v.z = v.z - s * v.z; v.w = v.w - s * v.w;
    if (NTS) __builtin_nontemporal_store(v, p); else *p = v;
}

// A: one-shot blocks, each thread U consecutive-by-block-stride float4 (torch-like): block covers 256*U*16 B contiguous
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_oneshot(f4* buf, int64_t n4, float s) {
    const int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + 256 * u < n4) v[u] = NTL ? __builtin_nontemporal_load(buf + base + 256 * u) : buf[base + 256 * u];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + 256 * u < n4) {
        v[u].x -= s * v[u].x; v[u].y -= s * v[u].y; v[u].z -= s * v[u].z; v[u].w -= s * v[u].w;
        if (NTS) __builtin_nontemporal_store(v[u], buf + base + 256 * u); else buf[base + 256 * u] = v[u];
    }
}

// B: persistent grid-stride over the same 256*U tiles (address order)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_persist(f4* buf, int64_t n4, float s) {
    const int64_t tiles = (n4 + 256 * U - 1) / (256 * U);
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t base = t * 256 * U + threadIdx.x;
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + 256 * u < n4) v[u] = NTL ? __builtin_nontemporal_load(buf + base + 256 * u) : buf[base + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + 256 * u < n4) {
            v[u].x -= s * v[u].x; v[u].y -= s * v[u].y; v[u].z -= s * v[u].z; v[u].w -= s * v[u].w;
            if (NTS) __builtin_nontemporal_store(v[u], buf + base + 256 * u); else buf[base + 256 * u] = v[u];
        }
    }
}

// C: row-per-wave (the RLS pass pattern): wave w owns rows 2w, 2w+1 of an n x n matrix, U chunks of 1 KiB in flight per row
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_rows(f4* buf, int n, float s, int groups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n4 = n / 4;
    for (int g = blockIdx.x * 4 + wave; g < groups; g += gridDim.x * 4) {
        f4* r0 = buf + (int64_t)(2 * g) * n4;
        f4* r1 = r0 + n4;
        for (int c = lane; c < n4; c += 64 * U) {
            f4 a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) if (c + 64 * u < n4) {
                a[u] = NTL ? __builtin_nontemporal_load(r0 + c + 64 * u) : r0[c + 64 * u];
                b[u] = NTL ? __builtin_nontemporal_load(r1 + c + 64 * u) : r1[c + 64 * u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) if (c + 64 * u < n4) {
                a[u].x -= s * a[u].x; a[u].y -= s * a[u].y; a[u].z -= s * a[u].z; a[u].w -= s * a[u].w;
                b[u].x -= s * b[u].x; b[u].y -= s * b[u].y; b[u].z -= s * b[u].z; b[u].w -= s * b[u].w;
                if (NTS) { __builtin_nontemporal_store(a[u], r0 + c + 64 * u); __builtin_nontemporal_store(b[u], r1 + c + 64 * u); }
                else { r0[c + 64 * u] = a[u]; r1[c + 64 * u] = b[u]; }
            }
        }
    }
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}

int main() {
    const int n = 32768;
    const int64_t n4 = (int64_t)n * n / 4;
    f4* buf; CHECK(hipMalloc(&buf, n4 * 16)); CHECK(hipMemset(buf, 0, n4 * 16));
    const double bytes = 2.0 * n4 * 16;
    auto rep = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
#define ONESHOT(U, L, S) rep("oneshot U=" #U " ntl=" #L " nts=" #S, timeit([&] { hipLaunchKernelGGL((k_oneshot<U, L, S>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, buf, n4, 1e-9f); }))
    ONESHOT(4, false, false); ONESHOT(4, true, true); ONESHOT(4, true, false); ONESHOT(4, false, true);
    ONESHOT(1, false, false); ONESHOT(2, false, false); ONESHOT(8, false, false); ONESHOT(8, true, true); ONESHOT(16, false, false);
#define PERSIST(U, L, S, G) rep("persist U=" #U " ntl=" #L " nts=" #S " grid=" #G, timeit([&] { hipLaunchKernelGGL((k_persist<U, L, S>), dim3(G), dim3(256), 0, 0, buf, n4, 1e-9f); }))
    PERSIST(4, false, false, 2048); PERSIST(4, true, true, 2048); PERSIST(4, true, true, 8192); PERSIST(8, true, true, 2048); PERSIST(8, false, false, 4096);
#define ROWS(U, L, S, G) rep("rows R=2 U=" #U " ntl=" #L " nts=" #S " grid=" #G, timeit([&] { hipLaunchKernelGGL((k_rows<U, L, S>), dim3(G), dim3(256), 0, 0, buf, n, 1e-9f, n / 2); }))
    ROWS(4, true, true, 2048); ROWS(4, false, false, 2048); ROWS(4, true, true, 4096); ROWS(8, true, true, 2048); ROWS(2, true, true, 4096);
    // r02: how much to keep in flight.  Persistent grids of k workgroups per CU, and one-shot grids whose residency is
    // capped at k workgroups per CU by an (unused) dynamic LDS allocation of 160 KiB / k.
#define PERSIST_G(U, G) rep("persist nt U=" #U " grid=" #G, timeit([&] { hipLaunchKernelGGL((k_persist<U, true, true>), dim3(G), dim3(256), 0, 0, buf, n4, 1e-9f); }))
    PERSIST_G(1, 512); PERSIST_G(1, 768); PERSIST_G(1, 1024); PERSIST_G(1, 1536); PERSIST_G(1, 2048); PERSIST_G(1, 4096);
    PERSIST_G(2, 256); PERSIST_G(2, 512); PERSIST_G(2, 768); PERSIST_G(2, 1024); PERSIST_G(2, 1536);
    PERSIST_G(4, 256); PERSIST_G(4, 512); PERSIST_G(4, 768); PERSIST_G(4, 1024);
    PERSIST_G(8, 256); PERSIST_G(8, 512); PERSIST_G(8, 768);
    (void)hipFuncSetAttribute((const void*)k_oneshot<4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_oneshot<1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_oneshot<2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int k : {1, 2, 3, 4, 6, 8, 12}) {
        const size_t lds = (size_t)(160 * 1024) / k - 64;
        char name[96];
        snprintf(name, sizeof name, "oneshot nt U=4, <= %d workgroups per CU", k);
        rep(name, timeit([&] { hipLaunchKernelGGL((k_oneshot<4, true, true>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), lds, 0, buf, n4, 1e-9f); }));
        snprintf(name, sizeof name, "oneshot nt U=2, <= %d workgroups per CU", k);
        rep(name, timeit([&] { hipLaunchKernelGGL((k_oneshot<2, true, true>), dim3((unsigned)((n4 + 511) / 512)), dim3(256), lds, 0, buf, n4, 1e-9f); }));
        snprintf(name, sizeof name, "oneshot nt U=1, <= %d workgroups per CU", k);
        rep(name, timeit([&] { hipLaunchKernelGGL((k_oneshot<1, true, true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), lds, 0, buf, n4, 1e-9f); }));
    }
    return 0;
}
