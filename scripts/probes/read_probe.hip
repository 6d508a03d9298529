// Stand-alone probe (not part of the library): read-only streaming of a 4 GiB buffer on gfx950 with different walks.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/read_probe scripts/probes/read_probe.hip && /tmp/read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sum4(f4 v) { return (v.x + v.y) + (v.z + v.w); }

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_oneshot(const f4* buf, int64_t n4, float* out) {
    const int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = base + 256 * u < n4 ? (NT ? __builtin_nontemporal_load(buf + base + 256 * u) : buf[base + 256 * u]) : f4{0, 0, 0, 0};
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += sum4(v[u]);
    if (s == 123.456f) out[0] = s;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_persist(const f4* buf, int64_t n4, float* out) {
    const int64_t tiles = (n4 + 256 * U - 1) / (256 * U);
    float s = 0.f;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t base = t * 256 * U + threadIdx.x;
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = base + 256 * u < n4 ? (NT ? __builtin_nontemporal_load(buf + base + 256 * u) : buf[base + 256 * u]) : f4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) s += sum4(v[u]);
    }
    if (s == 123.456f) out[0] = s;
}

// the step kernel's walk: a wave owns R rows, U chunks of 1 KiB per row in flight
template <int R, int U, bool NT>
__global__ __launch_bounds__(256) void k_rows(const f4* buf, int n, float* out, int groups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n4 = n / 4;
    float s = 0.f;
    for (int g = blockIdx.x * 4 + wave; g < groups; g += gridDim.x * 4) {
        const f4* r0 = buf + (int64_t)(R * g) * n4;
        for (int c = lane; c < n4; c += 64 * U) {
            f4 a[R][U];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < R; ++j) a[j][u] = c + 64 * u < n4 ? (NT ? __builtin_nontemporal_load(r0 + (int64_t)j * n4 + c + 64 * u) : r0[(int64_t)j * n4 + c + 64 * u]) : f4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < R; ++j) s += sum4(a[j][u]);
        }
    }
    if (s == 123.456f) out[0] = s;
}

// one-shot tiles WITH the arithmetic of a mat-vec: a wave owns RW rows x (256*CQ) columns, keeps its slice of the source
// vector in registers, and writes one partial sum per row; consecutive workgroups walk along the row (COLMAJOR false)
// or down the rows (true)
template <int RW, int CQ, bool NT, bool COLMAJOR>
__global__ __launch_bounds__(256) void k_tile_dot(const f4* W, const f4* src, int n, float* partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n4 = n / 4, tiles_c = n4 / (64 * CQ), groups = n / (4 * RW);
    const int tile = COLMAJOR ? blockIdx.x / groups : blockIdx.x % tiles_c;
    const int grp = COLMAJOR ? blockIdx.x % groups : blockIdx.x / tiles_c;
    const int row0 = (grp * 4 + wave) * RW;
    const int c0 = tile * 64 * CQ + lane;
    f4 s[CQ], a[RW][CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) s[q] = src[c0 + 64 * q];
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const f4* p = W + (int64_t)(row0 + j) * n4 + c0 + 64 * q;
            a[j][q] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const f4 w = a[j][q], r = s[q];
            acc += (w.x * r.x + w.y * r.y) + (w.z * r.z + w.w * r.w);
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) partial[(int64_t)(row0 + j) * tiles_c + tile] = acc;
    }
}

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 7; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}

int main() {
    const int n = 32768;
    const int64_t n4 = (int64_t)n * n / 4;
    f4* buf; float* out;
    if (hipMalloc(&buf, n4 * 16) != hipSuccess || hipMalloc(&out, 16) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, n4 * 16);
    const double bytes = 1.0 * n4 * 16;
    auto rep = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
#define ONESHOT(U, L) rep("oneshot U=" #U " nt=" #L, timeit([&] { hipLaunchKernelGGL((k_oneshot<U, L>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, buf, n4, out); }))
    ONESHOT(1, true); ONESHOT(2, true); ONESHOT(4, true); ONESHOT(4, false); ONESHOT(8, true); ONESHOT(16, true);
#define PERSIST(U, L, G) rep("persist U=" #U " nt=" #L " grid=" #G, timeit([&] { hipLaunchKernelGGL((k_persist<U, L>), dim3(G), dim3(256), 0, 0, buf, n4, out); }))
    PERSIST(4, true, 2048); PERSIST(8, true, 2048); PERSIST(8, true, 4096); PERSIST(16, true, 2048); PERSIST(8, false, 2048);
#define ROWS(R, U, L, G) rep("rows R=" #R " U=" #U " nt=" #L " grid=" #G, timeit([&] { hipLaunchKernelGGL((k_rows<R, U, L>), dim3(G), dim3(256), 0, 0, buf, n, out, n / R); }))
    ROWS(4, 2, true, 2048); ROWS(4, 2, false, 2048); ROWS(2, 4, true, 2048); ROWS(1, 8, true, 2048); ROWS(4, 4, true, 2048); ROWS(8, 1, true, 1024);
    float* partial; f4* src;
    if (hipMalloc(&partial, (size_t)n * 128 * 4) != hipSuccess || hipMalloc(&src, n * 4) != hipSuccess) return 1;
    (void)hipMemset(src, 0, n * 4);
#define TILE(RW, CQ, L, CM) rep("tile-dot RW=" #RW " CQ=" #CQ " nt=" #L " down-rows=" #CM, timeit([&] { hipLaunchKernelGGL((k_tile_dot<RW, CQ, L, CM>), dim3((unsigned)(n4 / (256 * RW * CQ))), dim3(256), 0, 0, buf, src, n, partial); }))
    TILE(4, 4, true, false); TILE(4, 4, true, true); TILE(4, 4, false, false); TILE(2, 8, true, false); TILE(1, 16, true, false);
    TILE(8, 2, true, false); TILE(4, 2, true, false); TILE(4, 8, true, false); TILE(2, 4, true, false); TILE(2, 8, true, true);
    TILE(1, 8, true, false); TILE(8, 4, true, false);
    return 0;
}
