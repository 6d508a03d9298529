// Stand-alone probe (not part of the library): what does one grid-wide barrier cost on gfx950, compared with the
// ~4 us per step of back-to-back launches, when G workgroups exchange a small vector every step?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/barrier_probe scripts/probes/barrier_probe.hip && /tmp/barrier_probe
// Every spin loop is bounded: a barrier that does not complete sets *fail and all workgroups leave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kSpinCap = 1 << 20;

// `stride`: only workgroups with blockIdx % stride == 0 take part (stride 8 = one XCD under round-robin dispatch).
// Each participant writes `per` floats of the vector, passes the barrier, reads all n floats.
__global__ __launch_bounds__(256) void k_barrier(unsigned* ctr, int* fail, float* vec0, float* vec1, int n, int per, int steps,
                                                  int stride, int members, float* sink) {
    if (blockIdx.x % stride) return;
    const int me = blockIdx.x / stride;
    float acc = 0.f;
    for (int t = 0; t < steps; ++t) {
        float* wr = (t & 1) ? vec1 : vec0;
        if (threadIdx.x < per) __hip_atomic_store(wr + me * per + threadIdx.x, (float)(t + me) + acc * 1e-30f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(t + 1) * members;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > kSpinCap || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        for (int i = threadIdx.x; i < n; i += 256) acc += __hip_atomic_load(wr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    unsigned* ctr; int* fail; float *v0, *v1, *sink;
    CHECK(hipMalloc(&ctr, 4)); CHECK(hipMalloc(&fail, 4)); CHECK(hipMalloc(&v0, 1 << 16)); CHECK(hipMalloc(&v1, 1 << 16)); CHECK(hipMalloc(&sink, 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int steps = 2000;
    struct { int grid, stride, n; } cases[] = {{8, 1, 512},  {32, 1, 512},  {64, 1, 512},  {128, 1, 2048}, {256, 1, 4096},
                                               {64, 8, 512}, {128, 8, 512}, {256, 8, 512}, {256, 8, 2048}, {16, 8, 512}};
    for (auto c : cases) {
        const int members = (c.grid + c.stride - 1) / c.stride, per = c.n / members;
        float best = 1e30f;
        int failed = 0;
        for (int rep = 0; rep < 3 && !failed; ++rep) {
            CHECK(hipMemset(ctr, 0, 4)); CHECK(hipMemset(fail, 0, 4));
            CHECK(hipEventRecord(a));
            void* args[] = {&ctr, &fail, &v0, &v1, (void*)&c.n, (void*)&per, (void*)&steps, (void*)&c.stride, (void*)&members, &sink};
            CHECK(hipLaunchCooperativeKernel((const void*)k_barrier, dim3(c.grid), dim3(256), args, 0, 0));
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            CHECK(hipMemcpy(&failed, fail, 4, hipMemcpyDeviceToHost));
            if (ms < best) best = ms;
        }
        printf("grid %3d stride %d members %3d n %4d: %s %.3f us per step\n", c.grid, c.stride, members, c.n, failed ? "FAILED" : "ok", best * 1e3f / steps);
        fflush(stdout);
        if (failed) break;
    }
    return 0;
}
