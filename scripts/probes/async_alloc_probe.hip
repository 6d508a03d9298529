// Which of the stream-ordered calls around a resident launch makes the host wait for the PREVIOUS launch on the stream?
// A 200 us kernel per iteration; host time per iteration with (a) hipMallocAsync + hipMemsetAsync + launch + hipFreeAsync,
// (b) a persistent buffer: hipMemsetAsync + launch, (c) the launch alone.   hipcc --offload-arch=gfx950 -O2 -o async_alloc_probe.bin async_alloc_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void spin(unsigned long long cycles, int* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
    }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const unsigned long long cycles = 20000;   // 100 MHz wall clock: 200 us
    const size_t bytes = 16 << 10;
    void* keep = nullptr;
    hipMalloc(&keep, bytes);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipStreamSynchronize(st);
            const int n = 200;
            const double t0 = now();
            for (int i = 0; i < n; ++i) {
                void* ws = keep;
                if (mode == 0) hipMallocAsync(&ws, bytes, st);
                if (mode <= 1) hipMemsetAsync(ws, 0, bytes, st);
                hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, st, cycles, (int*)ws);
                if (mode == 0) hipFreeAsync(ws, st);
            }
            const double host = now() - t0;
            hipStreamSynchronize(st);
            const double wall = now() - t0;
            if (rep == 1)
                printf("%s: host %.1f us per iteration, wall %.1f us\n",
                       mode == 0 ? "mallocAsync + memsetAsync + launch + freeAsync" : mode == 1 ? "persistent buffer: memsetAsync + launch" : "launch alone",
                       host / n * 1e6, wall / n * 1e6);
        }
    }
    return 0;
}
