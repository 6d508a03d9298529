// Host time of a kernel launch against the size of its by-value argument block (the coupled launches pass 0.6 - 3.1 KB):
// an empty kernel, 2000 launches back to back per size.   hipcc --offload-arch=gfx950 -O2 -o kernarg_cost_probe.bin kernarg_cost_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

template <int N>
struct Args {
    int v[N];
};
template <int N>
__global__ void touch(const Args<N> a, int* sink) {
    if (a.v[0] == 12345 && sink) *sink = a.v[N - 1];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int N>
void run(hipStream_t st) {
    Args<N> a = {};
    for (int rep = 0; rep < 2; ++rep) {
        hipStreamSynchronize(st);
        const int n = 2000;
        const double t0 = now();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(touch<N>, dim3(8), dim3(256), 0, st, a, (int*)nullptr);
        const double host = now() - t0;
        hipStreamSynchronize(st);
        const double wall = now() - t0;
        if (rep == 1) printf("%5d bytes of arguments: host %.2f us per launch, wall %.2f us\n", (int)sizeof(a), host / n * 1e6, wall / n * 1e6);
    }
}

int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    run<16>(st);
    run<144>(st);
    run<192>(st);
    run<336>(st);
    run<768>(st);
    run<960>(st);
    return 0;
}
