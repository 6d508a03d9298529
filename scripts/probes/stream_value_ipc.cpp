// Feasibility probe for a stream-ordered direct exchange: can process B write into process A's
// device buffer through a hipIpc mapping and signal A with hipStreamWriteValue32, while A's STREAM (not a CU-resident
// spin loop) waits with hipStreamWaitValue32?  Two processes on the ONE GPU of a gpurun box; every wait is bounded by the
// caller's `timeout`.  Build: hipcc --offload-arch=gfx950 -O2 -o stream_value_ipc stream_value_ipc.cpp
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                                        \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            printf("[%s] %s failed: %s\n", who, #x, hipGetErrorString(e_));                          \
            fflush(stdout);                                                                          \
            _exit(3);                                                                                \
        }                                                                                            \
    } while (0)

__global__ void fill(float* p, int n, float base) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = base + (float)i;
}
__global__ void copy(const float* a, float* b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

struct Handles {
    hipIpcMemHandle_t data, flag;
    int flag_is_signal;
};

// Phase 2, ping-pong: per round A runs a small kernel, writes t into B's word and waits for t in its own; B waits, runs a
// small kernel, writes t into A's word.  Everything is enqueued ahead on both streams: the round-trip time is what two
// stream-ordered hand-offs (write value -> the other stream's wait value -> dependent kernel) cost.
static const int kRounds = 500;

int main() {
    const int n = 1 << 16, steps = 64;
    int a2b[2], b2a[2];
    if (pipe(a2b) || pipe(b2a)) return 2;
    pid_t child = fork();  // before any HIP call in either process
    const char* who = child ? "A" : "B";
    if (child) {
        // ---- A: owns the buffer and the signal word, waits on its stream ----
        int can = 0;
        CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
        printf("[A] hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
        float *data = nullptr, *out = nullptr;
        uint32_t* flag = nullptr;
        CK(hipMalloc(&data, n * sizeof(float)));
        CK(hipMalloc(&out, n * sizeof(float)));
        Handles h;
        memset(&h, 0, sizeof h);
        h.flag_is_signal = 1;
        if (hipExtMallocWithFlags((void**)&flag, 4096, hipMallocSignalMemory) != hipSuccess) {
            (void)hipGetLastError();
            h.flag_is_signal = 0;
            CK(hipMalloc((void**)&flag, 4096));
        }
        CK(hipMemset(flag, 0, 4096));
        CK(hipMemset(data, 0, n * sizeof(float)));
        CK(hipIpcGetMemHandle(&h.data, data));
        hipError_t fe = hipIpcGetMemHandle(&h.flag, flag);
        if (fe != hipSuccess) {
            printf("[A] hipIpcGetMemHandle on %s memory failed: %s -> retry with plain hipMalloc\n",
                   h.flag_is_signal ? "signal" : "plain", hipGetErrorString(fe));
            (void)hipGetLastError();
            h.flag_is_signal = 0;
            CK(hipMalloc((void**)&flag, 4096));
            CK(hipMemset(flag, 0, 4096));
            CK(hipIpcGetMemHandle(&h.flag, flag));
        }
        CK(hipDeviceSynchronize());
        if (write(a2b[1], &h, sizeof h) != (ssize_t)sizeof h) return 2;
        hipStream_t st;
        CK(hipStreamCreate(&st));
        int bad = 0;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int t = 1; t <= steps; ++t) {  // the whole sequence is ENQUEUED ahead: nothing but the stream waits
            hipError_t we = hipStreamWaitValue32(st, flag, (uint32_t)t, hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (we != hipSuccess) {
                printf("[A] hipStreamWaitValue32 on %s memory: %s\n", h.flag_is_signal ? "signal" : "plain hipMalloc",
                       hipGetErrorString(we));
                _exit(4);
            }
        }
        hipLaunchKernelGGL(copy, dim3(n / 256), dim3(256), 0, st, data, out, n);
        CK(hipEventRecord(e1, st));
        char go = 1;
        if (write(a2b[1], &go, 1) != 1) return 2;  // B may start now
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        float* host = (float*)malloc(n * sizeof(float));
        CK(hipMemcpy(host, out, n * sizeof(float), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) bad += host[i] != (float)steps + (float)i;
        printf("[A] %d stream waits satisfied by B's stream writes (%s flag memory), %.1f us per step; payload %s\n", steps,
               h.flag_is_signal ? "hipMallocSignalMemory" : "plain hipMalloc", ms * 1e3 / steps, bad ? "WRONG" : "correct");
        // ---- phase 2 (A side) ----
        hipIpcMemHandle_t hb;
        if (read(b2a[0], &hb, sizeof hb) != (ssize_t)sizeof hb) return 2;
        uint32_t* flag_b = nullptr;
        CK(hipIpcOpenMemHandle((void**)&flag_b, hb, hipIpcMemLazyEnablePeerAccess));
        CK(hipMemset(flag, 0, 4096));
        CK(hipDeviceSynchronize());
        if (write(a2b[1], &go, 1) != 1) return 2;
        CK(hipEventRecord(e0, st));
        for (int t = 1; t <= kRounds; ++t) {
            hipLaunchKernelGGL(fill, dim3(1), dim3(256), 0, st, out, 256, (float)t);
            CK(hipStreamWriteValue32(st, flag_b, (uint32_t)t, 0));
            CK(hipStreamWaitValue32(st, flag, (uint32_t)t, hipStreamWaitValueGte, 0xFFFFFFFFu));
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("[A] ping-pong: %.2f us per round trip (two hand-offs + two 1-block kernels), %d rounds\n", ms * 1e3 / kRounds,
               kRounds);
        CK(hipIpcCloseMemHandle(flag_b));
        int status = 0;
        waitpid(child, &status, 0);
        printf("[A] child exit %d; RESULT %s\n", WEXITSTATUS(status), (!bad && WEXITSTATUS(status) == 0) ? "OK" : "FAIL");
        return bad || WEXITSTATUS(status);
    }
    // ---- B: maps A's buffer and signal word, writes the payload with a kernel, then signals from the stream ----
    Handles h;
    if (read(a2b[0], &h, sizeof h) != (ssize_t)sizeof h) _exit(2);
    float* data = nullptr;
    uint32_t* flag = nullptr;
    CK(hipIpcOpenMemHandle((void**)&data, h.data, hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle((void**)&flag, h.flag, hipIpcMemLazyEnablePeerAccess));
    char go;
    if (read(a2b[0], &go, 1) != 1) _exit(2);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int t = 1; t <= steps; ++t) {
        hipLaunchKernelGGL(fill, dim3(n / 256), dim3(256), 0, st, data, n, (float)t);
        CK(hipStreamWriteValue32(st, flag, (uint32_t)t, 0));
    }
    CK(hipStreamSynchronize(st));
    printf("[B] wrote %d payloads + signals through the IPC mappings\n", steps);
    // ---- phase 2 (B side) ----
    uint32_t* my_flag = nullptr;
    float* scratch = nullptr;
    CK(hipMalloc((void**)&my_flag, 4096));
    CK(hipMalloc(&scratch, 1024));
    CK(hipMemset(my_flag, 0, 4096));
    hipIpcMemHandle_t hb;
    CK(hipIpcGetMemHandle(&hb, my_flag));
    CK(hipDeviceSynchronize());
    if (write(b2a[1], &hb, sizeof hb) != (ssize_t)sizeof hb) _exit(2);
    if (read(a2b[0], &go, 1) != 1) _exit(2);
    for (int t = 1; t <= kRounds; ++t) {
        CK(hipStreamWaitValue32(st, my_flag, (uint32_t)t, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(fill, dim3(1), dim3(256), 0, st, scratch, 256, (float)t);
        CK(hipStreamWriteValue32(st, flag, (uint32_t)t, 0));
    }
    CK(hipStreamSynchronize(st));
    CK(hipIpcCloseMemHandle(data));
    CK(hipIpcCloseMemHandle(flag));
    fflush(stdout);
    _exit(0);
}
