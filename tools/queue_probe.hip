// queue_probe.hip — which HIP streams share a hardware queue?  A packet submitted to a stream after a long kernel
// was submitted to ANOTHER stream completes late iff both streams are multiplexed onto the same HW queue.
// Mirrors the engine's stream set-up (pocr_create): one set-up stream, then per slot a normal and a high-priority stream.
// usage: queue_probe [extra_streams_before] ; env GPU_MAX_HW_QUEUES can be varied by the caller.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin(long long cycles, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 1024) *sink = 1;
}
__global__ void tiny(int *p) { if (p && threadIdx.x == 1024) *p = 1; }

int main(int argc, char **argv) {
    const int extra = argc > 1 ? atoi(argv[1]) : 0;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    std::vector<hipStream_t> dummy(extra);
    for (auto &d : dummy) CK(hipStreamCreateWithFlags(&d, hipStreamNonBlocking));
    const char *names[5] = {"setup", "slot0.conv", "slot0.seq(hi)", "slot1.conv", "slot1.seq(hi)"};
    hipStream_t st[5];
    CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&st[2], hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&st[3], hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&st[4], hipStreamNonBlocking, hi));
    int *dbuf; int *hbuf;
    CK(hipMalloc(&dbuf, 64)); CK(hipHostMalloc(&hbuf, 64));
    // wall_clock64 ticks at 100 MHz: 2,000,000 ticks = 20 ms
    for (int i = 0; i < 5; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[i], (int *)nullptr); }
    CK(hipDeviceSynchronize());
    printf("priority range lo=%d hi=%d, extra streams before: %d\n", lo, hi, extra);
    printf("latency (ms) of a tiny kernel / a 4-byte D2H copy on stream ROW submitted while a 20 ms kernel runs on stream COL\n%-14s", "");
    for (int j = 0; j < 5; ++j) printf(" %-13s", names[j]);
    printf("\n");
    for (int i = 0; i < 5; ++i) {
        printf("%-14s", names[i]);
        for (int j = 0; j < 5; ++j) {
            if (i == j) { printf(" %-13s", "-"); continue; }
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[j], 2000000LL, (int *)nullptr);
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[i], (int *)nullptr);
            CK(hipStreamSynchronize(st[i]));
            const auto t1 = std::chrono::steady_clock::now();
            CK(hipMemcpyAsync(hbuf, dbuf, 4, hipMemcpyDeviceToHost, st[i]));
            CK(hipStreamSynchronize(st[i]));
            const auto t2 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            printf(" %5.1f/%-6.1f ", std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
        printf("\n");
    }
    return 0;
}
