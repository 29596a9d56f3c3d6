// hbm_write_probe.hip — what a pure store stream reaches on this part: fill kernels with different per-lane widths and
// per-wave contiguity (conv1 / the pooled P2 epilogues write 8-byte and 2-byte pieces), next to hipMemsetAsync.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/hbm_write_probe tools/hbm_write_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void fill16(f32x4 *p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (f32x4){1.f, 2.f, 3.f, 4.f}; }
__global__ void fill8(f32x2 *p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (f32x2){1.f, 2.f}; }
// conv1-like: lane (li & 3) -> pixel, (li >> 2) -> 8-byte piece inside a 32-byte run; four lanes cover 32 contiguous bytes of a 256-byte pixel
__global__ void fill8_scatter(f32x2 *p, size_t npix) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t px0 = ((size_t)blockIdx.x * 4) * 16; px0 < npix; px0 += (size_t)gridDim.x * 64) {
        const size_t px = px0 + (lane >> 4) * 4 + (lane & 3);
        if (px >= npix) continue;
        for (int piece = 0; piece < 2; ++piece)                   // h plane, l plane of one 32-channel chunk
            p[px * 32 + (wave >> 1) * 16 + piece * 8 + (wave & 1) * 4 + ((lane >> 2) & 3)] = (f32x2){1.f, 2.f};
    }
}
int main() {
    const size_t bytes = (size_t)1536 << 20;
    void *d; CK(hipMalloc(&d, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto fn) {
        fn(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        printf("%-28s %.3f ms  %.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
    };
    run("hipMemsetAsync", [&] { hipMemsetAsync(d, 1, bytes, 0); });
    run("16 B per lane, contiguous", [&] { hipLaunchKernelGGL(fill16, dim3(256 * 16), dim3(256), 0, 0, (f32x4 *)d, bytes / 16); });
    run("8 B per lane, contiguous", [&] { hipLaunchKernelGGL(fill8, dim3(256 * 16), dim3(256), 0, 0, (f32x2 *)d, bytes / 8); });
    run("8 B pieces, conv1-like", [&] { hipLaunchKernelGGL(fill8_scatter, dim3(256 * 16), dim3(256), 0, 0, (f32x2 *)d, bytes / 256); });
    return 0;
}
