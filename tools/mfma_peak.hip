// mfma_peak.hip — ceiling of v_mfma_f32_16x16x4_f32 on this chip for the accumulator pattern the conv
// kernel uses (20 independent accumulators, operands in VGPRs), at 1..3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float seed) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[4], b[4];
    for (int j = 0; j < 4; ++j) {
        if (seed > 0.f) { a[j] = seed + threadIdx.x * 1e-3f + j; b[j] = seed - threadIdx.x * 1e-3f - j; }
        else {   // random full-range operands (DVFS: data toggling costs clock)
            unsigned h = (threadIdx.x * 4u + j + blockIdx.x * 1024u) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            a[j] = ((h & 0xffff) / 32768.0f - 1.0f); b[j] = (((h >> 16) & 0xffff) / 32768.0f - 1.0f) * 0.05f;
        }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[(i + j) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *d; CK(hipMalloc(&d, 256 * 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
        const int blocks = 256 * wgs_per_cu;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_loop<20>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double flops = (double)blocks * 4 * iters * 80.0 * 2048.0;
            if (rep == 2) printf("NACC=20 waves/SIMD=%d: %.2f ms  %.1f TF\n", wgs_per_cu, ms, flops / (ms * 1e-3) / 1e12);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_loop<4>, dim3(512), dim3(256), 0, 0, d, iters * 5, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double flops = 512.0 * 4 * iters * 5 * 16.0 * 2048.0;
        if (rep == 2) printf("NACC=4 waves/SIMD=2: %.2f ms  %.1f TF\n", ms, flops / (ms * 1e-3) / 1e12);
    }
    // long run to see the sustained (power-limited) rate: ~0.5 s
    CK(hipEventRecord(e0));
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(mfma_loop<20>, dim3(512), dim3(256), 0, 0, d, iters * 2, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("sustained 10 launches: %.1f ms  %.1f TF\n", ms, 10.0 * 512 * 4 * iters * 2 * 80.0 * 2048.0 / (ms * 1e-3) / 1e12);
    CK(hipEventRecord(e0));
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(mfma_loop<20>, dim3(512), dim3(256), 0, 0, d, iters * 2, -1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("sustained, RANDOM operands: %.1f ms  %.1f TF\n", ms, 10.0 * 512 * 4 * iters * 2 * 80.0 * 2048.0 / (ms * 1e-3) / 1e12);
    return 0;
}
