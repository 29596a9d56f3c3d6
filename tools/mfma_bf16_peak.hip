// Sustained rate of v_mfma_f32_16x16x32_bf16 on gfx950: NACC independent accumulators per wave, W waves per SIMD,
// zero / random operands (the power-limited clock depends on how many bits toggle).  What "peak" can a bf16 kernel reach here?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int NACC>
__global__ __launch_bounds__(256) void k(const u32x4 *in, float *out, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x + 64 * i) & 1023]; b[i] = in[(threadIdx.x + 64 * i + 256) & 1023]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[(i >> 2) & 3]), acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    std::vector<unsigned> h(4096);
    u32x4 *din; float *dout;
    CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&dout, 4096 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (auto &v : h) v = mode ? ((unsigned)(rand() & 0x3fff) << 16 | (unsigned)(rand() & 0x3fff) | 0x3c003c00u) : 0u;   // bf16 values of magnitude ~1
        CK(hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice));
        for (int wgs = 1; wgs <= 2; ++wgs) {
            const int iters = 20000, blocks = 256 * wgs;
            float best = 1e9f, total = 0;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k<20>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) { total += ms; if (ms < best) best = ms; }
            }
            const double flops = (double)blocks * 4 * iters * 20 * 16384.0;
            printf("%s operands, %d wave(s)/SIMD, 20 accumulators: avg %.2f ms = %.0f TF, best %.0f TF (x3 split ceiling: %.1f TF algorithmic)\n",
                   mode ? "random" : "zero", wgs, total / 10, flops / (total / 10 * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12, flops / (total / 10 * 1e-3) / 1e12 / 6);
        }
    }
    return 0;
}
