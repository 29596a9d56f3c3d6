// Sustained (power-limited) rate of the two f16 MFMA shapes of gfx950 with random operands: v_mfma_f32_16x16x32_f16 (the shape
// every kernel of the library uses: 2 KB of operands per 16 k FLOP) against v_mfma_f32_32x32x16_f16 (2 KB per 32 k FLOP:
// half the register-file reads per FLOP).  Same accumulator registers (96 per lane), 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const u32x4 *in, float *out, int iters) {
    u32x4 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = in[(threadIdx.x + 64 * i) & 1023];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = in[(threadIdx.x + 64 * i + 512) & 1023];
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 24; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i & 7]), __builtin_bit_cast(f16x8, b[(i >> 3) & 3]), acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 6; ++i)     // the same 2 x 16 k-deep FLOPs per accumulator register as above
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(i + 4 * j) & 7]), __builtin_bit_cast(f16x8, b[(i >> 1) & 3]), acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    std::vector<unsigned> h(4096);
    u32x4 *din; float *dout;
    CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&dout, 4096 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (auto &v : h) v = mode ? ((unsigned)(rand() & 0x3ff) << 16 | (unsigned)(rand() & 0x3ff) | 0x38003800u | ((unsigned)(rand() & 1) << 31) | ((unsigned)(rand() & 1) << 15)) : 0u;
        CK(hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice));
        for (int wgs = 1; wgs <= 2; ++wgs)
            for (int shape = 16; shape <= 32; shape += 16) {
                const int iters = 12000, blocks = 256 * wgs;
                float best = 1e9f, total = 0;
                for (int rep = 0; rep < 12; ++rep) {
                    CK(hipEventRecord(e0));
                    if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
                    else hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep >= 2) { total += ms; if (ms < best) best = ms; }
                }
                const double flops = (double)blocks * 4 * iters * 24 * 16384.0;
                printf("%s operands, %d wave(s)/SIMD, %s: avg %.2f ms = %.0f TF, best %.0f TF\n", mode ? "random" : "zero", wgs,
                       shape == 16 ? "16x16x32_f16" : "32x32x16_f16", total / 10, flops / (total / 10 * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12);
            }
    }
    return 0;
}
