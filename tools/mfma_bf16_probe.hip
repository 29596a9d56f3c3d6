// mfma_bf16_probe.hip — how exactly does v_mfma_f32_16x16x32_bf16 sum its 32 products?  One wave, A (16x32) and B (32x16) of
// exactly representable bf16 values at a chosen magnitude, C = 0 or a large value; result vs a float64 reference.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned short *A, const unsigned short *B, const float *C, float *D, int reps) {
    const int lane = threadIdx.x, li = lane & 15, kq = lane >> 4;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(4 * kq + r) * 16 + li];
    for (int it = 0; it < reps; ++it) {
        u32x4 a, b;
        for (int j = 0; j < 4; ++j) {
            const unsigned short *ap = A + ((size_t)it * 16 + li) * 32 + 8 * kq + 2 * j;      // A[it][row li][k]
            const unsigned short *bp = B + ((size_t)it * 16 + li) * 32 + 8 * kq + 2 * j;      // B[it][col li][k]
            a[j] = ap[0] | ((unsigned)ap[1] << 16);
            b[j] = bp[0] | ((unsigned)bp[1] << 16);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * kq + r) * 16 + li] = acc[r];
}
static float bf(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int reps = 72;
    unsigned r = 777;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (int mode = 0; mode < 4; ++mode) {
        const float bscale = mode == 0 ? 0.02f : 0.02f / 256;            // "hi" magnitude vs "mid" magnitude
        const float c0 = mode >= 2 ? 0.3f : 0.f;
        const bool mixed = mode == 3;                                    // B magnitudes vary by 2^-8 inside one MFMA
        std::vector<unsigned short> A(reps * 16 * 32), B(reps * 16 * 32);
        for (auto &v : A) { float x = rnd(); unsigned u; memcpy(&u, &x, 4); v = u >> 16; }
        for (size_t i = 0; i < B.size(); ++i) { float x = rnd() * bscale * ((mixed && (i & 1)) ? 256.f : 1.f); unsigned u; memcpy(&u, &x, 4); B[i] = u >> 16; }
        std::vector<float> C(256, c0), D(256);
        unsigned short *dA, *dB; float *dC, *dD;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, reps);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        double maxerr = 0, maxval = 0;
        for (int row = 0; row < 16; ++row) for (int col = 0; col < 16; ++col) {
            double ref = c0;
            for (int it = 0; it < reps; ++it) for (int k = 0; k < 32; ++k)
                ref += (double)bf(A[((size_t)it * 16 + row) * 32 + k]) * (double)bf(B[((size_t)it * 16 + col) * 32 + k]);
            maxerr = fmax(maxerr, fabs(ref - D[row * 16 + col])); maxval = fmax(maxval, fabs(ref - c0));
        }
        printf("mode %d (B scale %.2e, C0 %.1f%s): max |sum| %.3e  max err %.3e  (rel to sum %.2e)\n", mode, bscale, c0, mixed ? ", mixed magnitudes" : "", maxval, maxerr, maxerr / maxval);
    }
    return 0;
}
