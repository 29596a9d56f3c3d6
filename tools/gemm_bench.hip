// gemm_bench.hip — the persistent LDS-DMA GEMM (pero_ocr_amd/csrc/gemm_f16x2.hpp) against the kernel it replaces
// (conv3x3_bf16x3_kernel in GEMM mode / as the aggregation conv, P2 input) and against float64 on sampled outputs.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -o tools/bin/gemm_bench tools/gemm_bench.hip
// Run  : tools/bin/gemm_bench [M=36864] [K=512] [N=2048]      plain GEMM, fp32 and P2 output
//        tools/bin/gemm_bench agg [lines=256] [T=144] [AH=5]   the aggregation conv (gathered rows), 512 -> 512
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../pero_ocr_amd/csrc/conv_igemm.hpp"
#include "../pero_ocr_amd/csrc/conv_bf16x3.hpp"
#include "../pero_ocr_amd/csrc/gemm_f16x2.hpp"
using namespace pocr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned g_r = 12345;
static float rnd() { g_r = g_r * 1664525u + 1013904223u; return ((g_r >> 8) & 0xffff) / 32768.0f - 1.0f; }
static void split2(float v, uint16_t &h, uint16_t &l) {
    const _Float16 hh = (_Float16)v, ll = (_Float16)((v - (float)hh) * 2048.0f);
    memcpy(&h, &hh, 2); memcpy(&l, &ll, 2);
}
// fp32 [rows][C] -> P2
static std::vector<uint16_t> to_p2(const std::vector<float> &x, size_t rows, int C) {
    std::vector<uint16_t> o(rows * C * 2);
    for (size_t r = 0; r < rows; ++r)
        for (int c = 0; c < C; ++c) split2(x[r * C + c], o[r * C * 2 + (c >> 5) * 64 + (c & 31)], o[r * C * 2 + (c >> 5) * 64 + 32 + (c & 31)]);
    return o;
}
// W[co][ci][tap] -> wsplit[tap][ci/32][cout16][plane][lane][8]
static std::vector<uint16_t> wsplit(const std::vector<float> &W, int ntap, int cin, int cout) {
    const int c16 = cout / 16;
    std::vector<uint16_t> o((size_t)ntap * (cin / 32) * c16 * 2 * 64 * 8);
    size_t p = 0;
    for (int tap = 0; tap < ntap; ++tap)
        for (int g = 0; g < cin / 32; ++g)
            for (int s = 0; s < c16; ++s)
                for (int pl = 0; pl < 2; ++pl)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = 16 * s + (lane & 15), ci = 32 * g + 8 * (lane >> 4) + j;
                            uint16_t h, l;
                            split2(W[((size_t)co * cin + ci) * ntap + tap], h, l);
                            o[p++] = pl ? l : h;
                        }
    return o;
}
template <class F> static float timeit(F f, hipStream_t st, int reps, float *best) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) f();
    CK(hipStreamSynchronize(st)); CK(hipGetLastError());
    float sum = 0; *best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st)); f(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; if (ms < *best) *best = ms;
    }
    return sum / reps;
}
template <class K> static void launch_old(K kern, int TW, int NT, ConvArgs a, hipStream_t st) {
    a.tiles_w = (a.Wo + TW - 1) / TW; a.tiles_h = a.Ho; a.tiles_n = (a.cout16 * 16) / NT;
    hipLaunchKernelGGL(kern, dim3((unsigned)conv_grid_blocks(a)), dim3(256), 0, st, a);
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const bool agg = argc > 1 && !strcmp(argv[1], "agg");
    if (!agg) {
        const int M = argc > 1 ? atoi(argv[1]) : 36864, K = argc > 2 ? atoi(argv[2]) : 512, N = argc > 3 ? atoi(argv[3]) : 2048;
        std::vector<float> A((size_t)M * K), W((size_t)N * K), bias(N);
        for (auto &v : A) v = rnd();
        for (auto &v : W) v = rnd() * 0.05f;
        for (auto &v : bias) v = rnd();
        const int pad = argc > 5 ? atoi(argv[5]) : 0;                 // extra bytes per row of A (L2 channel spreading)
        auto Ap0 = to_p2(A, M, K); auto Ws = wsplit(W, 1, K, N);
        const size_t lda = (size_t)K * 4 + pad;
        std::vector<uint16_t> Ap((size_t)M * lda / 2, 0);
        for (int m = 0; m < M; ++m) memcpy(&Ap[(size_t)m * lda / 2], &Ap0[(size_t)m * K * 2], (size_t)K * 4);
        void *dA, *dW; float *dB, *dY0, *dY1, *dY2, *dY3;
        CK(hipMalloc(&dA, Ap.size() * 2)); CK(hipMalloc(&dW, Ws.size() * 2)); CK(hipMalloc(&dB, N * 4));
        const size_t ybytes = (size_t)M * N * 4;
        CK(hipMalloc(&dY0, ybytes)); CK(hipMalloc(&dY1, ybytes)); CK(hipMalloc(&dY2, ybytes)); CK(hipMalloc(&dY3, ybytes));
        CK(hipMemcpy(dA, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, Ws.data(), Ws.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, bias.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dY0, 0xff, ybytes)); CK(hipMemset(dY1, 0xff, ybytes)); CK(hipMemset(dY2, 0xff, ybytes)); CK(hipMemset(dY3, 0xff, ybytes));
        ConvArgs c{};
        c.x = (const float *)dA; c.wfrag = (const float *)dW; c.bias = dB; c.n = 1; c.H = 1; c.W = M; c.Ho = 1; c.Wo = M; c.cin = K;
        c.cout16 = N / 16; c.cout_valid = N; c.out_stride = N;
        GemmP2Args g{};
        g.a = dA; g.w = dW; g.bias = dB; g.M = M; g.nk = K / 32; g.N16 = N / 16; g.n_valid = N; g.ldy = N;
        g.mt_total = (M + kGemmBM - 1) / kGemmBM; g.nt_total = N / kGemmBN;
        g.nb = argc > 4 ? atoi(argv[4]) : g.nt_total; g.lda = (int64_t)lda;
        unsigned *dflag; CK(hipMalloc(&dflag, 8)); CK(hipMemset(dflag, 0, 8));
        g.range_flag = dflag;
        const int grid = gemm_f16x2_grid(M, N, cus);
        const double flops = 2.0 * M * (double)N * K;
        printf("GEMM %d x %d x %d (M K N), %d CUs, grid %d, nb %d, row pad %d B\n", M, K, N, cus, grid, g.nb, pad);
        float best;
        const bool run_old = pad == 0;
        auto old_f32 = [&] { if (!run_old) return; ConvArgs a = c; a.y = dY0; launch_old(conv3x3_bf16x3_kernel<1, 8, 4, 2, 1, 1, ACT_NONE, false, 2, false, 1, 1, 0, 0, false, 2, true, false>, 128, 128, a, st); };
        auto old_p2 = [&] { if (!run_old) return; ConvArgs a = c; a.y = dY2; launch_old(conv3x3_bf16x3_kernel<1, 8, 4, 2, 1, 1, ACT_RELU, false, 2, false, 1, 1, 0, 0, false, 2, true, true>, 128, 128, a, st); };
        auto new_f32 = [&] { GemmP2Args a = g; a.y = dY1; hipLaunchKernelGGL((gemm_f16x2_kernel<ACT_NONE, false, false>), dim3(grid), dim3(kGemmThreads), 0, st, a); };
        auto new_p2 = [&] { GemmP2Args a = g; a.y = dY3; hipLaunchKernelGGL((gemm_f16x2_kernel<ACT_RELU, true, false>), dim3(grid), dim3(kGemmThreads), 0, st, a); };
        float ms = timeit(old_f32, st, 10, &best);
        printf("  old 128x128 P2 in -> fp32      avg %.3f ms best %.3f ms  %.1f TF(alg)\n", ms, best, flops / (ms * 1e-3) / 1e12);
        ms = timeit(new_f32, st, 10, &best);
        printf("  new 256x128 persistent -> fp32 avg %.3f ms best %.3f ms  %.1f TF(alg)\n", ms, best, flops / (ms * 1e-3) / 1e12);
        ms = timeit(old_p2, st, 10, &best);
        printf("  old 128x128 P2 in -> P2 (relu) avg %.3f ms best %.3f ms  %.1f TF(alg)\n", ms, best, flops / (ms * 1e-3) / 1e12);
        ms = timeit(new_p2, st, 10, &best);
        printf("  new 256x128 persistent -> P2   avg %.3f ms best %.3f ms  %.1f TF(alg)\n", ms, best, flops / (ms * 1e-3) / 1e12);
        std::vector<float> y0((size_t)M * N), y1((size_t)M * N);
        CK(hipMemcpy(y0.data(), dY0, ybytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dY1, ybytes, hipMemcpyDeviceToHost));
        size_t diff = 0; for (size_t i = 0; i < y0.size(); ++i) diff += memcmp(&y0[i], &y1[i], 4) != 0;
        printf("  fp32 outputs: %zu of %zu differ bitwise from the old kernel\n", diff, y0.size());
        CK(hipMemcpy(y0.data(), dY2, ybytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dY3, ybytes, hipMemcpyDeviceToHost));
        diff = 0; for (size_t i = 0; i < y0.size(); ++i) diff += memcmp(&y0[i], &y1[i], 4) != 0;
        printf("  P2 outputs:   %zu of %zu words differ bitwise from the old kernel\n", diff, y0.size());
        CK(hipMemcpy(y1.data(), dY1, ybytes, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int s = 0; s < 2000; ++s) {
            g_r = g_r * 1664525u + 1013904223u; const size_t m = (s < 8 ? M - 1 - s : g_r % M);
            g_r = g_r * 1664525u + 1013904223u; const size_t n = g_r % N;
            double acc = bias[n];
            for (int k = 0; k < K; ++k) acc += (double)A[m * K + k] * (double)W[n * K + k];
            worst = std::max(worst, std::fabs(acc - (double)y1[m * N + n]));
        }
        unsigned flag[2]; CK(hipMemcpy(flag, dflag, 8, hipMemcpyDeviceToHost));
        float fm; memcpy(&fm, &flag[0], 4);
        printf("  new kernel vs float64 on 2000 sampled outputs: max |err| %.3e;  range flag max |y| = %g\n", worst, fm);
        return 0;
    }
    // ---- aggregation conv: lines x [AH][T][512] (P2) -> [lines * T][512], LeakyReLU
    const int n = argc > 2 ? atoi(argv[2]) : 256, T = argc > 3 ? atoi(argv[3]) : 144, AH = argc > 4 ? atoi(argv[4]) : 5, C = 512, E = 512;
    const size_t rows = (size_t)n * T;
    std::vector<float> X((size_t)n * AH * T * C), W((size_t)E * C * AH), bias(E);
    for (auto &v : X) v = rnd();
    for (auto &v : W) v = rnd() * 0.03f;
    for (auto &v : bias) v = rnd();
    auto Xp = to_p2(X, (size_t)n * AH * T, C); auto Ws = wsplit(W, AH, C, E);
    void *dX, *dW; float *dB, *dY0, *dY1;
    CK(hipMalloc(&dX, Xp.size() * 2)); CK(hipMalloc(&dW, Ws.size() * 2)); CK(hipMalloc(&dB, E * 4));
    CK(hipMalloc(&dY0, rows * E * 4)); CK(hipMalloc(&dY1, rows * E * 4));
    CK(hipMemcpy(dX, Xp.data(), Xp.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, Ws.data(), Ws.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, bias.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dY0, 0xff, rows * E * 4)); CK(hipMemset(dY1, 0xff, rows * E * 4));
    std::vector<int32_t> row_line(rows), row_t(rows), line_w(n, T); std::vector<int64_t> in_off(n + 1);
    for (int i = 0; i < n; ++i) { in_off[i] = (int64_t)i * AH * T * C; for (int t = 0; t < T; ++t) { row_line[(size_t)i * T + t] = i; row_t[(size_t)i * T + t] = t; } }
    int32_t *dRL, *dRT, *dLW; int64_t *dIO;
    CK(hipMalloc(&dRL, rows * 4)); CK(hipMalloc(&dRT, rows * 4)); CK(hipMalloc(&dLW, n * 4)); CK(hipMalloc(&dIO, (n + 1) * 8));
    CK(hipMemcpy(dRL, row_line.data(), rows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dRT, row_t.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dLW, line_w.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dIO, in_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    ConvArgs c{};
    c.x = (const float *)dX; c.wfrag = (const float *)dW; c.bias = dB; c.y = dY0; c.n = n; c.H = AH; c.W = T; c.Ho = 1; c.Wo = T; c.cin = C;
    c.cout16 = E / 16; c.cout_valid = E; c.out_stride = E;
    GemmP2Args g{};
    g.a = dX; g.w = dW; g.bias = dB; g.y = dY1; g.M = (int)rows; g.nk = AH * (C / 32); g.N16 = E / 16; g.n_valid = E; g.ldy = E;
    g.mt_total = (int)((rows + kGemmBM - 1) / kGemmBM); g.nt_total = E / kGemmBN; g.nb = g.nt_total;
    g.row_line = dRL; g.row_t = dRT; g.line_w = dLW; g.in_off = dIO; g.cpt = C / 32; g.ntap = AH; g.cin = C;
    const int grid = gemm_f16x2_grid((int)rows, E, cus);
    const double flops = 2.0 * rows * (double)E * C * AH;
    printf("aggregation conv %d lines x [%d][%d][%d] -> %d, grid %d\n", n, AH, T, C, E, grid);
    float best, ms;
    auto old_k = [&] {
        if (AH == 5) launch_old(conv3x3_bf16x3_kernel<1, 3, 2, 1, 1, 1, ACT_LEAKY, false, 2, true, 5, 1, 0, 0, false, 2, true, false>, 48, 128, c, st);
        else launch_old(conv3x3_bf16x3_kernel<1, 3, 2, 1, 1, 1, ACT_LEAKY, false, 2, true, 4, 1, 0, 0, false, 2, true, false>, 48, 128, c, st);
    };
    auto new_k = [&] { hipLaunchKernelGGL((gemm_f16x2_kernel<ACT_LEAKY, false, true>), dim3(grid), dim3(kGemmThreads), 0, st, g); };
    ms = timeit(old_k, st, 10, &best);
    printf("  old 48 x 128 direct weights    avg %.3f ms best %.3f ms  %.1f TF(alg)\n", ms, best, flops / (ms * 1e-3) / 1e12);
    ms = timeit(new_k, st, 10, &best);
    printf("  new 256x128 persistent gather  avg %.3f ms best %.3f ms  %.1f TF(alg)\n", ms, best, flops / (ms * 1e-3) / 1e12);
    std::vector<float> y0(rows * E), y1(rows * E);
    CK(hipMemcpy(y0.data(), dY0, rows * E * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dY1, rows * E * 4, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < y0.size(); ++i) diff += memcmp(&y0[i], &y1[i], 4) != 0;
    printf("  outputs: %zu of %zu differ bitwise from the old kernel\n", diff, y0.size());
    double worst = 0;
    for (int s = 0; s < 500; ++s) {
        g_r = g_r * 1664525u + 1013904223u; const size_t m = g_r % rows;
        g_r = g_r * 1664525u + 1013904223u; const size_t e = g_r % E;
        const size_t i = m / T, t = m % T;
        double acc = bias[e];
        for (int tap = 0; tap < AH; ++tap)
            for (int k = 0; k < C; ++k) acc += (double)X[((i * AH + tap) * T + t) * C + k] * (double)W[((size_t)e * C + k) * AH + tap];
        if (acc < 0) acc *= 0.01;
        worst = std::max(worst, std::fabs(acc - (double)y1[m * E + e]));
    }
    printf("  new kernel vs float64 on 500 sampled outputs: max |err| %.3e\n", worst);
    return 0;
}
