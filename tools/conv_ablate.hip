// conv_ablate.hip — where the time of the split-precision conv kernel goes: one layer's shipped configuration, built
// several times with parts of the main loop switched off (-DPOCR_BF16X3_DBG=n: 1 no A reads, 2 no weight loads, 4 no A
// staging, 8 no barrier; results are then wrong, only the time matters).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -DPOCR_BF16X3_DBG=<n> -o tools/bin/conv_ablate_<n> tools/conv_ablate.hip
// Run  : tools/bin/conv_ablate_<n> [layer 9|8|6|4|2] [n_lines=256] [w_pad=576]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pero_ocr_amd/csrc/conv_igemm.hpp"
#include "../pero_ocr_amd/csrc/conv_bf16x3.hpp"
using namespace pocr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
template <class K>
static void launch(K kern, int TH, int TW, int NT, ConvArgs a, hipStream_t st) {
    a.tiles_w = (a.Wo + TW - 1) / TW; a.tiles_h = (a.Ho + TH - 1) / TH; a.tiles_n = (a.cout16 * 16) / NT;
    hipLaunchKernelGGL(kern, dim3((unsigned)conv_grid_blocks(a)), dim3(256), 0, st, a);
}
struct Shape { int cin, cout, H, W, ph, pw; };
struct Variant { const char *name; void (*fn)(ConvArgs, hipStream_t); int split; };
#define V(NAME, TH, MW, NS, WM, PH, PW, ACT, BN, MINW, BDIR, SPL) static void NAME(ConvArgs a, hipStream_t st) { \
    launch(conv3x3_bf16x3_kernel<TH, MW, NS, WM, PH, PW, ACT, BN, MINW, BDIR, 3, 3, 1, 1, false, SPL>, TH, 16 * MW, NS * (4 / WM) * 16, a, st); }
#define VP(NAME, TH, MW, NS, WM, PH, PW, ACT, BN, MINW, BDIR) static void NAME(ConvArgs a, hipStream_t st) { \
    launch(conv3x3_bf16x3_kernel<TH, MW, NS, WM, PH, PW, ACT, BN, MINW, BDIR, 3, 3, 1, 1, false, 2, true, true>, TH, 16 * MW, NS * (4 / WM) * 16, a, st); }
VP(p9, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2, true)
VP(p6, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2, true)
VP(p4, 4, 1, 2, 1, 2, 2, ACT_RELU, false, 2, true)
VP(p3, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2, true)
V(h3, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2, true, 2)
V(b9, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2, true, 3)
V(h9, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2, true, 2)
V(h8, 5, 1, 2, 1, 1, 1, ACT_LEAKY, false, 2, true, 2)
V(h6, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2, true, 2)
V(h4, 4, 1, 2, 1, 2, 2, ACT_RELU, false, 2, true, 2)
V(h2, 4, 2, 2, 2, 2, 2, ACT_RELU, false, 3, false, 2)
int main(int argc, char **argv) {
    const int layer = argc > 1 ? atoi(argv[1]) : 9, n = argc > 2 ? atoi(argv[2]) : 256, wpad = argc > 3 ? atoi(argv[3]) : 576;
    Shape s; std::vector<Variant> vars;
    if (layer == 9) { s = {512, 512, 5, wpad / 4, 1, 1}; vars = {{"bf16x3 5x16 NT128", b9, 3}, {"f16x2 5x16 NT128", h9, 2}, {"f16x2 P2 5x16 NT128", p9, 2}}; }
    else if (layer == 8) { s = {256, 512, 5, wpad / 4, 1, 1}; vars = {{"f16x2 5x16 NT128", h8, 2}}; }
    else if (layer == 6) { s = {256, 256, 10, wpad / 4, 1, 1}; vars = {{"f16x2 5x16 NT128", h6, 2}, {"f16x2 P2 5x16 NT128", p6, 2}}; }
    else if (layer == 4) { s = {128, 128, 20, wpad / 2, 2, 2}; vars = {{"f16x2 4x16 NT128", h4, 2}, {"f16x2 P2 4x16 NT128", p4, 2}}; }
    else if (layer == 3) { s = {64, 128, 20, wpad / 2, 1, 1}; vars = {{"f16x2 5x16 NT128", h3, 2}, {"f16x2 P2 5x16 NT128", p3, 2}}; }
    else { s = {64, 64, 40, wpad, 2, 2}; vars = {{"f16x2 lds 4x32 NT64 3WG", h2, 2}}; }
    const size_t xin = (size_t)n * s.H * s.W * s.cin, yout = (size_t)n * (s.H / s.ph) * (s.W / s.pw) * s.cout;
    std::vector<float> hx(xin);
    unsigned r = 12345;
    for (auto &v : hx) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xffff) / 32768.0f - 1.0f; }
    std::vector<uint16_t> hw((size_t)9 * s.cin * 512 * 3);
    for (auto &v : hw) { r = r * 1664525u + 1013904223u; v = (uint16_t)(0x2c00 + ((r >> 9) & 0x3ff)) | (uint16_t)((r >> 3) & 0x8000); }   // random small f16 / bf16 bit patterns
    float *dx, *dw, *db, *dy;
    CK(hipMalloc(&dx, xin * 4)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&db, 4096 * 4)); CK(hipMalloc(&dy, yout * 4));
    CK(hipMemcpy(dx, hx.data(), xin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(db, 0, 4096 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * n * s.H * s.W * (double)s.cout * s.cin * 9;
    printf("DBG=%d conv%d %d->%d @%dx%d n=%d\n", POCR_BF16X3_DBG, layer, s.cin, s.cout, s.H, s.W, n);
    for (auto &v : vars) {
        ConvArgs a{};
        a.x = dx; a.wfrag = dw; a.bias = db; a.bn_scale = db; a.bn_shift = db; a.y = dy;
        a.n = n; a.H = s.H; a.W = s.W; a.Ho = s.H; a.Wo = s.W; a.cin = s.cin; a.cout16 = s.cout / 16; a.cout_valid = s.cout; a.out_stride = s.cout;
        for (int w = 0; w < 3; ++w) v.fn(a, st);
        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        float sum = 0, best = 1e30f;
        for (int rep = 0; rep < 10; ++rep) {
            CK(hipEventRecord(e0, st)); v.fn(a, st); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; best = ms < best ? ms : best;
        }
        printf("  %-28s avg %.3f ms best %.3f ms  %.1f TF(alg)\n", v.name, sum / 10, best, flops / (sum / 10 * 1e-3) / 1e12);
    }
    return 0;
}
