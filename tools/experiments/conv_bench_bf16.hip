// PARKED (round 6): this tool drives the POCR_BF16X3_DBG ablations (no A reads / no weight loads / no staging / no barrier) of csrc/conv_bf16x3.hpp, which were compile-time switches of the
// library until round 6 removed them (the measurements they produced: profiles/r02_conv_bf16x3_bench.txt, r03_conv_f16x2_ablation.txt,
// r03_conv_tile_trace.txt, r05_conv_rows.txt).  It builds against the round-5 tree: `git worktree add /tmp/r05 4aef8f0` and compile there.
// conv_bench_bf16.hip — the bf16x3 conv kernel (csrc/conv_bf16x3.hpp) against the shipped fp32-MFMA kernel on one layer:
// time, max |difference| between the two, and both against a float64 CPU reference on sampled outputs.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -o /tmp/conv_bench_bf16 tools/conv_bench_bf16.hip
//        (~18 minutes: a few of the round-2 bf16x3 variants with issue-order templates send the scheduler's solver on long
//        searches; tools/conv_ablate.hip holds the round-3 candidates and builds in 20 s)
// Run  : /tmp/conv_bench_bf16 <layer 2..9> [n_lines=256] [w_pad=576]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../pero_ocr_amd/csrc/conv_igemm.hpp"
#include "../pero_ocr_amd/csrc/conv_bf16x3.hpp"
using namespace pocr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
struct Shape { int cin, cout, H, W, ph, pw, act; bool bn; };
template <class K>
static void launch(K kern, int TH, int TW, int NT, ConvArgs a, hipStream_t st) {
    a.tiles_w = (a.Wo + TW - 1) / TW; a.tiles_h = (a.Ho + TH - 1) / TH; a.tiles_n = (a.cout16 * 16) / NT;
    hipLaunchKernelGGL(kern, dim3((unsigned)conv_grid_blocks(a)), dim3(256), 0, st, a);
}
struct Variant { const char *name; void (*fn)(ConvArgs, hipStream_t); int nt; int bf; };     // bf: 0 fp32 MFMA, 1 / 3 bf16x3, 2 f16x2
#define VF32(NAME, TH, MW, NS, PH, PW, ACT, BN, PIPE) static void NAME(ConvArgs a, hipStream_t st) { \
    launch(conv_igemm_kernel<3, 3, 1, 1, TH, MW, NS, 4, 16, PH, PW, ACT, BN, STAGE_F32_NHWC, PIPE>, TH, 16 * MW, NS * 64, a, st); }
#define VBF(NAME, TH, MW, NS, WM, PH, PW, ACT, BN, MINW) static void NAME(ConvArgs a, hipStream_t st) { \
    launch(conv3x3_bf16x3_kernel<TH, MW, NS, WM, PH, PW, ACT, BN, MINW>, TH, 16 * MW, NS * (4 / WM) * 16, a, st); }
#define VBD(NAME, TH, MW, NS, WM, PH, PW, ACT, BN, MINW) static void NAME(ConvArgs a, hipStream_t st) { \
    launch(conv3x3_bf16x3_kernel<TH, MW, NS, WM, PH, PW, ACT, BN, MINW, true>, TH, 16 * MW, NS * (4 / WM) * 16, a, st); }
// f16x2 (two f16 planes, three MFMAs per product block): LDS weights / weights straight from L2
#define VH(NAME, TH, MW, NS, WM, PH, PW, ACT, BN, MINW) static void NAME(ConvArgs a, hipStream_t st) { \
    launch(conv3x3_bf16x3_kernel<TH, MW, NS, WM, PH, PW, ACT, BN, MINW, false, 3, 3, 1, 1, false, 2>, TH, 16 * MW, NS * (4 / WM) * 16, a, st); }
#define VHD(NAME, TH, MW, NS, WM, PH, PW, ACT, BN, MINW) static void NAME(ConvArgs a, hipStream_t st) { \
    launch(conv3x3_bf16x3_kernel<TH, MW, NS, WM, PH, PW, ACT, BN, MINW, true, 3, 3, 1, 1, false, 2>, TH, 16 * MW, NS * (4 / WM) * 16, a, st); }
VHD(h9_a, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2)      // as shipped bf16x3 conv9: 5x16, NT128, 2 WG/CU
VHD(h9_b, 5, 2, 2, 1, 1, 1, ACT_LEAKY, true, 2)      // 5x32: MS 10
VHD(h9_c, 5, 2, 2, 1, 1, 1, ACT_LEAKY, true, 1)
VHD(h9_d, 5, 1, 4, 1, 1, 1, ACT_LEAKY, true, 1)      // NT256, MS 5, NS 4
VHD(h9_f, 5, 3, 1, 1, 1, 1, ACT_LEAKY, true, 2)      // 5x48 (144 = 3 x 48), NS 1, NT 64: MS 15
VHD(h6_a, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2)
VHD(h6_b, 10, 1, 2, 1, 1, 1, ACT_RELU, false, 2)     // MS 10
VHD(h6_c, 5, 2, 2, 1, 1, 1, ACT_RELU, false, 2)
VHD(h8_a, 5, 1, 2, 1, 1, 1, ACT_LEAKY, false, 2)
VHD(h8_b, 5, 2, 2, 1, 1, 1, ACT_LEAKY, false, 2)
VHD(h7_a, 2, 2, 2, 1, 2, 1, ACT_RELU, false, 2)
VHD(h7_b, 10, 1, 2, 1, 2, 1, ACT_RELU, false, 2)
VHD(h7_c, 2, 4, 2, 1, 2, 1, ACT_RELU, false, 2)
VHD(h4_a, 4, 1, 2, 1, 2, 2, ACT_RELU, false, 2)
VHD(h4_b, 4, 2, 2, 1, 2, 2, ACT_RELU, false, 2)
VHD(h3_a, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2)
VHD(h3_b, 4, 2, 2, 1, 1, 1, ACT_RELU, false, 2)
VH(h2_a, 4, 2, 2, 2, 2, 2, ACT_RELU, false, 2)
VHD(h2_c, 4, 2, 2, 2, 2, 2, ACT_RELU, false, 2)
VHD(h2_d, 4, 4, 2, 2, 2, 2, ACT_RELU, false, 2)
VHD(h2_e, 8, 2, 2, 2, 2, 2, ACT_RELU, false, 2)
VHD(h2_f, 4, 4, 4, 4, 2, 2, ACT_RELU, false, 2)      // 4x64 px, all four waves share the 64 channels: NS 4, MS 4
VBD(d9_a, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2)      // direct weights, 5x16 px, NT128, 2 WG/CU
VBD(d9_b, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 3)      // 3 WG/CU
VBD(d9_c, 5, 1, 4, 1, 1, 1, ACT_LEAKY, true, 1)      // NT256
VBD(d9_d, 10, 1, 2, 1, 1, 1, ACT_LEAKY, true, 1)     // (only H=10 layers)
VBD(d9_e, 5, 2, 2, 1, 1, 1, ACT_LEAKY, true, 1)      // 5x32 px, N-split, NT128: MS 10
VBD(d6_a, 10, 1, 2, 1, 1, 1, ACT_RELU, false, 1)
VBD(d6_b, 10, 1, 2, 1, 1, 1, ACT_RELU, false, 2)
VBD(d6_c, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2)
VBD(d6_d, 5, 1, 4, 1, 1, 1, ACT_RELU, false, 1)
VBD(d8_a, 5, 1, 2, 1, 1, 1, ACT_LEAKY, false, 2)
VBD(d8_b, 5, 1, 2, 1, 1, 1, ACT_LEAKY, false, 3)
VBD(d7_a, 10, 1, 2, 1, 2, 1, ACT_RELU, false, 1)
VBD(d7_b, 10, 1, 2, 1, 2, 1, ACT_RELU, false, 2)
VBD(d4_a, 4, 2, 2, 1, 2, 2, ACT_RELU, false, 1)
VBD(d4_b, 4, 2, 2, 1, 2, 2, ACT_RELU, false, 2)
VBD(d3_a, 4, 2, 2, 1, 1, 1, ACT_RELU, false, 1)
VBD(d3_b, 4, 2, 2, 1, 1, 1, ACT_RELU, false, 2)
VBF(b2_d, 8, 2, 2, 2, 2, 2, ACT_RELU, false, 2)      // 8x32 px, NT64, 2 x 2
VBF(b2_e, 4, 4, 2, 2, 2, 2, ACT_RELU, false, 2)      // 4x64 px, NT64, 2 x 2 (MS 8)
VBD(d2_c, 8, 2, 2, 2, 2, 2, ACT_RELU, false, 2)
VBD(d2_d, 4, 4, 2, 2, 2, 2, ACT_RELU, false, 2)
VBF(b7_b, 10, 2, 2, 2, 2, 1, ACT_RELU, false, 1)     // 10x32 px, NT64?? (WN 2 x NS 2 = 64 channels)
VBF(b7_c, 6, 1, 2, 1, 2, 1, ACT_RELU, false, 2)      // 6x16 px: 2 tiles cover 10 rows (+20 % rows)
VBD(d7_c, 6, 1, 2, 1, 2, 1, ACT_RELU, false, 2)
VBD(d7_d, 2, 2, 2, 1, 2, 1, ACT_RELU, false, 2)      // 2x32 px, NT128: MS 4
VBF(b2_f, 4, 2, 2, 2, 2, 2, ACT_RELU, false, 2)      // 4x32 px, NT64, 2 x 2, MS 4: 2+ WG/CU
VBD(d2_f, 4, 2, 2, 2, 2, 2, ACT_RELU, false, 2)
VBF(b2_g, 4, 2, 2, 2, 2, 2, ACT_RELU, false, 3)
VBD(d3_c, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2)      // conv3 on 5x16 tiles (the conv5/6 configuration)
VBD(d3_d, 4, 1, 2, 1, 1, 1, ACT_RELU, false, 2)
VBD(d3_e, 4, 1, 2, 1, 1, 1, ACT_RELU, false, 3)
VBD(d4_c, 4, 1, 2, 1, 2, 2, ACT_RELU, false, 2)      // conv4 (pool 2x2) on 4x16 tiles
VBD(d4_d, 4, 1, 2, 1, 2, 2, ACT_RELU, false, 3)
VBD(d4_e, 2, 2, 2, 1, 2, 2, ACT_RELU, false, 2)      // 2x32
VBD(d2_a, 4, 4, 4, 4, 2, 2, ACT_RELU, false, 2)
VBD(d2_b, 4, 4, 1, 1, 2, 2, ACT_RELU, false, 2)
VF32(f9, 5, 1, 4, 1, 1, ACT_LEAKY, true, PIPE_DEEP)
VBF(b9_a, 5, 1, 4, 1, 1, 1, ACT_LEAKY, true, 1)      // 5x16 px, NT256
VBF(b9_b, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2)      // 5x16 px, NT128, 2 WG/CU
VBF(b9_c, 5, 3, 2, 1, 1, 1, ACT_LEAKY, true, 1)      // 5x48 px, NT128, waves split N
VBF(b9_d, 5, 2, 4, 2, 1, 1, ACT_LEAKY, true, 1)      // 5x32 px, NT128, waves 2 (M) x 2 (N)
VBF(b9_e, 5, 2, 4, 2, 1, 1, ACT_LEAKY, true, 2)      // same, 2 WG/CU
VBF(b9_f, 5, 4, 4, 4, 1, 1, ACT_LEAKY, true, 1)      // 5x64 px, NT64, waves split M
VBF(b9_g, 5, 4, 8, 4, 1, 1, ACT_LEAKY, true, 1)      // 5x64 px, NT128, waves split M
VBF(b9_h, 5, 2, 8, 2, 1, 1, ACT_LEAKY, true, 1)      // 5x32 px, NT256, waves 2 x 2
VF32(f6, 10, 1, 2, 1, 1, ACT_RELU, false, PIPE_DEEP)
VBF(b6_a, 10, 1, 2, 1, 1, 1, ACT_RELU, false, 1)
VBF(b6_b, 5, 1, 4, 1, 1, 1, ACT_RELU, false, 1)
VBF(b6_c, 10, 2, 2, 2, 1, 1, ACT_RELU, false, 1)     // 10x32 px, NT64, 2 x 2
VBF(b6_d, 10, 2, 4, 2, 1, 1, ACT_RELU, false, 1)     // 10x32 px, NT128, 2 x 2
VF32(f3, 4, 2, 2, 1, 1, ACT_RELU, false, PIPE_INTERLEAVED)
VBF(b3_a, 4, 2, 2, 1, 1, 1, ACT_RELU, false, 1)
VBF(b3_b, 4, 2, 2, 1, 1, 1, ACT_RELU, false, 2)
VBF(b3_c, 4, 4, 4, 2, 1, 1, ACT_RELU, false, 1)
VF32(f7, 10, 1, 2, 2, 1, ACT_RELU, false, PIPE_INTERLEAVED)
VBF(b7_a, 10, 1, 2, 1, 2, 1, ACT_RELU, false, 1)
VF32(f8, 5, 1, 4, 1, 1, ACT_LEAKY, false, PIPE_DEEP)
VBF(b8_a, 5, 1, 2, 1, 1, 1, ACT_LEAKY, false, 2)
VBF(b8_b, 5, 1, 4, 1, 1, 1, ACT_LEAKY, false, 1)
VF32(f5, 10, 1, 2, 1, 1, ACT_RELU, false, PIPE_DEEP)
VF32(f4, 4, 2, 2, 2, 2, ACT_RELU, false, PIPE_INTERLEAVED)
VBF(b4_a, 4, 2, 2, 1, 2, 2, ACT_RELU, false, 1)
VBF(b4_b, 4, 4, 4, 2, 2, 2, ACT_RELU, false, 1)      // 4x64 px, NT128, 2 x 2
VBF(b4_c, 4, 4, 2, 2, 2, 2, ACT_RELU, false, 2)      // 4x64 px, NT64, 2 x 2, 2 WG/CU
VF32(f2, 4, 4, 1, 2, 2, ACT_RELU, false, PIPE_DEEP)
VBF(b2_a, 4, 4, 1, 1, 2, 2, ACT_RELU, false, 1)
VBF(b2_b, 4, 4, 4, 4, 2, 2, ACT_RELU, false, 2)      // 4x64 px, NT64, waves split M
VBF(b2_c, 8, 4, 4, 4, 2, 2, ACT_RELU, false, 1)      // 8x64 px, NT64, waves split M

int main(int argc, char **argv) {
    int layer = argc > 1 ? atoi(argv[1]) : 9, n = argc > 2 ? atoi(argv[2]) : 256, wpad = argc > 3 ? atoi(argv[3]) : 576;
    Shape shapes[10] = {{}, {}, {64, 64, 40, wpad, 2, 2, ACT_RELU, false}, {64, 128, 20, wpad / 2, 1, 1, ACT_RELU, false},
                        {128, 128, 20, wpad / 2, 2, 2, ACT_RELU, false}, {128, 256, 10, wpad / 4, 1, 1, ACT_RELU, false},
                        {256, 256, 10, wpad / 4, 1, 1, ACT_RELU, false}, {256, 256, 10, wpad / 4, 2, 1, ACT_RELU, false},
                        {256, 512, 5, wpad / 4, 1, 1, ACT_LEAKY, false}, {512, 512, 5, wpad / 4, 1, 1, ACT_LEAKY, true}};
    Shape s = shapes[layer];
    std::vector<Variant> vars;
    if (layer == 9) vars = {{"fp32 MFMA (shipped conv9)", f9, 256, false}, {"bf16x3 5x16 NT256 N-split", b9_a, 256, true},
                            {"bf16x3 5x16 NT128 2WG/CU", b9_b, 128, true}, {"bf16x3 5x48 NT128 N-split", b9_c, 128, true},
                            {"bf16x3 5x32 NT128 2x2", b9_d, 128, true}, {"bf16x3 5x32 NT128 2x2 2WG/CU", b9_e, 128, true},
                            {"bf16x3 5x64 NT64 M-split", b9_f, 64, true}, {"bf16x3 5x64 NT128 M-split", b9_g, 128, true},
                            {"bf16x3 5x32 NT256 2x2", b9_h, 256, true},
                            {"direct-B 5x16 NT128 2WG", d9_a, 128, true}, {"direct-B 5x16 NT128 3WG", d9_b, 128, true},
                            {"direct-B 5x16 NT256", d9_c, 256, true}, {"direct-B 5x32 NT128 N-split", d9_e, 128, true}};
    else if (layer == 6) vars = {{"fp32 MFMA (shipped conv6)", f6, 128, false}, {"bf16x3 10x16 NT128", b6_a, 128, true}, {"bf16x3 5x16 NT256", b6_b, 256, true},
                            {"bf16x3 10x32 NT64 2x2", b6_c, 64, true}, {"bf16x3 10x32 NT128 2x2", b6_d, 128, true},
                            {"direct-B 10x16 NT128", d6_a, 128, true}, {"direct-B 10x16 NT128 2WG", d6_b, 128, true},
                            {"direct-B 5x16 NT128 2WG", d6_c, 128, true}, {"direct-B 5x16 NT256", d6_d, 256, true}};
    else if (layer == 3) vars = {{"fp32 MFMA (shipped conv3)", f3, 128, false}, {"bf16x3 4x32 NT128", b3_a, 128, true}, {"bf16x3 4x32 NT128 2WG", b3_b, 128, true}, {"bf16x3 4x64 NT128 2x2", b3_c, 128, true},
                            {"direct-B 4x32 NT128", d3_a, 128, true}, {"direct-B 4x32 NT128 2WG", d3_b, 128, true},
                            {"direct-B 5x16 NT128 2WG", d3_c, 128, true}, {"direct-B 4x16 NT128 2WG", d3_d, 128, true}, {"direct-B 4x16 NT128 3WG", d3_e, 128, true}};
    else if (layer == 5) vars = {{"fp32 MFMA (shipped conv5)", f5, 128, false}, {"bf16x3 10x16 NT128", b6_a, 128, true}, {"direct-B 10x16 NT128", d6_a, 128, true},
                            {"direct-B 10x16 NT128 2WG", d6_b, 128, true}, {"direct-B 5x16 NT128 2WG", d6_c, 128, true}};
    else if (layer == 7) vars = {{"fp32 MFMA (shipped conv7)", f7, 128, false}, {"bf16x3 10x16 NT128 pool 2x1", b7_a, 128, true},
                            {"direct-B 10x16 NT128", d7_a, 128, true}, {"direct-B 10x16 NT128 2WG", d7_b, 128, true},
                            {"bf16x3 10x32 NT64 2x2", b7_b, 64, true}, {"bf16x3 6x16 NT128 2WG", b7_c, 128, true},
                            {"direct-B 6x16 NT128 2WG", d7_c, 128, true}, {"direct-B 2x32 NT128 2WG", d7_d, 128, true}};
    else if (layer == 8) vars = {{"fp32 MFMA (shipped conv8)", f8, 256, false}, {"bf16x3 5x16 NT128 2WG", b8_a, 128, true}, {"bf16x3 5x16 NT256", b8_b, 256, true},
                            {"direct-B 5x16 NT128 2WG", d8_a, 128, true}, {"direct-B 5x16 NT128 3WG", d8_b, 128, true}};
    else if (layer == 4) vars = {{"fp32 MFMA (shipped conv4)", f4, 128, false}, {"bf16x3 4x32 NT128 pool", b4_a, 128, true},
                            {"bf16x3 4x64 NT128 2x2 pool", b4_b, 128, true}, {"bf16x3 4x64 NT64 2x2 2WG", b4_c, 64, true},
                            {"direct-B 4x32 NT128", d4_a, 128, true}, {"direct-B 4x32 NT128 2WG", d4_b, 128, true},
                            {"direct-B 4x16 NT128 2WG", d4_c, 128, true}, {"direct-B 4x16 NT128 3WG", d4_d, 128, true}, {"direct-B 2x32 NT128 2WG", d4_e, 128, true}};
    else if (layer == 2) vars = {{"fp32 MFMA (shipped conv2)", f2, 64, false}, {"bf16x3 4x64 NT64 N-split", b2_a, 64, true},
                            {"bf16x3 4x64 NT64 M-split 2WG", b2_b, 64, true}, {"bf16x3 8x64 NT64 M-split", b2_c, 64, true},
                            {"direct-B 4x64 NT64 M-split 2WG", d2_a, 64, true}, {"direct-B 4x64 NT64 N-split 2WG", d2_b, 64, true},
                            {"bf16x3 8x32 NT64 2x2", b2_d, 64, true}, {"bf16x3 4x64 NT64 2x2", b2_e, 64, true},
                            {"direct-B 8x32 NT64 2x2", d2_c, 64, true}, {"direct-B 4x64 NT64 2x2", d2_d, 64, true},
                            {"bf16x3 4x32 NT64 2x2 2WG", b2_f, 64, true}, {"direct-B 4x32 NT64 2x2 2WG", d2_f, 64, true}, {"bf16x3 4x32 NT64 2x2 3WG", b2_g, 64, true}};
    else { printf("layer %d not covered\n", layer); return 1; }
    if (!getenv("ALLBF")) {            // default: the fp32 kernel, the shipped bf16x3 configuration, and the f16x2 candidates
        std::vector<Variant> keep = {vars[0]};
        const char *shipped[10] = {"", "", "bf16x3 4x32 NT64 2x2 2WG", "direct-B 5x16 NT128 2WG", "direct-B 4x16 NT128 2WG", "direct-B 5x16 NT128 2WG",
                                   "direct-B 5x16 NT128 2WG", "direct-B 2x32 NT128 2WG", "direct-B 5x16 NT128 2WG", "direct-B 5x16 NT128 2WG"};
        for (auto &v : vars) if (!strcmp(v.name, shipped[layer])) keep.push_back(v);
        vars = keep;
        if (layer == 9) { vars.insert(vars.end(), {{"f16x2 direct 5x16 NT128 2WG", h9_a, 128, 2}, {"f16x2 direct 5x32 NT128 2WG", h9_b, 128, 2}, {"f16x2 direct 5x32 NT128 1WG", h9_c, 128, 2},
                                  {"f16x2 direct 5x16 NT256 1WG", h9_d, 256, 2}, {"f16x2 direct 5x48 NT64 2WG", h9_f, 64, 2},
                                  
                                  }); }
        if (layer == 8) vars.insert(vars.end(), {{"f16x2 direct 5x16 NT128 2WG", h8_a, 128, 2}, {"f16x2 direct 5x32 NT128 2WG", h8_b, 128, 2}});
        if (layer == 6 || layer == 5) vars.insert(vars.end(), {{"f16x2 direct 5x16 NT128 2WG", h6_a, 128, 2}, {"f16x2 direct 10x16 NT128 2WG", h6_b, 128, 2},
                                  {"f16x2 direct 5x32 NT128 2WG", h6_c, 128, 2}});
        if (layer == 7) vars.insert(vars.end(), {{"f16x2 direct 2x32 NT128 2WG", h7_a, 128, 2}, {"f16x2 direct 10x16 NT128 2WG", h7_b, 128, 2}, {"f16x2 direct 2x64 NT128 2WG", h7_c, 128, 2}});
        if (layer == 4) vars.insert(vars.end(), {{"f16x2 direct 4x16 NT128 2WG", h4_a, 128, 2}, {"f16x2 direct 4x32 NT128 2WG", h4_b, 128, 2}});
        if (layer == 3) vars.insert(vars.end(), {{"f16x2 direct 5x16 NT128 2WG", h3_a, 128, 2}, {"f16x2 direct 4x32 NT128 2WG", h3_b, 128, 2}});
        if (layer == 2) vars.insert(vars.end(), {{"f16x2 lds 4x32 NT64 2x2 2WG", h2_a, 64, 2}, {"f16x2 direct 4x32 NT64 2x2", h2_c, 64, 2},
                                  {"f16x2 direct 4x64 NT64 2x2", h2_d, 64, 2}, {"f16x2 direct 8x32 NT64 2x2", h2_e, 64, 2}, {"f16x2 direct 4x64 NT64 M-split NS4", h2_f, 64, 2}});
    }
    const size_t xin = (size_t)n * s.H * s.W * s.cin;
    const int Hout = s.H / s.ph, Wout = s.W / s.pw;
    const size_t yout = (size_t)n * Hout * Wout * s.cout;
    const int c16max = ((s.cout + 255) / 256 * 256) / 16;
    std::vector<float> hx(xin), hb(c16max * 16), hs(c16max * 16), hh(c16max * 16);
    unsigned r = 12345;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xffff) / 32768.0f - 1.0f; };
    const int xbits = argc > 4 ? atoi(argv[4]) : 24;             // experiment: keep only the top `xbits` significand bits of x
    for (auto &v : hx) {
        v = rnd() * (1.0f + 0.001f * rnd());                     // full 24-bit mantissas
        if (xbits < 24) { unsigned b; memcpy(&b, &v, 4); b &= ~((1u << (24 - xbits)) - 1u); memcpy(&v, &b, 4); }
    }
    for (auto &v : hb) v = 0.1f * rnd();
    for (auto &v : hs) v = 1.0f + 0.2f * rnd();
    for (auto &v : hh) v = 0.1f * rnd();
    const float scale = 1.0f / sqrtf((float)s.cin * 9);
    const int wbits = argc > 5 ? atoi(argv[5]) : 24;
    auto W = [&](int co, int ci, int tap) {
        unsigned h = (unsigned)(co * 73856093u) ^ (unsigned)(ci * 19349663u) ^ (unsigned)(tap * 83492791u);
        h = h * 1664525u + 1013904223u;
        float v = (((h >> 8) & 0xffff) / 32768.0f - 1.0f) * scale;
        h = h * 1664525u + 1013904223u;
        v = v * (1.0f + 1e-3f * (((h >> 8) & 0xffff) / 32768.0f - 1.0f));
        if (wbits < 24) { unsigned b; memcpy(&b, &v, 4); b &= ~((1u << (24 - wbits)) - 1u); memcpy(&v, &b, 4); }
        return v;
    };
    float *dx, *dw, *db, *ds, *dh, *dy, *dy0;
    const size_t wbytes = (size_t)9 * s.cin * c16max * 16 * 6 + 1024;
    CK(hipMalloc(&dx, xin * 4)); CK(hipMalloc(&dw, wbytes)); CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMalloc(&ds, hb.size() * 4)); CK(hipMalloc(&dh, hb.size() * 4)); CK(hipMalloc(&dy, yout * 4)); CK(hipMalloc(&dy0, yout * 4));
    CK(hipMemcpy(dx, hx.data(), xin * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, hh.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * n * s.H * s.W * (double)s.cout * s.cin * 9;
    std::vector<float> y0(yout), y(yout);
    printf("layer conv%d: %d->%d @%dx%d n=%d  %.1f GFLOP\n", layer, s.cin, s.cout, s.H, s.W, n, flops / 1e9);
    // float64 reference on sampled outputs (pre-pool positions are needed for pooled layers: sample whole pooled outputs)
    const int NSAMP = 400;
    std::vector<size_t> samp(NSAMP);
    std::vector<double> ref(NSAMP);
    for (int k = 0; k < NSAMP; ++k) {
        r = r * 1664525u + 1013904223u;
        samp[k] = (size_t)(r % (unsigned)(yout / 97)) * 97 % yout;
        const size_t idx = samp[k];
        const int co = idx % s.cout; size_t p = idx / s.cout; const int wo = p % Wout; p /= Wout; const int ho = p % Hout; const int img = p / Hout;
        double best = -1e300;
        for (int py = 0; py < s.ph; ++py) for (int px = 0; px < s.pw; ++px) {
            const int hc = ho * s.ph + py, wc = wo * s.pw + px;
            double acc = 0;
            for (int tap = 0; tap < 9; ++tap) {
                const int hi = hc + tap / 3 - 1, wi = wc + tap % 3 - 1;
                if (hi < 0 || hi >= s.H || wi < 0 || wi >= s.W) continue;
                const float *xp = &hx[(((size_t)img * s.H + hi) * s.W + wi) * s.cin];
                for (int ci = 0; ci < s.cin; ++ci) acc += (double)xp[ci] * (double)W(co, ci, tap);
            }
            double t = acc + hb[co];
            t = s.act == ACT_RELU ? (t > 0 ? t : 0) : (t > 0 ? t : 0.01 * t);
            if (s.bn) t = t * hs[co] + hh[co];
            best = t > best ? t : best;
        }
        ref[k] = best;
    }
    for (size_t vi = 0; vi < vars.size(); ++vi) {
        const int cout16 = ((s.cout + vars[vi].nt - 1) / vars[vi].nt * vars[vi].nt) / 16;
        if (!vars[vi].bf) {
            std::vector<float> hw((size_t)9 * (s.cin / 16) * cout16 * 256);
            size_t o = 0;
            for (int tap = 0; tap < 9; ++tap) for (int g = 0; g < s.cin / 16; ++g) for (int sg = 0; sg < cout16; ++sg)
                for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j, ++o) {
                    const int co = 16 * sg + (lane & 15), ci = 16 * g + 4 * (lane >> 4) + j;
                    hw[o] = co < s.cout ? W(co, ci, tap) : 0.f;
                }
            CK(hipMemcpy(dw, hw.data(), o * 4, hipMemcpyHostToDevice));
        } else if (vars[vi].bf == 2) {
            std::vector<uint16_t> hw((size_t)9 * (s.cin / 32) * cout16 * 2 * 64 * 8);
            size_t o = 0;
            for (int tap = 0; tap < 9; ++tap) for (int g = 0; g < s.cin / 32; ++g) for (int sg = 0; sg < cout16; ++sg)
                for (int pl = 0; pl < 2; ++pl) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j, ++o) {
                    const int co = 16 * sg + (lane & 15), ci = 32 * g + 8 * (lane >> 4) + j;
                    const float w = co < s.cout ? W(co, ci, tap) : 0.f;
                    const _Float16 h = (_Float16)w;
                    const _Float16 l = (_Float16)((w - (float)h) * 2048.0f);
                    const _Float16 v = pl ? l : h;
                    memcpy(&hw[o], &v, 2);
                }
            CK(hipMemcpy(dw, hw.data(), o * 2, hipMemcpyHostToDevice));
        } else {
            std::vector<uint16_t> hw((size_t)9 * (s.cin / 32) * cout16 * 3 * 64 * 8);
            size_t o = 0;
            for (int tap = 0; tap < 9; ++tap) for (int g = 0; g < s.cin / 32; ++g) for (int sg = 0; sg < cout16; ++sg)
                for (int pl = 0; pl < 3; ++pl) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j, ++o) {
                    const int co = 16 * sg + (lane & 15), ci = 32 * g + 8 * (lane >> 4) + j;
                    const float w = co < s.cout ? W(co, ci, tap) : 0.f;
                    unsigned wb; memcpy(&wb, &w, 4);
                    const unsigned h = wb & 0xffff0000u; float hf; memcpy(&hf, &h, 4);
                    const float r1 = w - hf; unsigned r1b; memcpy(&r1b, &r1, 4);
                    const unsigned m = r1b & 0xffff0000u; float mf; memcpy(&mf, &m, 4);
                    const float r2 = r1 - mf; unsigned r2b; memcpy(&r2b, &r2, 4);
                    const unsigned parts[3] = {getenv("ONLYMID") ? 0u : h, m, r2b & 0xffff0000u};
                    hw[o] = (uint16_t)(parts[pl] >> 16);
                }
            CK(hipMemcpy(dw, hw.data(), o * 2, hipMemcpyHostToDevice));
        }
        ConvArgs a{};
        a.x = dx; a.wfrag = dw; a.bias = db; a.bn_scale = ds; a.bn_shift = dh; a.y = vi == 0 ? dy0 : dy;
        a.n = n; a.H = s.H; a.W = s.W; a.Ho = s.H; a.Wo = s.W; a.cin = s.cin; a.cout16 = cout16; a.cout_valid = s.cout; a.out_stride = s.cout;
        CK(hipMemsetAsync(a.y, 0, yout * 4, st));
        vars[vi].fn(a, st);
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        float best = 1e30f, sum = 0;
        const int reps = 5;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipEventRecord(e0, st)); vars[vi].fn(a, st); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        std::vector<float> &yy = vi == 0 ? y0 : y;
        CK(hipMemcpy(yy.data(), a.y, yout * 4, hipMemcpyDeviceToHost));
        double maxdiff = 0, maxref = 0, rmsref = 0;
        if (vi) for (size_t k = 0; k < yout; ++k) { double d = fabs((double)y[k] - y0[k]); if (d > maxdiff) maxdiff = d; }
        for (int k = 0; k < NSAMP; ++k) { double d = fabs((double)yy[samp[k]] - ref[k]); maxref = d > maxref ? d : maxref; rmsref += d * d;
            if (getenv("DBG") && vi == 1 && d > 3e-6) { size_t p = samp[k] / s.cout; printf("    bad sample: co %d wo %d ho %d img %d err %.2e\n", (int)(samp[k] % s.cout), (int)(p % Wout), (int)((p / Wout) % Hout), (int)(p / Wout / Hout), d); } }
        if (getenv("DBG") && vi == 1 && s.ph == 1) {
            for (int k = 0; k < 4; ++k) {          // S_h = sum x * hi(w), S_hm = + x * mid(w): which one does the GPU match?
                const size_t idx = samp[k];
                const int co = idx % s.cout; size_t p = idx / s.cout; const int wo = p % Wout; p /= Wout; const int ho = p % Hout; const int img = p / Hout;
                double sh = 0, shm = 0, full = 0;
                for (int tap = 0; tap < 9; ++tap) {
                    const int hi = ho + tap / 3 - 1, wi = wo + tap % 3 - 1;
                    if (hi < 0 || hi >= s.H || wi < 0 || wi >= s.W) continue;
                    const float *xp = &hx[(((size_t)img * s.H + hi) * s.W + wi) * s.cin];
                    for (int ci = 0; ci < s.cin; ++ci) {
                        const float w = W(co, ci, tap); unsigned wb; memcpy(&wb, &w, 4);
                        const unsigned hb_ = wb & 0xffff0000u; float hf; memcpy(&hf, &hb_, 4);
                        const float r1 = w - hf; unsigned r1b; memcpy(&r1b, &r1, 4); const unsigned mb = r1b & 0xffff0000u; float mf; memcpy(&mf, &mb, 4);
                        sh += (double)xp[ci] * hf; shm += (double)xp[ci] * ((double)hf + mf); full += (double)xp[ci] * w;
                    }
                }
                auto post = [&](double t) { t += hb[co]; t = s.act == ACT_RELU ? (t > 0 ? t : 0) : (t > 0 ? t : 0.01 * t); return s.bn ? t * hs[co] + hh[co] : t; };
                printf("    sample %d: gpu %.9f  full %.9f  S_h %.9f  S_hm %.9f  S_m %.9f\n", k, yy[idx], post(full), post(sh), post(shm), post(shm - sh));
            }
        }
        printf("  %-34s avg %.3f ms best %.3f ms  %.1f TF(alg)  max|d vs fp32 kernel| %.2e  vs float64: max %.2e rms %.2e\n", vars[vi].name, sum / reps, best,
               flops / (sum / reps * 1e-3) / 1e12, maxdiff, maxref, sqrt(rmsref / NSAMP));
    }
    return 0;
}
