// PARKED (round 6): this tool drives per-workgroup phase stamps (POCR_BF16X3_TRACE) and the POCR_BF16X3_DBG ablations of csrc/conv_bf16x3.hpp, which were compile-time switches of the
// library until round 6 removed them (the measurements they produced: profiles/r02_conv_bf16x3_bench.txt, r03_conv_f16x2_ablation.txt,
// r03_conv_tile_trace.txt, r05_conv_rows.txt).  It builds against the round-5 tree: `git worktree add /tmp/r05 4aef8f0` and compile there.
// conv_ablate.hip — where the time of the split-precision conv kernel goes: one layer's shipped configuration, built
// several times with parts of the main loop switched off (-DPOCR_BF16X3_DBG=n: 1 no A reads, 2 no weight loads, 4 no A
// staging, 8 no barrier; results are then wrong, only the time matters).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -DPOCR_BF16X3_DBG=<n> -o tools/bin/conv_ablate_<n> tools/conv_ablate.hip
// Run  : tools/bin/conv_ablate_<n> [layer 9|8|6|4|2] [n_lines=256] [w_pad=576] [only this variant of the layer's list, -1 = all] [sustained launches]
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
// shipped P2 configurations
VP(p9, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2, true)
VP(p8, 5, 1, 2, 1, 1, 1, ACT_LEAKY, false, 2, true)
VP(p7, 2, 2, 2, 1, 2, 1, ACT_RELU, false, 2, true)
VP(p6, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2, true)
VP(p4, 4, 1, 2, 1, 2, 2, ACT_RELU, false, 2, true)
VP(p3, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2, true)
VP(p2, 4, 2, 2, 2, 2, 2, ACT_RELU, false, 3, false)
// candidates (POCR sweep of round 3, P2 input / output)
VP(p2_b, 8, 2, 2, 2, 2, 2, ACT_RELU, false, 2, false)     // conv2: 8x32, LDS weights
VP(p2_c, 4, 2, 2, 2, 2, 2, ACT_RELU, false, 3, true)      // conv2: 4x32, weights from L2
VP(p2_d, 8, 2, 2, 2, 2, 2, ACT_RELU, false, 2, true)      // conv2: 8x32, weights from L2
VP(p2_e, 4, 4, 4, 4, 2, 2, ACT_RELU, false, 2, false)     // conv2: 4x64, waves split the pixels, NS 4
VP(p2_f, 4, 4, 4, 4, 2, 2, ACT_RELU, false, 2, true)
VP(p2_g, 4, 2, 2, 2, 2, 2, ACT_RELU, false, 2, false)     // shipped tile, 2 WG/CU budget
VP(p2_h, 2, 4, 2, 2, 2, 2, ACT_RELU, false, 3, false)     // 2x64
VP(p3_b, 4, 1, 2, 1, 1, 1, ACT_RELU, false, 3, true)      // conv3/5/6: 4x16, 3 WG/CU   (H = 20 only)
VP(p3_c, 4, 2, 2, 1, 1, 1, ACT_RELU, false, 2, true)      // 4x32 (MS 8)
VP(p3_d, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 3, true)      // shipped tile, 3 WG/CU budget
VP(p3_e, 2, 2, 2, 1, 1, 1, ACT_RELU, false, 3, true)      // 2x32
VP(p3_f, 5, 2, 1, 1, 1, 1, ACT_RELU, false, 2, true)      // 5x32, NS 1 (NT 64): MS 10
VP(p3_g, 10, 1, 1, 1, 1, 1, ACT_RELU, false, 2, true)     // 10x16, NS 1 (NT 64): MS 10
VP(p4_b, 4, 1, 2, 1, 2, 2, ACT_RELU, false, 3, true)      // conv4: shipped tile, 3 WG/CU budget
VP(p4_c, 4, 2, 2, 1, 2, 2, ACT_RELU, false, 2, true)      // 4x32
VP(p4_d, 2, 2, 2, 1, 2, 2, ACT_RELU, false, 3, true)      // 2x32
VP(p4_e, 10, 1, 1, 1, 2, 2, ACT_RELU, false, 2, true)     // conv4: 10x16 NS 1
VP(p4_f, 4, 2, 1, 1, 2, 2, ACT_RELU, false, 2, true)      // conv4: 4x32 NS 1 (MS 8)
VP(p4_g, 20, 1, 1, 1, 2, 2, ACT_RELU, false, 2, true)     // conv4: 20x16 NS 1 (MS 20)
VP(p7_b, 2, 2, 2, 1, 2, 1, ACT_RELU, false, 3, true)
VP(p7_c, 2, 1, 2, 1, 2, 1, ACT_RELU, false, 3, true)      // 2x16 (MS 2)
VP(p7_d, 10, 1, 1, 1, 2, 1, ACT_RELU, false, 2, true)     // 10x16 NS 1
VP(p7_e, 2, 4, 1, 1, 2, 1, ACT_RELU, false, 2, true)      // 2x64 NS 1 (MS 8)
VP(p9_b, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 3, true)      // conv8/9: 3 WG/CU budget
VP(p9_c, 5, 2, 1, 1, 1, 1, ACT_LEAKY, true, 2, true)      // 5x32 NS 1 (NT 64)
VP(p9_d, 5, 1, 1, 1, 1, 1, ACT_LEAKY, true, 4, true)      // 5x16 NS 1 (NT 64), 4 WG/CU
// one workgroup per CU (512 registers per lane): every weight fragment feeds 10 row strips AND every A fragment two channel tiles
VP(p9_e, 5, 2, 2, 1, 1, 1, ACT_LEAKY, true, 1, true)      // conv8/9: 5x32 NT128 (MS 10, NS 2)
VP(p6_x, 10, 1, 2, 1, 1, 1, ACT_RELU, false, 1, true)     // conv3/5/6: 10x16 NT128 (MS 10, NS 2)
VP(p6_y, 10, 2, 1, 1, 1, 1, ACT_RELU, false, 1, true)     // 10x32 NT64 (MS 20, NS 1)
VP(p4_x, 10, 1, 2, 1, 2, 2, ACT_RELU, false, 1, true)     // conv4: 10x16 NT128
VP(p7_x, 10, 1, 2, 1, 2, 1, ACT_RELU, false, 1, true)     // conv7: 10x16 NT128
V(h9, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2, true, 2)
// GEMM mode (1x1): rows x cin -> cout, "pixels" = rows.  (TH, MW, NS, WM, MINW, BDIR, PRE_IN, PRE_OUT)
#define VG(NAME, MW, NS, WM, MINW, BDIR, PIN, POUT) static void NAME(ConvArgs a, hipStream_t st) { \
    launch(conv3x3_bf16x3_kernel<1, MW, NS, WM, 1, 1, ACT_NONE, false, MINW, BDIR, 1, 1, 0, 0, false, 2, PIN, POUT>, 1, 16 * MW, NS * (4 / WM) * 16, a, st); }
VG(g_ship, 8, 4, 2, 2, false, false, false)     // shipped: 128 x 128, LDS weights, split inside
VG(g_ship_d, 8, 4, 2, 2, true, false, false)   // fp32 in, weights straight from L2
VG(g_pin, 8, 4, 2, 2, false, true, false)       // the same with pre-split input
VG(g_pin_d, 8, 4, 2, 2, true, true, false)      // + weights straight from L2
VG(g_pin_out, 8, 4, 2, 2, false, true, true)    // pre-split in and out
VG(g_pin_8x2, 8, 2, 1, 2, true, true, false)    // 128 rows x 128 cols, waves split the columns (MS 8, NS 2), direct weights
VG(g_pin_4x4, 4, 4, 1, 1, true, true, false)    // 64 rows x 256 cols (MS 4, NS 4), direct weights
VG(g_pin_8x4m, 8, 4, 4, 2, false, true, false)  // 128 rows x 64 cols, waves split the rows (MS 2, NS 4), LDS weights
VG(g_pin_16, 16, 4, 4, 1, false, true, false)   // 256 rows x 64 cols, waves split the rows (MS 4, NS 4), LDS weights
#ifdef POCR_BF16X3_TRACE
#include <map>
#include <algorithm>
// -DPOCR_BF16X3_TRACE: phase stamps of the LAST launch (100 MHz wall clock: 0 entry, 1 first tile staged, 2 main loop done,
// 3 stores issued, 4 stores drained) and, per CU, how long the workgroup slots sat between two workgroups
__global__ void trace_copy_kernel(unsigned long long *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = g_conv_trace[i];
}
static void trace_report(const ConvArgs &a) {
    (void)a;
    const size_t nb = 1u << 15;         // every slot; blocks that never ran keep a zero stamp (the launch copies clear nothing: run one layer per process)
    std::vector<unsigned long long> t(nb * 8);
    unsigned long long *dp;
    CK(hipMalloc(&dp, nb * 8 * sizeof(*dp)));
    hipLaunchKernelGGL(trace_copy_kernel, dim3(256), dim3(256), 0, 0, dp, nb * 8);
    CK(hipMemcpy(t.data(), dp, nb * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    { hipError_t fe = hipFree(dp); if (fe != hipSuccess) printf("    (hipFree: %d)\n", (int)fe); }
    unsigned long long tmin = ~0ull, tmax = 0;
    double ph[4] = {0, 0, 0, 0};
    std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
    for (size_t b = 0; b < nb; ++b) {
        const unsigned long long *q = &t[b * 8];
        if (!q[4]) continue;
        tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[4]);
        for (int k = 0; k < 4; ++k) ph[k] += (double)(q[k + 1] - q[k]);
        cu[(unsigned)((q[5] >> 32) << 8) | (unsigned)((q[5] >> 8) & 0xff)].push_back({q[0], q[4]});
    }
    const double us = 0.01;
    size_t ran = 0;
    for (auto &kv : cu) ran += kv.second.size();
    if (cu.empty()) { printf("    trace: no stamps (t0 of block 0: %llu %llu %llu %llu %llu)\n", t[0], t[1], t[2], t[3], t[4]); fflush(stdout); return; }
    printf("    trace: span %.1f us over %zu workgroups on %zu CUs; mean per workgroup: stage-first-tile %.2f us, main loop %.2f us, epilogue issue %.2f us, store drain %.2f us\n",
           (tmax - tmin) * us, ran, cu.size(), ph[0] / ran * us, ph[1] / ran * us, ph[2] / ran * us, ph[3] / ran * us);
    double busy = 0, first = 0, last = 0; size_t maxwg = 0, minwg = 1 << 30;
    for (auto &kv : cu) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end());
        for (auto &w : v) busy += (double)(w.second - w.first);
        first += (double)(v.front().first - tmin); last += (double)(tmax - v.back().second);
        maxwg = std::max(maxwg, v.size()); minwg = std::min(minwg, v.size());
    }
    printf("    per CU: workgroups %zu..%zu, resident workgroup-time / span = %.2f (2.0 = both slots always occupied), mean first start +%.1f us, mean last end -%.1f us\n",
           minwg, maxwg, busy / cu.size() / (double)(tmax - tmin), first / cu.size() * us, last / cu.size() * us);
    auto &v = cu.begin()->second;
    printf("    CU %#x timeline (start..end us):", cu.begin()->first);
    for (size_t i = 0; i < v.size() && i < 12; ++i) printf(" %.1f..%.1f", (v[i].first - tmin) * us, (v[i].second - tmin) * us);
    printf("\n");
}
#endif
int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int layer = argc > 1 ? atoi(argv[1]) : 9, wpad = argc > 3 ? atoi(argv[3]) : 576;
    int n = argc > 2 ? atoi(argv[2]) : 256;
    if (layer >= 100) n = 1;
    Shape s; std::vector<Variant> vars;
    if (layer >= 100) {       // GEMM mode: 100 = 512 -> 2048 (FFN1 / LSTM projection), 101 = 2048 -> 512 (FFN2), 102 = 512 -> 512
        const int rows = 53248;
        s = {layer == 101 ? 2048 : 512, layer == 100 ? 2048 : 512, 1, rows, 1, 1};
        vars = {{"GEMM 128x128 LDS, split inside (shipped)", g_ship, 2}, {"GEMM 128x128 direct weights, split inside", g_ship_d, 2}, {"GEMM P2 in", g_pin, 2}, {"GEMM P2 in, direct weights", g_pin_d, 2},
                {"GEMM P2 in + out", g_pin_out, 2}, {"GEMM P2 128x128 N-split direct", g_pin_8x2, 2}, {"GEMM P2 64x256 direct", g_pin_4x4, 2},
                {"GEMM P2 128x64 M-split LDS", g_pin_8x4m, 2}, {"GEMM P2 256x64 M-split LDS", g_pin_16, 2}};
    } else if (layer == 9) { s = {512, 512, 5, wpad / 4, 1, 1}; vars = {{"f16x2 in-kernel split 5x16 NT128", h9, 2}, {"P2 5x16 NT128 (shipped)", p9, 2}, {"P2 5x16 NT128 3WG", p9_b, 2}, {"P2 5x32 NT64", p9_c, 2}, {"P2 5x16 NT64 4WG", p9_d, 2}, {"P2 5x32 NT128 1WG", p9_e, 2}}; }
    else if (layer == 8) { s = {256, 512, 5, wpad / 4, 1, 1}; vars = {{"P2 5x16 NT128 (shipped)", p8, 2}}; }
    else if (layer == 7) { s = {256, 256, 10, wpad / 4, 2, 1}; vars = {{"P2 2x32 NT128 (shipped)", p7, 2}, {"P2 2x32 3WG", p7_b, 2}, {"P2 2x16 3WG", p7_c, 2}, {"P2 10x16 NT64", p7_d, 2}, {"P2 2x64 NT64", p7_e, 2}, {"P2 10x16 NT128 1WG", p7_x, 2}}; }
    else if (layer == 6 || layer == 5) { s = {layer == 5 ? 128 : 256, 256, 10, wpad / 4, 1, 1}; vars = {{"P2 5x16 NT128 (shipped)", p6, 2}, {"P2 5x16 3WG", p3_d, 2}, {"P2 2x32 3WG", p3_e, 2}, {"P2 5x32 NT64", p3_f, 2}, {"P2 10x16 NT64", p3_g, 2}, {"P2 10x16 NT128 1WG", p6_x, 2}, {"P2 10x32 NT64 1WG", p6_y, 2}}; }
    else if (layer == 4) { s = {128, 128, 20, wpad / 2, 2, 2}; vars = {{"P2 4x16 NT128 (shipped)", p4, 2}, {"P2 4x16 3WG", p4_b, 2}, {"P2 4x32", p4_c, 2}, {"P2 10x16 NT64", p4_e, 2}, {"P2 4x32 NT64", p4_f, 2}, {"P2 10x16 NT128 1WG", p4_x, 2}}; }
    else if (layer == 3) { s = {64, 128, 20, wpad / 2, 1, 1}; vars = {{"P2 5x16 NT128 (shipped)", p3, 2}, {"P2 4x16 3WG", p3_b, 2}, {"P2 4x32", p3_c, 2}, {"P2 5x16 3WG", p3_d, 2}, {"P2 2x32 3WG", p3_e, 2}, {"P2 5x32 NT64", p3_f, 2}, {"P2 10x16 NT64", p3_g, 2}, {"P2 10x16 NT128 1WG", p6_x, 2}, {"P2 10x32 NT64 1WG", p6_y, 2}}; }
    else { s = {64, 64, 40, wpad, 2, 2}; vars = {{"P2 lds 4x32 NT64 3WG (shipped)", p2, 2}, {"P2 lds 8x32", p2_b, 2}, {"P2 direct 4x32 3WG", p2_c, 2}, {"P2 direct 8x32", p2_d, 2}, {"P2 lds 4x64 M-split NS4", p2_e, 2}, {"P2 direct 4x64 M-split NS4", p2_f, 2}, {"P2 lds 4x32 2WG", p2_g, 2}, {"P2 direct 10x16 NS1", p4_e, 2}, {"P2 direct 4x32 NS1", p4_f, 2}}; }
    const size_t xin = (size_t)n * s.H * s.W * s.cin, yout = (size_t)n * (s.H / s.ph) * (s.W / s.pw) * s.cout;
    std::vector<float> hx(xin);
    unsigned r = 12345;
    for (auto &v : hx) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xffff) / 32768.0f - 1.0f; }
    std::vector<uint16_t> hw((size_t)9 * s.cin * 2048 * 3);
    for (auto &v : hw) { r = r * 1664525u + 1013904223u; v = (uint16_t)(0x2c00 + ((r >> 9) & 0x3ff)) | (uint16_t)((r >> 3) & 0x8000); }   // random small f16 / bf16 bit patterns
    float *dx, *dw, *db, *dy;
    CK(hipMalloc(&dx, xin * 4)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&db, 4096 * 4)); CK(hipMalloc(&dy, yout * 4));
    CK(hipMemcpy(dx, hx.data(), xin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(db, 0, 4096 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * n * s.H * s.W * (double)s.cout * s.cin * (layer >= 100 ? 1 : 9);
    printf("DBG=%d conv%d %d->%d @%dx%d n=%d\n", POCR_BF16X3_DBG, layer, s.cin, s.cout, s.H, s.W, n);
    const int sustained = argc > 5 ? atoi(argv[5]) : 0;  // > 0: time this many back-to-back launches at the power cap instead of 10 single ones
    const int only = argc > 4 ? atoi(argv[4]) : -1;      // run (and trace) only this variant of the layer's list
    if (only >= 0 && only < (int)vars.size()) vars = {vars[only]};
    for (auto &v : vars) {
        ConvArgs a{};
        a.x = dx; a.wfrag = dw; a.bias = db; a.bn_scale = db; a.bn_shift = db; a.y = dy;
        a.n = n; a.H = s.H; a.W = s.W; a.Ho = s.H; a.Wo = s.W; a.cin = s.cin; a.cout16 = s.cout / 16; a.xcd_g = 2; a.cout_valid = s.cout; a.out_stride = s.cout;
        for (int w = 0; w < 3; ++w) v.fn(a, st);
        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        float sum = 0, best = 1e30f;
        if (sustained > 0) {
            // power-limited steady state (profiles/r04_power_cap.txt): `sustained` launches back to back; the first third brings the
            // package to its cap and is not timed - what this measures is the variant's ENERGY per launch, which is what the engine pays
            for (int rep = 0; rep < sustained / 3; ++rep) v.fn(a, st);
            CK(hipEventRecord(e0, st));
            for (int rep = 0; rep < sustained; ++rep) v.fn(a, st);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  %-28s sustained over %d launches %.3f ms  %.1f TF(alg)\n", v.name, sustained, ms / sustained, flops / (ms / sustained * 1e-3) / 1e12);
            continue;
        }
        for (int rep = 0; rep < 10; ++rep) {
            CK(hipEventRecord(e0, st)); v.fn(a, st); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; best = ms < best ? ms : best;
        }
        printf("  %-28s avg %.3f ms best %.3f ms  %.1f TF(alg)\n", v.name, sum / 10, best, flops / (sum / 10 * 1e-3) / 1e12);
#ifdef POCR_BF16X3_TRACE
        if (&v == &vars[0]) trace_report(a);
#endif
    }
    return 0;
}
