// conv_bench.hip — within-probe A/B of conv_igemm_kernel tile configurations, main-loop pipelines and
// ablations on one layer shape.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -o /tmp/conv_bench tools/conv_bench.hip
// Run  : /tmp/conv_bench <layer 2..9> [n_lines=256] [w_pad=576]
// Every variant runs on the same random input/weights; outputs are compared with variant 0
// (the k-order inside a 16-channel group is identical for all tilings, so they must agree exactly).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../pero_ocr_amd/csrc/conv_igemm.hpp"
using namespace pocr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { int cin, cout, H, W, ph, pw, act; bool bn; };

template <class K>
static void launch(K kern, int TH, int TW, int NT, int nthr, ConvArgs a, hipStream_t st) {
    a.tiles_w = (a.Wo + TW - 1) / TW; a.tiles_h = (a.Ho + TH - 1) / TH; a.tiles_n = (a.cout16 * 16) / NT;
    size_t blocks = conv_grid_blocks(a);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(nthr), 0, st, a);
}

struct Variant { const char *name; void (*fn)(ConvArgs, hipStream_t); int nt; };

#define VARP(NAME, TH, MW, NS, NW, KC, PH, PW, ACT, BN, PIPE, ABL)                                                 \
    static void NAME(ConvArgs a, hipStream_t st) {                                                                 \
        launch(conv_igemm_kernel<3, 3, 1, 1, TH, MW, NS, NW, KC, PH, PW, ACT, BN, STAGE_F32_NHWC, PIPE, ABL>, TH,   \
               16 * MW, NS * NW * 16, NW * 64, a, st);                                                             \
    }
#define VARW(NAME, TH, MW, NS, NW, PH, PW, ACT, BN, MINW)                                                           \
    static void NAME(ConvArgs a, hipStream_t st) {                                                                 \
        launch(conv_igemm_kernel<3, 3, 1, 1, TH, MW, NS, NW, 16, PH, PW, ACT, BN, STAGE_F32_NHWC, PIPE_INTERLEAVED, 0, MINW>, TH, \
               16 * MW, NS * NW * 16, NW * 64, a, st);                                                             \
    }
VARW(h5_w3_nt128, 5, 1, 2, 4, 1, 1, ACT_LEAKY, true, 3)
VARW(h5_w4_nt64,  5, 1, 1, 4, 1, 1, ACT_LEAKY, true, 4)
VARW(h5_w3_nt256, 5, 1, 4, 4, 1, 1, ACT_LEAKY, true, 3)
VARW(h10_w3,      5, 1, 2, 4, 1, 1, ACT_RELU, false, 3)
VARW(h20_w3,      4, 2, 1, 4, 1, 1, ACT_RELU, false, 3)
VARW(h20_w3_ns2,  4, 2, 2, 4, 1, 1, ACT_RELU, false, 3)
VARW(p22_w3_nt64, 4, 4, 1, 4, 2, 2, ACT_RELU, false, 3)
VARW(p22_w3_nt128, 4, 2, 2, 4, 2, 2, ACT_RELU, false, 3)
VARW(p22_w3_mw2_nt64, 4, 2, 1, 4, 2, 2, ACT_RELU, false, 3)
VARW(p22_w4_mw2_nt64, 4, 2, 1, 4, 2, 2, ACT_RELU, false, 4)
#define P0 PIPE_PLAIN
#define P3 PIPE_INTERLEAVED
#define P4 PIPE_DEEP
#define P5 PIPE_GLDS
#define P6 PIPE_BREG
#define P7 PIPE_DEEP3
// conv8/9-shaped (H = 5)
VARP(h5_plain,      5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P0, 0)
VARP(h5_plain_kc32, 5, 1, 4, 4, 32, 1, 1, ACT_LEAKY, true, P0, 0)
VARP(h5_plain_nw8,  5, 1, 2, 8, 16, 1, 1, ACT_LEAKY, true, P0, 0)
VARP(h5_plain_mw2,  5, 2, 2, 4, 16, 1, 1, ACT_LEAKY, true, P0, 0)
VARP(h5_p3,         5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P3, 0)
VARP(h5_p3_nt128,   5, 1, 2, 4, 16, 1, 1, ACT_LEAKY, true, P3, 0)
VARP(h5_p3_nt64,    5, 1, 1, 4, 16, 1, 1, ACT_LEAKY, true, P3, 0)
VARP(h5_p4,         5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P4, 0)
VARP(h5_p5,         5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P5, 0)
VARP(h5_p5_nt128,   5, 1, 2, 4, 16, 1, 1, ACT_LEAKY, true, P5, 0)
VARP(h5_p4_nt128,   5, 1, 2, 4, 16, 1, 1, ACT_LEAKY, true, P4, 0)
VARP(h5_p6,         5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P6, 0)
VARP(h5_p7,         5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P7, 0)
VARP(h5_p7_nt128,   5, 1, 2, 4, 16, 1, 1, ACT_LEAKY, true, P7, 0)
VARP(h5_p3_mw2,     5, 2, 2, 4, 16, 1, 1, ACT_LEAKY, true, P3, 0)
VARP(h5_p4_mw2,     5, 2, 2, 4, 16, 1, 1, ACT_LEAKY, true, P4, 0)
VARP(h5_p6_nt128,   5, 1, 2, 4, 16, 1, 1, ACT_LEAKY, true, P6, 0)
VARP(h5_a1,         5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P3, 1)
VARP(h5_a2,         5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P3, 2)
VARP(h5_a3,         5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P3, 3)
VARP(h5_a4,         5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P3, 4)
VARP(h5_a8,         5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P3, 8)
VARP(h5_a15,        5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P3, 15)
VARP(h5_a18,        5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, P3, 18)
// conv5/6-shaped (H = 10)
VARP(h10_plain,     10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, P0, 0)
VARP(h10_plain_th5, 5, 1, 4, 4, 16, 1, 1, ACT_RELU, false, P0, 0)
VARP(h10_p3,        10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, P3, 0)
VARP(h10_p3_th5,    5, 1, 4, 4, 16, 1, 1, ACT_RELU, false, P3, 0)
VARP(h10_p4,        10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, P4, 0)
VARP(h10_p5,        10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, P5, 0)
VARP(h10_p6,        10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, P6, 0)
VARP(h10_p7,        10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, P7, 0)
VARP(h10_p6_th5,    5, 1, 4, 4, 16, 1, 1, ACT_RELU, false, P6, 0)
// conv3-shaped (H = 20, no pool) and conv2/4 (pool 2x2)
VARP(h20_plain,     4, 2, 2, 4, 16, 1, 1, ACT_RELU, false, P0, 0)
VARP(h20_p3,        4, 2, 2, 4, 16, 1, 1, ACT_RELU, false, P3, 0)
VARP(h20_p3_th10,   10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, P3, 0)
VARP(h20_p4,        4, 2, 2, 4, 16, 1, 1, ACT_RELU, false, P4, 0)
VARP(h20_p6,        4, 2, 2, 4, 16, 1, 1, ACT_RELU, false, P6, 0)
VARP(h20_p7,        4, 2, 2, 4, 16, 1, 1, ACT_RELU, false, P7, 0)
VARP(p22_plain,     4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, P0, 0)
VARP(p22_plain_nt64, 4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, P0, 0)
VARP(p22_p3,        4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, P3, 0)
VARP(p22_p3_nt64,   4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, P3, 0)
VARP(p22_p3_th10,   10, 1, 2, 4, 16, 2, 2, ACT_RELU, false, P3, 0)
VARP(p22_p4,        4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, P4, 0)
VARP(p22_p5,        4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, P5, 0)
VARP(p22_p5_nt64,   4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, P5, 0)
VARP(p22_p6,        4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, P6, 0)
VARP(p22_p7,        4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, P7, 0)
VARP(p22_p7_nt64,   4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, P7, 0)
VARP(p22_p6_nt64,   4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, P6, 0)
VARP(p22_p4_nt64,   4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, P4, 0)

int main(int argc, char **argv) {
    int layer = argc > 1 ? atoi(argv[1]) : 9;
    int n = argc > 2 ? atoi(argv[2]) : 256;
    int wpad = argc > 3 ? atoi(argv[3]) : 576;
    Shape shapes[10] = {{}, {}, {64, 64, 40, wpad, 2, 2, ACT_RELU, false}, {64, 128, 20, wpad / 2, 1, 1, ACT_RELU, false},
                        {128, 128, 20, wpad / 2, 2, 2, ACT_RELU, false}, {128, 256, 10, wpad / 4, 1, 1, ACT_RELU, false},
                        {256, 256, 10, wpad / 4, 1, 1, ACT_RELU, false}, {256, 256, 10, wpad / 4, 2, 1, ACT_RELU, false},
                        {256, 512, 5, wpad / 4, 1, 1, ACT_LEAKY, false}, {512, 512, 5, wpad / 4, 1, 1, ACT_LEAKY, true}};
    Shape s = shapes[layer];
    if (argc > 4) s.cin = atoi(argv[4]);     // experiment: longer K loop per workgroup
    std::vector<Variant> vars;
    if (layer >= 8) vars = {{"plain TH5 MW1 NS4 NW4 KC16 (base)", h5_plain, 256}, {"plain KC32", h5_plain_kc32, 256},
                            {"plain NS2 NW8", h5_plain_nw8, 256}, {"plain MW2 NS2 NT128", h5_plain_mw2, 128},
                            {"interleaved (shipped)", h5_p3, 256}, {"interleaved NT128", h5_p3_nt128, 128},
                            {"interleaved NT64", h5_p3_nt64, 64}, {"interleaved NT128, 3 waves/SIMD", h5_w3_nt128, 128},
                            {"interleaved NT64, 4 waves/SIMD", h5_w4_nt64, 64}, {"interleaved NT256 forced 3 waves/SIMD", h5_w3_nt256, 256}, {"deep prefetch", h5_p4, 256}, {"glds weights", h5_p5, 256}, {"glds weights NT128", h5_p5_nt128, 128}, {"deep prefetch NT128", h5_p4_nt128, 128},
                            {"weights in registers (BREG)", h5_p6, 256}, {"BREG NT128", h5_p6_nt128, 128},
                            {"deep3 prefetch", h5_p7, 256}, {"deep3 prefetch NT128", h5_p7_nt128, 128}, {"interleaved MW2 NS2 NT128", h5_p3_mw2, 128}, {"deep MW2 NS2 NT128", h5_p4_mw2, 128},
                            {"ABL no global loads", h5_a1, 256}, {"ABL no LDS writes (loads die too)", h5_a2, 256},
                            {"ABL no loads/writes", h5_a3, 256}, {"ABL no ds_read", h5_a4, 256}, {"ABL no barrier", h5_a8, 256},
                            {"ABL MFMA stream only", h5_a15, 256}, {"ABL loads waited at step end, no writes", h5_a18, 256}};
    else if (layer >= 5) vars = {{"plain TH10 MW1 NS2 NW4 (base)", h10_plain, 128}, {"plain TH5 NS4 NT256", h10_plain_th5, 256},
                            {"interleaved (shipped)", h10_p3, 128}, {"interleaved TH5 NS4 NT256", h10_p3_th5, 256},
                            {"deep prefetch", h10_p4, 128}, {"glds weights", h10_p5, 128}, {"weights in registers (BREG)", h10_p6, 128}, {"deep3 prefetch", h10_p7, 128}, {"BREG TH5 NS4 NT256", h10_p6_th5, 256}, {"interleaved TH5 NS2 NT128, 3 waves/SIMD", h10_w3, 128}};
    else if (layer == 3) vars = {{"plain TH4 MW2 NS2 NW4 (base)", h20_plain, 128}, {"interleaved (shipped)", h20_p3, 128},
                            {"interleaved TH10 MW1", h20_p3_th10, 128}, {"deep prefetch", h20_p4, 128}, {"weights in registers (BREG)", h20_p6, 128}, {"deep3 prefetch", h20_p7, 128},
                            {"interleaved, 3 waves/SIMD", h20_w3_ns2, 128}, {"interleaved NS1 NT64, 3 waves/SIMD", h20_w3, 64}};
    else if (layer == 2 || layer == 4) vars = {{"plain TH4 MW2 NS2 NW4 NT128 (base)", p22_plain, 128}, {"plain TH4 MW4 NS1 NT64", p22_plain_nt64, 64},
                            {"interleaved NT128 (conv4 shipped)", p22_p3, 128}, {"interleaved NT64 (conv2 shipped)", p22_p3_nt64, 64},
                            {"interleaved TH10 MW1", p22_p3_th10, 128}, {"deep prefetch NT128", p22_p4, 128},
                            {"deep prefetch NT64", p22_p4_nt64, 64}, {"glds NT128", p22_p5, 128}, {"glds NT64", p22_p5_nt64, 64},
                            {"BREG NT128", p22_p6, 128}, {"BREG NT64", p22_p6_nt64, 64}, {"deep3 NT128", p22_p7, 128}, {"deep3 NT64", p22_p7_nt64, 64},
                            {"interleaved NT64, 3 waves/SIMD", p22_w3_nt64, 64}, {"interleaved NT128, 3 waves/SIMD", p22_w3_nt128, 128},
                            {"interleaved MW2 NT64, 3 waves/SIMD", p22_w3_mw2_nt64, 64}, {"interleaved MW2 NT64, 4 waves/SIMD", p22_w4_mw2_nt64, 64}};
    else { printf("layer %d not covered\n", layer); return 1; }

    const size_t xin = (size_t)n * s.H * s.W * s.cin;
    const int Hout = s.H / s.ph, Wout = s.W / s.pw;
    const size_t yout = (size_t)n * Hout * Wout * s.cout;
    const int cout16max = ((s.cout + 255) / 256 * 256) / 16;
    const size_t wsz = (size_t)9 * (s.cin / 16) * cout16max * 256;
    std::vector<float> hx(xin), hw(wsz), hb(cout16max * 16), hs(cout16max * 16), hh(cout16max * 16);
    unsigned r = 12345;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto &v : hx) v = rnd();
    for (auto &v : hb) v = 0.1f * rnd();
    for (auto &v : hs) v = 1.0f + 0.2f * rnd();
    for (auto &v : hh) v = 0.1f * rnd();
    float *dx, *dw, *db, *ds, *dh, *dy, *dy0;
    CK(hipMalloc(&dx, xin * 4)); CK(hipMalloc(&dw, wsz * 4)); CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMalloc(&ds, hb.size() * 4)); CK(hipMalloc(&dh, hb.size() * 4)); CK(hipMalloc(&dy, yout * 4)); CK(hipMalloc(&dy0, yout * 4));
    CK(hipMemcpy(dx, hx.data(), xin * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, hh.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * n * s.H * s.W * (double)s.cout * s.cin * 9;
    std::vector<float> y0(yout), y(yout);
    const float scale = 1.0f / sqrtf((float)s.cin * 9);
    printf("layer conv%d: %d->%d @%dx%d n=%d  %.1f GFLOP\n", layer, s.cin, s.cout, s.H, s.W, n, flops / 1e9);
    for (size_t vi = 0; vi < vars.size(); ++vi) {
        // weights in fragment order for this variant's cout16 (zero padded); logical weight = hash(co,ci,tap)
        const int cout16 = ((s.cout + vars[vi].nt - 1) / vars[vi].nt * vars[vi].nt) / 16;
        size_t o = 0;
        for (int tap = 0; tap < 9; ++tap)
            for (int g = 0; g < s.cin / 16; ++g)
                for (int sg = 0; sg < cout16; ++sg)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j, ++o) {
                            const int co = 16 * sg + (lane & 15), ci = 16 * g + 4 * (lane >> 4) + j;
                            unsigned h = (unsigned)(co * 73856093u) ^ (unsigned)(ci * 19349663u) ^ (unsigned)(tap * 83492791u);
                            h = h * 1664525u + 1013904223u;
                            hw[o] = co < s.cout ? (((h >> 8) & 0xffff) / 32768.0f - 1.0f) * scale : 0.f;
                        }
        CK(hipMemcpy(dw, hw.data(), o * 4, hipMemcpyHostToDevice));
        ConvArgs a{};
        a.x = dx; a.wfrag = dw; a.bias = db; a.bn_scale = ds; a.bn_shift = dh; a.y = vi == 0 ? dy0 : dy;
        a.n = n; a.H = s.H; a.W = s.W; a.Ho = s.H; a.Wo = s.W; a.cin = s.cin; a.cout16 = cout16; a.cout_valid = s.cout; a.out_stride = s.cout;
        CK(hipMemsetAsync(a.y, 0, yout * 4, st));
        vars[vi].fn(a, st);
        CK(hipStreamSynchronize(st));
        float best = 1e30f, sum = 0;
        const int reps = 5;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipEventRecord(e0, st));
            vars[vi].fn(a, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        CK(hipGetLastError());
        double maxdiff = 0;
        if (vi == 0) CK(hipMemcpy(y0.data(), dy0, yout * 4, hipMemcpyDeviceToHost));
        else {
            CK(hipMemcpy(y.data(), dy, yout * 4, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < yout; ++k) { double d = fabs((double)y[k] - y0[k]); if (d > maxdiff) maxdiff = d; }
        }
        printf("  %-36s avg %.3f ms  best %.3f ms  %.1f TF (best %.1f)  maxdiff_vs_base %.2e\n", vars[vi].name, sum / reps, best,
               flops / (sum / reps * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12, maxdiff);
    }
    return 0;
}
