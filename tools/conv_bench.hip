// conv_bench.hip — within-probe A/B of conv_igemm_kernel tile configurations on one layer shape.
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

#define VAR(NAME, TH, MW, NS, NW, KC, PH, PW, ACT, BN)                                                             \
    static void NAME(ConvArgs a, hipStream_t st) {                                                                 \
        launch(conv_igemm_kernel<3, 3, 1, 1, TH, MW, NS, NW, KC, PH, PW, ACT, BN, STAGE_F32_NHWC>, TH, 16 * MW,     \
               NS * NW * 16, NW * 64, a, st);                                                                      \
    }

#define VARP(NAME, TH, MW, NS, NW, KC, PH, PW, ACT, BN, PIPE)                                                      \
    static void NAME(ConvArgs a, hipStream_t st) {                                                                 \
        launch(conv_igemm_kernel<3, 3, 1, 1, TH, MW, NS, NW, KC, PH, PW, ACT, BN, STAGE_F32_NHWC, PIPE>, TH,        \
               16 * MW, NS * NW * 16, NW * 64, a, st);                                                             \
    }
VARP(h5_pipe1,     5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 1)
VARP(h5_pipe1_nt128, 5, 1, 2, 4, 16, 1, 1, ACT_LEAKY, true, 1)
VARP(h10_pipe1,    10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, 1)
VARP(h10_pipe1_th5, 5, 1, 4, 4, 16, 1, 1, ACT_RELU, false, 1)
VARP(h20_pipe1,    4, 2, 2, 4, 16, 1, 1, ACT_RELU, false, 1)
VARP(p22_pipe1,    4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, 1)
VARP(p22_pipe1_nt64, 4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, 1)
VARP(h5_pipe2,     5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 2)
VARP(h10_pipe2,    10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, 2)
VARP(p22_pipe2,    4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, 2)
VARP(p22_pipe2_nt64, 4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, 2)
VARP(h5_abl1,     5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 9)
VARP(h5_abl2,     5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 10)
VARP(h5_abl3,     5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 11)
VARP(h5_abl4,     5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 12)
VARP(h5_abl5,     5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 13)
VARP(h5_abl16,    5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 24)
VARP(h5_abl32,    5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 40)
VARP(h5_abl7,     5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 15)
VARP(h5_pipe3,     5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 3)
VARP(h5_pipe4,     5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, 4)
VARP(h5_pipe4_nt128, 5, 1, 2, 4, 16, 1, 1, ACT_LEAKY, true, 4)
VARP(h10_pipe4,    10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, 4)
VARP(p22_pipe4,    4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, 4)
VARP(p22_pipe4_nt64, 4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, 4)
VARP(h20_pipe4,    4, 2, 2, 4, 16, 1, 1, ACT_RELU, false, 4)
VARP(h5_pipe3_nt128, 5, 1, 2, 4, 16, 1, 1, ACT_LEAKY, true, 3)
VARP(h5_pipe3_nt64,  5, 1, 1, 4, 16, 1, 1, ACT_LEAKY, true, 3)
VARP(h5_pipe3_nw8,   5, 1, 1, 8, 16, 1, 1, ACT_LEAKY, true, 3)
VARP(h10_pipe3,    10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, 3)
VARP(h20_pipe3,    4, 2, 2, 4, 16, 1, 1, ACT_RELU, false, 3)
VARP(p22_pipe3,    4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, 3)
VARP(p22_pipe3_nt64, 4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, 3)
#define VARA(NAME, ABL)                                                                                              \
    static void NAME(ConvArgs a, hipStream_t st) {                                                                 \
        launch(conv_igemm_kernel<3, 3, 1, 1, 5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, STAGE_F32_NHWC, 3, ABL>, 5, 16, 256, 256, a, st); \
    }
VARA(p3_a1, 1) VARA(p3_a2, 2) VARA(p3_a3, 3) VARA(p3_a4, 4) VARA(p3_a7, 7) VARA(p3_a8, 8) VARA(p3_a15, 15) VARA(p3_a18, 18)
// conv8/9-shaped (H=5): pool none
VAR(h5_base,      5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true)
VAR(h5_kc32_nt128,5, 1, 2, 4, 32, 1, 1, ACT_LEAKY, true)
VAR(h5_nw8,       5, 1, 2, 8, 16, 1, 1, ACT_LEAKY, true)
VAR(h5_nw8_kc32,  5, 1, 2, 8, 32, 1, 1, ACT_LEAKY, true)
VAR(h5_nt128,     5, 1, 2, 4, 16, 1, 1, ACT_LEAKY, true)
VAR(h5_mw2_nt128, 5, 2, 2, 4, 16, 1, 1, ACT_LEAKY, true)
VAR(h5_kc32,      5, 1, 4, 4, 32, 1, 1, ACT_LEAKY, true)
// conv5/6-shaped (H=10)
VAR(h10_base,     10, 1, 2, 4, 16, 1, 1, ACT_RELU, false)
VAR(h10_th5_ns4,  5, 1, 4, 4, 16, 1, 1, ACT_RELU, false)
VAR(h10_nw8,      10, 1, 1, 8, 16, 1, 1, ACT_RELU, false)
VAR(h10_kc32,     10, 1, 2, 4, 32, 1, 1, ACT_RELU, false)
VAR(h10_th5_mw2,  5, 2, 2, 4, 16, 1, 1, ACT_RELU, false)
// conv3-shaped (H=20, no pool) and conv2/4 (pool 2x2)
VAR(h20_base,     4, 2, 2, 4, 16, 1, 1, ACT_RELU, false)
VAR(h20_th10,     10, 1, 2, 4, 16, 1, 1, ACT_RELU, false)
VAR(h20_th5_ns4,  5, 1, 4, 4, 16, 1, 1, ACT_RELU, false)
VAR(h20_th4mw4ns1,4, 4, 1, 4, 16, 1, 1, ACT_RELU, false)
VAR(p22_base,     4, 2, 2, 4, 16, 2, 2, ACT_RELU, false)
VAR(p22_th10,     10, 1, 2, 4, 16, 2, 2, ACT_RELU, false)
VAR(p22_th4mw4,   4, 4, 1, 4, 16, 2, 2, ACT_RELU, false)
VAR(p22_th2mw4ns2,2, 4, 2, 4, 16, 2, 2, ACT_RELU, false)
VAR(p22_kc32,     4, 2, 2, 4, 32, 2, 2, ACT_RELU, false)

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
    if (layer >= 8) vars = {{"TH5 MW1 NS4 NW4 KC16 (base)", h5_base, 256}, {"TH5 MW1 NS2 NW4 KC32 NT128", h5_kc32_nt128, 128},
                            {"TH5 MW1 NS2 NW8 KC16", h5_nw8, 256}, {"TH5 MW1 NS2 NW8 KC32", h5_nw8_kc32, 256},
                            {"TH5 MW1 NS2 NW4 KC16 NT128", h5_nt128, 128}, {"TH5 MW2 NS2 NW4 KC16 NT128", h5_mw2_nt128, 128},
                            {"TH5 MW1 NS4 NW4 KC32", h5_kc32, 256},
                            {"PIPE1 TH5 MW1 NS4 NW4 KC16", h5_pipe1, 256}, {"PIPE1 TH5 MW1 NS2 NW4 NT128", h5_pipe1_nt128, 128},
                            {"PIPE2 TH5 MW1 NS4 NW4", h5_pipe2, 256}, {"PIPE3 TH5 MW1 NS4 NW4", h5_pipe3, 256}, {"PIPE4 TH5 MW1 NS4 NW4", h5_pipe4, 256}, {"PIPE4 TH5 MW1 NS2 NW4 NT128", h5_pipe4_nt128, 128}, {"PIPE3 TH5 MW1 NS2 NW4 NT128", h5_pipe3_nt128, 128},
                            {"PIPE3 TH5 MW1 NS1 NW4 NT64", h5_pipe3_nt64, 64}, {"PIPE3 TH5 MW1 NS1 NW8 NT128", h5_pipe3_nw8, 128},
                            {"P3ABL no-loads", p3_a1, 256}, {"P3ABL no-stores", p3_a2, 256}, {"P3ABL no-loads/stores", p3_a3, 256},
                            {"P3ABL no-ds_read", p3_a4, 256}, {"P3ABL no-ld/st/read", p3_a7, 256}, {"P3ABL no-barrier", p3_a8, 256},
                            {"P3ABL mfma only", p3_a15, 256}, {"P3ABL loads, waited at step end, no stores", p3_a18, 256},
                            {"ABL no-loads/stores", h5_abl1, 256}, {"ABL no-barrier", h5_abl2, 256}, {"ABL no-loads no-barrier", h5_abl3, 256},
                            {"ABL no-ds_read", h5_abl4, 256}, {"ABL no-loads no-ds_read", h5_abl5, 256}, {"ABL mfma only", h5_abl7, 256},
                            {"ABL stores-no-loads", h5_abl16, 256}, {"ABL loads-no-stores", h5_abl32, 256}};
    else if (layer == 5 || layer == 6) vars = {{"TH10 MW1 NS2 NW4 KC16 (base)", h10_base, 128}, {"TH5 MW1 NS4 NW4 KC16", h10_th5_ns4, 256},
                            {"TH10 MW1 NS1 NW8 KC16", h10_nw8, 128}, {"TH10 MW1 NS2 NW4 KC32", h10_kc32, 128},
                            {"TH5 MW2 NS2 NW4 KC16", h10_th5_mw2, 128},
                            {"PIPE1 TH10 MW1 NS2 NW4", h10_pipe1, 128}, {"PIPE1 TH5 MW1 NS4 NW4", h10_pipe1_th5, 256},
                            {"PIPE2 TH10 MW1 NS2 NW4", h10_pipe2, 128}, {"PIPE3 TH10 MW1 NS2 NW4", h10_pipe3, 128}, {"PIPE4 TH10 MW1 NS2 NW4", h10_pipe4, 128}};
    else if (layer == 3) vars = {{"TH4 MW2 NS2 NW4 KC16 (base)", h20_base, 128}, {"TH10 MW1 NS2 NW4", h20_th10, 128},
                            {"TH5 MW1 NS4 NW4 (NT256: cout pad)", h20_th5_ns4, 256}, {"TH4 MW4 NS1 NW4 NT64", h20_th4mw4ns1, 64},
                            {"PIPE1 TH4 MW2 NS2 NW4", h20_pipe1, 128}, {"PIPE3 TH4 MW2 NS2 NW4", h20_pipe3, 128}, {"PIPE4 TH4 MW2 NS2 NW4", h20_pipe4, 128}};
    else if (layer == 2 || layer == 4) vars = {{"TH4 MW2 NS2 NW4 KC16 NT128", p22_base, 128}, {"TH10 MW1 NS2 NW4", p22_th10, 128},
                            {"TH4 MW4 NS1 NW4 NT64", p22_th4mw4, 64}, {"TH2 MW4 NS2 NW4 NT128", p22_th2mw4ns2, 128},
                            {"TH4 MW2 NS2 NW4 KC32", p22_kc32, 128},
                            {"PIPE1 TH4 MW2 NS2 NW4 NT128", p22_pipe1, 128}, {"PIPE1 TH4 MW4 NS1 NW4 NT64", p22_pipe1_nt64, 64},
                            {"PIPE2 TH4 MW2 NS2 NW4 NT128", p22_pipe2, 128}, {"PIPE2 TH4 MW4 NS1 NW4 NT64", p22_pipe2_nt64, 64},
                            {"PIPE3 TH4 MW2 NS2 NW4 NT128", p22_pipe3, 128}, {"PIPE3 TH4 MW4 NS1 NW4 NT64", p22_pipe3_nt64, 64},
                            {"PIPE4 TH4 MW2 NS2 NW4 NT128", p22_pipe4, 128}, {"PIPE4 TH4 MW4 NS1 NW4 NT64", p22_pipe4_nt64, 64}};
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
