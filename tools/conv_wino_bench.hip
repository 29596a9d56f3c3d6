// conv_wino_bench.hip — the Winograd F(2,3) kernel (csrc/conv_wino.hpp) against the shipped direct f16x2 / P2 kernel on one layer:
// time of both, max |difference| between them, and each against a float64 CPU reference on sampled outputs.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/conv_wino_bench tools/conv_wino_bench.hip
// Run  : tools/bin/conv_wino_bench <layer 3..9> [n_lines=256] [w_pad=576] [sustained launches=0] [xcd_g=1]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../pero_ocr_amd/csrc/conv_igemm.hpp"
#include "../pero_ocr_amd/csrc/conv_bf16x3.hpp"
#include "../pero_ocr_amd/csrc/conv_rows.hpp"
#include "experiments/conv_wino.hpp"
using namespace pocr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
struct Shape { int cin, cout, H, W, ph, pw, act; bool bn; };
template <class K>
static void launch(K kern, int TH, int TW, int NT, ConvArgs a, hipStream_t st) {
    a.tiles_w = (a.Wo + TW - 1) / TW; a.tiles_h = (a.Ho + TH - 1) / TH; a.tiles_n = (a.cout16 * 16) / NT;
    hipLaunchKernelGGL(kern, dim3((unsigned)conv_grid_blocks(a)), dim3(256), 0, st, a);
}
#define VP(NAME, TH, MW, NS, WM, PH, PW, ACT, BN, MINW, BDIR) static void NAME(ConvArgs a, hipStream_t st) { \
    launch(conv3x3_bf16x3_kernel<TH, MW, NS, WM, PH, PW, ACT, BN, MINW, BDIR, 3, 3, 1, 1, false, 2, true, true>, TH, 16 * MW, NS * (4 / WM) * 16, a, st); }
// the shipped P2 configurations (pocr_hip.hip POCR_CONVP)
VP(d9, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2, true)
VP(d8, 5, 1, 2, 1, 1, 1, ACT_LEAKY, false, 2, true)
VP(d7, 10, 1, 1, 1, 2, 1, ACT_RELU, false, 2, true)
VP(d56, 10, 1, 1, 1, 1, 1, ACT_RELU, false, 2, true)
VP(d4, 10, 1, 1, 1, 2, 2, ACT_RELU, false, 2, true)
VP(d3, 10, 1, 1, 1, 1, 1, ACT_RELU, false, 2, true)
// the persistent form of the shipped configurations (conv_rows.hpp); POCR_ROWS_WGS = workgroups per CU (default 2)
template <class K>
static void launch_rows(K kern, int TH, int TW, int NT, ConvArgs a, hipStream_t st) {
    a.tiles_w = (a.Wo + TW - 1) / TW; a.tiles_h = (a.Ho + TH - 1) / TH; a.tiles_n = (a.cout16 * 16) / NT;
    const size_t blocks = conv_grid_blocks(a);
    static const int wgs = getenv("POCR_ROWS_WGS") ? atoi(getenv("POCR_ROWS_WGS")) : 2;
    size_t grid = blocks;
    if (wgs > 0 && (size_t)(256 * wgs) < blocks) grid = 256 * wgs;
    a.nblocks = (int)blocks;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, st, a);
}
#define RP(NAME, TH, MW, NS, WM, PH, PW, ACT, BN, MINW) static void NAME(ConvArgs a, hipStream_t st) { \
    launch_rows(conv3x3_rows_kernel<TH, MW, NS, WM, PH, PW, ACT, BN, MINW, true>, TH, 16 * MW, NS * (4 / WM) * 16, a, st); }
RP(r9, 5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2)
RP(r8, 5, 1, 2, 1, 1, 1, ACT_LEAKY, false, 2)
RP(r7, 10, 1, 1, 1, 2, 1, ACT_RELU, false, 2)
RP(r56, 10, 1, 1, 1, 1, 1, ACT_RELU, false, 2)
RP(r4, 10, 1, 1, 1, 2, 2, ACT_RELU, false, 2)
RP(r3, 10, 1, 1, 1, 1, 1, ACT_RELU, false, 2)
// experiment: 5 x 16 x 64 tiles, three workgroups per CU (POCR_ROWS_ALT=1)
RP(r56_5, 5, 1, 1, 1, 1, 1, ACT_RELU, false, 3)
RP(r4_4, 4, 1, 1, 1, 2, 2, ACT_RELU, false, 3)
// experiment: the 5-row layers on 5 x 32 x 64 tiles (a wave: ten 16-pixel strips x 16 channels, like the 10 x 16 x 64 tile) (POCR_ROWS_ALT=2)
RP(r9_w, 5, 2, 1, 1, 1, 1, ACT_LEAKY, true, 2)
RP(r8_w, 5, 2, 1, 1, 1, 1, ACT_LEAKY, false, 2)
// experiment: the direct kernel with its waves splitting PIXELS, weights shared through LDS (fewer bytes from L2 per output)
VP(x9_22, 5, 2, 2, 2, 1, 1, ACT_LEAKY, true, 2, false)     // 5x32 px x 64 ch, waves 2 (px) x 2 (ch)
VP(x9_41, 5, 4, 2, 4, 1, 1, ACT_LEAKY, true, 2, false)     // 5x64 px x 32 ch, waves 4 (px)
VP(x9_42, 5, 4, 4, 4, 1, 1, ACT_LEAKY, true, 1, false)     // 5x64 px x 64 ch, waves 4 (px), NS 4: one workgroup per CU
VP(x9_24, 5, 2, 4, 2, 1, 1, ACT_LEAKY, true, 1, false)     // 5x32 px x 128 ch, waves 2 x 2, NS 4
// ... and with the weights straight from L2 (row-streaming loop): the waves that share channels request the SAME fragments - L1 serves the second
VP(x9_r22, 5, 2, 2, 2, 1, 1, ACT_LEAKY, true, 2, true)     // 5x32 px x 64 ch, waves 2 (px) x 2 (ch)
VP(x9_r41, 5, 4, 2, 4, 1, 1, ACT_LEAKY, true, 2, true)     // 5x64 px x 32 ch, waves 4 (px)
template <int TH, int PH, int PW, int ACT, bool BN>
static void wino(WinoArgs a, hipStream_t st) {
    a.tiles_n = (a.cout16 * 16) / 64;
    hipLaunchKernelGGL((conv3x3_wino_kernel<TH, PH, PW, ACT, BN>), dim3((unsigned)wino_grid_blocks(a)), dim3(512), 0, st, a);
}
static uint16_t f16bits(_Float16 v) { uint16_t b; memcpy(&b, &v, 2); return b; }
static float f16val(uint16_t b) { _Float16 v; memcpy(&v, &b, 2); return (float)v; }

#ifdef POCR_BF16X3_TRACE
#include <algorithm>
__global__ void trace_copy_kernel(unsigned long long *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { out[i] = g_conv_trace[i]; g_conv_trace[i] = 0; }
}
static void trace_report(bool rows = false) {
    const size_t nb = 1u << 15;
    std::vector<unsigned long long> t(nb * 8);
    unsigned long long *dp;
    CK(hipMalloc(&dp, nb * 8 * sizeof(*dp)));
    hipLaunchKernelGGL(trace_copy_kernel, dim3(256), dim3(256), 0, 0, dp, nb * 8);
    CK(hipMemcpy(t.data(), dp, nb * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    CK(hipFree(dp));
    if (rows) {      // conv_rows.hpp: per workgroup [0] start, [1] sum of main loops, [2] sum of epilogues, [3] sum of everything else, [4] end, [6] tiles
        double tm = 0, te = 0, to = 0, tiles = 0, span = 0; size_t ran = 0;
        for (size_t b = 0; b < nb; ++b) { const unsigned long long *q = &t[b * 8]; if (!q[4]) continue; ++ran; tm += q[1]; te += q[2]; to += q[3]; tiles += q[6]; span += q[4] - q[0]; }
        if (!ran) { printf("    trace: no stamps\n"); return; }
        printf("    trace (%zu workgroups, %.1f tiles each, %.1f us each): mean per tile: main loop %.2f us, epilogue %.2f us, between tiles %.2f us\n",
               ran, tiles / ran, span / ran * 0.01, tm / tiles * 0.01, te / tiles * 0.01, to / tiles * 0.01);
        return;
    }
    unsigned long long tmin = ~0ull, tmax = 0; double ph[4] = {0, 0, 0, 0}; size_t ran = 0; double lat = 0, latn = 0;
    for (size_t b = 0; b < nb; ++b) {
        const unsigned long long *q = &t[b * 8];
        if (!q[4]) continue;
        ++ran; tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[4]);
        for (int k = 0; k < 4; ++k) ph[k] += (double)(q[k + 1] - q[k]);
        lat += (double)q[6]; latn += (double)q[7];
    }
    if (!ran) { printf("    trace: no stamps\n"); return; }
    printf("    trace (first %zu workgroups): span %.1f us; mean per workgroup: prologue %.2f us, main loop %.2f us, epilogue issue %.2f us, store drain %.2f us\n",
           ran, (tmax - tmin) * 0.01, ph[0] / ran * 0.01, ph[1] / ran * 0.01, ph[2] / ran * 0.01, ph[3] / ran * 0.01);
    if (latn > 0) printf("    activation-load bursts: mean issue-to-data %.2f us over %.0f bursts\n", lat / latn * 0.01, latn);
    // Do the workgroups that share a CU run in phase?  Per CU (XCC id, HW_ID bits 8..15): starts sorted in time; for every workgroup the
    // distance from its start to the latest start of ANOTHER workgroup on that CU, as a fraction of its own duration (0 = started
    // together: their VALU phases - prologue, epilogue - coincide and cannot hide behind each other's MFMA phase; 0.5 = alternating)
    {
        std::vector<std::pair<unsigned, size_t>> key;
        for (size_t b = 0; b < nb; ++b) if (t[b * 8 + 4]) key.push_back({(unsigned)((t[b * 8 + 5] >> 32) << 8 | ((t[b * 8 + 5] >> 8) & 0xff)), b});
        std::sort(key.begin(), key.end(), [&](const auto &x, const auto &y) { return x.first != y.first ? x.first < y.first : t[x.second * 8] < t[y.second * 8]; });
        size_t hist[10] = {0}, ncu = 0, cnt = 0; double both_main = 0, any = 0;
        for (size_t i = 0; i < key.size();) {
            size_t j = i; while (j < key.size() && key[j].first == key[i].first) ++j;
            ++ncu;
            for (size_t k = i + 1; k < j; ++k) {
                const unsigned long long *q = &t[key[k].second * 8], *pq = &t[key[k - 1].second * 8];
                const double dur = (double)(q[3] - q[0]);
                if (pq[3] <= q[0] || dur <= 0) continue;            // the previous one had ended: not co-resident
                const double f = (double)(q[0] - pq[0]) / dur;
                ++hist[std::min(9, (int)(f * 10))]; ++cnt;
                // time both are inside their main loops, relative to this one's main loop
                const double lo = (double)std::max(q[1], pq[1]), hi = (double)std::min(q[2], pq[2]);
                both_main += std::max(0.0, hi - lo); any += (double)(q[2] - q[1]);
            }
            i = j;
        }
        printf("    phase of co-resident workgroups (%zu CUs, %zu pairs): start offset / own duration, deciles:", ncu, cnt);
        for (int k = 0; k < 10; ++k) printf(" %.2f", cnt ? (double)hist[k] / cnt : 0.0);
        printf("; share of a main loop spent next to the neighbour's main loop %.2f\n", any > 0 ? both_main / any : 0.0);
    }
}
#endif
int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int layer = argc > 1 ? atoi(argv[1]) : 9, n = argc > 2 ? atoi(argv[2]) : 256, wpad = argc > 3 ? atoi(argv[3]) : 576;
    const int sustained = argc > 4 ? atoi(argv[4]) : 0, xcd_g = argc > 5 ? atoi(argv[5]) : 1;
    Shape shapes[10] = {{}, {}, {64, 64, 40, wpad, 2, 2, ACT_RELU, false}, {64, 128, 20, wpad / 2, 1, 1, ACT_RELU, false},
                        {128, 128, 20, wpad / 2, 2, 2, ACT_RELU, false}, {128, 256, 10, wpad / 4, 1, 1, ACT_RELU, false},
                        {256, 256, 10, wpad / 4, 1, 1, ACT_RELU, false}, {256, 256, 10, wpad / 4, 2, 1, ACT_RELU, false},
                        {256, 512, 5, wpad / 4, 1, 1, ACT_LEAKY, false}, {512, 512, 5, wpad / 4, 1, 1, ACT_LEAKY, true}};
    if (layer < 3 || layer > 9) { printf("layer %d not covered\n", layer); return 1; }
    const Shape s = shapes[layer];
    const size_t xin = (size_t)n * s.H * s.W * s.cin;
    const int Hout = s.H / s.ph, Wout = s.W / s.pw;
    const size_t yout = (size_t)n * Hout * Wout * s.cout;
    std::vector<float> hx(xin), hb(s.cout), hs(s.cout), hh(s.cout);
    unsigned r = 12345;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto &v : hx) { v = rnd() * (1.0f + 0.001f * rnd()); if (v < 0 && layer > 3) v *= 0.05f; }      // mostly non-negative, like activations behind a ReLU
    for (auto &v : hb) v = 0.1f * rnd();
    for (auto &v : hs) v = 1.0f + 0.2f * rnd();
    for (auto &v : hh) v = 0.1f * rnd();
    const float scale = 1.0f / sqrtf((float)s.cin * 9);
    auto W = [&](int co, int ci, int tap) {
        unsigned h = (unsigned)(co * 73856093u) ^ (unsigned)(ci * 19349663u) ^ (unsigned)(tap * 83492791u);
        h = h * 1664525u + 1013904223u;
        float v = (((h >> 8) & 0xffff) / 32768.0f - 1.0f) * scale;
        h = h * 1664525u + 1013904223u;
        return v * (1.0f + 1e-3f * (((h >> 8) & 0xffff) / 32768.0f - 1.0f));
    };
    // the operands as f16x2 represents them (what both kernels really multiply): the float64 reference uses the fp32 values
    std::vector<uint16_t> xp2(xin * 2);
    for (size_t px = 0; px < xin / s.cin; ++px)
        for (int c = 0; c < s.cin; ++c) {
            const float v = hx[px * s.cin + c];
            const _Float16 h = (_Float16)v, l = (_Float16)((v - (float)h) * 2048.0f);
            const size_t o = px * s.cin * 2 + (size_t)(c >> 5) * 64 + (c & 31);
            xp2[o] = f16bits(h); xp2[o + 32] = f16bits(l);
        }
#if POCR_WINO_CM_IN
    std::vector<uint16_t> xcm(xin * 2);                  // [line][row][chunk][pixel][h 32 | l 32]
    for (size_t rowi = 0; rowi < (size_t)n * s.H; ++rowi)
        for (int px = 0; px < s.W; ++px)
            for (int g = 0; g < s.cin / 32; ++g)
                memcpy(&xcm[((rowi * (s.cin / 32) + g) * s.W + px) * 64], &xp2[((rowi * s.W + px) * (s.cin / 32) + g) * 64], 128);
    float *dxcm; CK(hipMalloc(&dxcm, xin * 4)); CK(hipMemcpy(dxcm, xcm.data(), xin * 4, hipMemcpyHostToDevice));
#endif
    const int cout16 = s.cout / 16;
    auto build = [&](int ntaps, auto &&wf) {              // wsplit[tap][cin/32][cout16][plane][lane][8]
        std::vector<uint16_t> hw((size_t)ntaps * (s.cin / 32) * cout16 * 2 * 64 * 8);
        size_t o = 0;
        for (int tap = 0; tap < ntaps; ++tap) for (int g = 0; g < s.cin / 32; ++g) for (int sg = 0; sg < cout16; ++sg)
            for (int pl = 0; pl < 2; ++pl) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j, ++o) {
                const int co = 16 * sg + (lane & 15), ci = 32 * g + 8 * (lane >> 4) + j;
                const double w = wf(co, ci, tap);
                const _Float16 h = (_Float16)w, l = (_Float16)((w - (double)h) * 2048.0);
                hw[o] = f16bits(pl ? l : h);
            }
        return hw;
    };
    const std::vector<uint16_t> wdir = build(9, [&](int co, int ci, int tap) { return (double)W(co, ci, tap); });
    const std::vector<uint16_t> wwin = build(12, [&](int co, int ci, int t) {
        const int kk = t / 3, dy = t % 3;
        const double g0 = W(co, ci, dy * 3), g1 = W(co, ci, dy * 3 + 1), g2 = W(co, ci, dy * 3 + 2);
        return kk == 0 ? g0 : kk == 1 ? (g0 + g1 + g2) * 0.5 : kk == 2 ? (g0 - g1 + g2) * 0.5 : g2;
    });
    float *dx, *dw, *dw2, *db, *ds, *dh, *dy, *dy2;
    CK(hipMalloc(&dx, xin * 4)); CK(hipMalloc(&dw, wdir.size() * 2)); CK(hipMalloc(&dw2, wwin.size() * 2)); CK(hipMalloc(&db, s.cout * 4));
    CK(hipMalloc(&ds, s.cout * 4)); CK(hipMalloc(&dh, s.cout * 4)); CK(hipMalloc(&dy, yout * 4)); CK(hipMalloc(&dy2, yout * 4));
    CK(hipMemcpy(dx, xp2.data(), xin * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, wdir.data(), wdir.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw2, wwin.data(), wwin.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), s.cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, hs.data(), s.cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, hh.data(), s.cout * 4, hipMemcpyHostToDevice));
    // Winograd tiles: TH rows x two half tiles of 16 columns; half tiles of all lines packed two by two (row tile outermost)
    const int TH = (layer == 4 || layer == 7) ? 4 : 5;
    std::vector<WinoTile> wt;
    {
        std::vector<std::pair<int, int>> halves;          // (line, (h0 << 16) | wt)
        for (int h0 = 0; h0 < s.H; h0 += TH)
            for (int ln = 0; ln < n; ++ln)
                for (int w8 = 0; w8 < (s.W + 15) / 16; ++w8) halves.push_back({ln, (h0 << 16) | w8});
        for (size_t i = 0; i < halves.size(); i += 2) {
            WinoTile t{{halves[i].first, -1}, {halves[i].second, 0}};
            if (i + 1 < halves.size()) { t.line[1] = halves[i + 1].first; t.ht_wt[1] = halves[i + 1].second; }
            wt.push_back(t);
        }
    }
    std::vector<int32_t> lw(n, s.W);
    std::vector<int64_t> ioff(n), ooff(n);
    for (int i = 0; i < n; ++i) { ioff[i] = (int64_t)i * s.H * s.W * s.cin; ooff[i] = (int64_t)i * Hout * Wout * s.cout; }
    WinoTile *dwt; int32_t *dlw; int64_t *dio, *doo;
    CK(hipMalloc(&dwt, wt.size() * sizeof(WinoTile))); CK(hipMalloc(&dlw, n * 4)); CK(hipMalloc(&dio, n * 8)); CK(hipMalloc(&doo, n * 8));
    CK(hipMemcpy(dwt, wt.data(), wt.size() * sizeof(WinoTile), hipMemcpyHostToDevice));
    CK(hipMemcpy(dlw, lw.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dio, ioff.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(doo, ooff.data(), n * 8, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * n * s.H * s.W * (double)s.cout * s.cin * 9;
    printf("layer conv%d: %d->%d @%dx%d n=%d  %.1f GFLOP (direct form), %zu Winograd tiles of %d rows\n", layer, s.cin, s.cout, s.H, s.W, n, flops / 1e9, wt.size(), TH);
    // float64 reference on sampled outputs (of the fp32 inputs / weights)
    const int NSAMP = 600;
    std::vector<size_t> samp(NSAMP);
    std::vector<double> ref(NSAMP);
    for (int k = 0; k < NSAMP; ++k) {
        r = r * 1664525u + 1013904223u;
        samp[k] = (size_t)(r % (unsigned)(yout / 97)) * 97 % yout;
        if (k < 64) {                                   // force samples at the line's right / left edge and in the last half tile
            size_t p = samp[k] / s.cout; const int co = samp[k] % s.cout;
            const int wo = (k & 1) ? Wout - 1 - (k >> 1) % 3 : (k >> 1) % 3;
            p = p / Wout * Wout + wo; samp[k] = p * s.cout + co;
        }
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
    std::vector<uint16_t> raw(yout * 2);
    std::vector<float> yd(yout), yw(yout);
    auto readback = [&](float *dev, std::vector<float> &out) {
        CK(hipMemcpy(raw.data(), dev, yout * 4, hipMemcpyDeviceToHost));
        for (size_t px = 0; px < yout / s.cout; ++px)
            for (int c = 0; c < s.cout; ++c) {
                const size_t o = px * s.cout * 2 + (size_t)(c >> 5) * 64 + (c & 31);
                out[px * s.cout + c] = f16val(raw[o]) + f16val(raw[o + 32]) * (1.0f / 2048.0f);
            }
    };
    const int nvar = layer == 9 ? 9 : 3;
    for (int vi = 0; vi < nvar; ++vi) {
        WinoArgs a{};
        a.x = dx; a.wfrag = vi == 1 ? dw2 : dw; a.bias = db; a.bn_scale = ds; a.bn_shift = dh; a.y = vi ? dy2 : dy;
        const bool is_w = vi == 1;
        a.n = n; a.H = s.H; a.W = s.W; a.Ho = s.H; a.Wo = s.W; a.cin = s.cin; a.cout16 = cout16; a.cout_valid = s.cout; a.out_stride = s.cout;
        a.xcd_g = xcd_g;
        #if POCR_WINO_CM_IN
        if (is_w) a.x = dxcm;
#endif
        if (is_w) { a.x_bytes = (uint32_t)(xin * 4); a.wtiles = dwt; a.n_ptiles = (int)wt.size(); a.line_w = dlw; a.in_off = dio; a.out_off = doo; }
        auto run = [&]() {
            static const int altv = getenv("POCR_ROWS_ALT") ? atoi(getenv("POCR_ROWS_ALT")) : 0;
            const bool alt = altv == 1;
            if (vi == 2 && altv == 2 && layer == 9) r9_w(a, st);
            else if (vi == 2 && altv == 2 && layer == 8) r8_w(a, st);
            else if (vi == 2 && alt && (layer == 3 || layer == 5 || layer == 6)) r56_5(a, st);
            else if (vi == 2 && alt && layer == 4) r4_4(a, st);
            else if (vi == 2) { switch (layer) { case 9: r9(a, st); break; case 8: r8(a, st); break; case 7: r7(a, st); break; case 6: case 5: r56(a, st); break; case 4: r4(a, st); break; default: r3(a, st); } }
            else if (vi >= 3) { switch (vi) { case 3: x9_22(a, st); break; case 4: x9_41(a, st); break; case 5: x9_42(a, st); break; case 6: x9_24(a, st); break; case 7: x9_r22(a, st); break; default: x9_r41(a, st); } }
            else if (!vi) {
                switch (layer) { case 9: d9(a, st); break; case 8: d8(a, st); break; case 7: d7(a, st); break; case 6: case 5: d56(a, st); break; case 4: d4(a, st); break; default: d3(a, st); }
            } else {
                switch (layer) {
                case 9: wino<5, 1, 1, ACT_LEAKY, true>(a, st); break;
                case 8: wino<5, 1, 1, ACT_LEAKY, false>(a, st); break;
                case 7: wino<4, 2, 1, ACT_RELU, false>(a, st); break;
                case 4: wino<4, 2, 2, ACT_RELU, false>(a, st); break;
                default: wino<5, 1, 1, ACT_RELU, false>(a, st);
                }
            }
        };
        CK(hipMemsetAsync(a.y, 0xff, yout * 4, st));
#ifdef POCR_BF16X3_TRACE
        { unsigned long long *dp; CK(hipMalloc(&dp, (size_t)(1u << 18) * 8)); hipLaunchKernelGGL(trace_copy_kernel, dim3(256), dim3(256), 0, st, dp, (size_t)1u << 18); CK(hipStreamSynchronize(st)); CK(hipFree(dp)); }   // (stamps of the previous variant's timing runs)
#endif
        run();
        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        readback(a.y, vi ? yw : yd);
#ifdef POCR_BF16X3_TRACE
        trace_report(vi == 2);
#endif
        for (int w = 0; w < 2; ++w) run();
        float best = 1e30f, sum = 0;
        const int reps = 10;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipEventRecord(e0, st)); run(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
        }
        float sus = 0;
        if (sustained > 0) {
            for (int rep = 0; rep < sustained / 3; ++rep) run();
            CK(hipEventRecord(e0, st));
            for (int rep = 0; rep < sustained; ++rep) run();
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&sus, e0, e1)); sus /= sustained;
        }
        const std::vector<float> &yy = vi ? yw : yd;
        double maxref = 0, rmsref = 0, maxdiff = 0; size_t nnan = 0;
        for (int k = 0; k < NSAMP; ++k) { const double d = fabs((double)yy[samp[k]] - ref[k]); maxref = d > maxref ? d : maxref; rmsref += d * d; }
        if (vi) for (size_t k = 0; k < yout; ++k) { const double d = fabs((double)yw[k] - yd[k]); if (!(d == d)) ++nnan; else if (d > maxdiff) maxdiff = d; }
        const char *vname[9] = {"direct f16x2 P2 (shipped)", "Winograd F(2,3) f16x2", "direct, persistent (conv_rows.hpp)", "direct 5x32x64 2x2 LDS weights", "direct 5x64x32 4x1 LDS weights", "direct 5x64x64 4x1 NS4 1WG", "direct 5x32x128 2x2 NS4 1WG",
                                "direct 5x32x64 2x2 rows, L2 weights", "direct 5x64x32 4x1 rows, L2 weights"};
        printf("  %-34s avg %.3f ms best %.3f ms  %.1f TF(alg)", vname[vi], sum / reps, best, flops / (sum / reps * 1e-3) / 1e12);
        if (sustained > 0) printf("  sustained x%d: %.3f ms %.1f TF", sustained, sus, flops / (sus * 1e-3) / 1e12);
        printf("  vs float64: max %.2e rms %.2e", maxref, sqrt(rmsref / NSAMP));
        if (vi) printf("  max|wino - direct| %.2e (NaN %zu)", maxdiff, nnan);
        printf("\n");
    }
    return 0;
}
