// parsenet_host.hpp — host orchestration + C ABI of the layout network (included by pocr_hip.hip, which provides
// DevBuf, build_wfrag, launch_conv and the conv tile configurations).  Replaces Net.__init__ / TorchParseNet.get_maps,
// pero_ocr/layout_engines/torch_parsenet.py:8-20, 37-58.  Topology and tensor order: pero_ocr_amd/parsenet_spec.py.

namespace {

// tile configurations the layout network adds to the recogniser's (same kernel, conv_igemm.hpp)
//                 KH KW P  P  TH MW NS NW KC PH PW
POCR_CONV(pn_pool256_k, 3, 3, 1, 1, 10, 1, 2, 4, 16, 2, 2, ACT_RELU, false, STAGE_F32_NHWC, PIPE_DEEP)        // 256->256 + pool 2x2
POCR_CONV(pn_up_small_k, 3, 3, 1, 1, 10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, STAGE_UPCAT, PIPE_DEEP)           // decoder, small maps (NT 128)
POCR_CONV(pn_up_mid_k,   3, 3, 1, 1, 4, 2, 2, 4, 16, 1, 1, ACT_RELU, false, STAGE_UPCAT, PIPE_INTERLEAVED)     // decoder @1/4 (NT 128)
POCR_CONV(pn_up_big_k,   3, 3, 1, 1, 4, 4, 1, 4, 16, 1, 1, ACT_RELU, false, STAGE_UPCAT, PIPE_DEEP)            // decoder @1/2, 1/1 (NT 64)

struct PnLayer { int cin, cout, nt; DevBuf w, b; int cout16; bool b3; };

}  // namespace

struct pocr_parsenet {
    int device = 0;
    hipStream_t stream = nullptr;
    PnLayer enc[13], dec[6];
    DevBuf head_w, head_b, lut, lines, tiles, wline, ooff;
    DevBuf page, small;                    // uint8 page as uploaded / after the area down-sampling
    DevBuf taps;                           // separable area weights of a fractional down-sampling (pocr_parsenet_get_maps_area)
    void *pin_taps = nullptr;
    size_t pin_taps_cap = 0;
    DevBuf x[7], p[7], y[6], out;          // skips x0..x5 + bottleneck x6, pooled maps p1..p6, decoder outputs y5..y0
    void *pin_in = nullptr, *pin_out = nullptr;
    size_t pin_in_cap = 0, pin_out_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    DevBuf range;                          // f16x2 range guard (conv_igemm.hpp): one set of 8 words per conv layer, 1 .. 18 (e1 .. d0)
    unsigned *range_host = nullptr;
    // a page on which a layer leaves f16's range is run again on bf16x3 (fp32's range) by a second network created on first use
    // from the retained weight blob (38 MB on the host; 77 MB of weights + its own activation buffers on the device, only then)
    std::vector<float> weights_host;
    pocr_parsenet *shadow = nullptr;
    int64_t range_fallbacks = 0;
};

namespace {

const int kPnEnc[13][3] = {{3, 64, 1}, {64, 64, 2}, {64, 128, 1}, {128, 128, 2}, {128, 256, 1}, {256, 256, 2}, {256, 256, 1},
                           {256, 256, 2}, {256, 256, 1}, {256, 256, 2}, {256, 256, 1}, {256, 256, 2}, {256, 256, 1}};
const int kPnDec[6][3] = {{256, 256, 256}, {256, 256, 256}, {256, 256, 256}, {256, 256, 128}, {128, 128, 64}, {64, 64, 64}};   // up, skip, out

size_t pn_num_floats() {
    size_t t = 0;
    for (auto &l : kPnEnc) t += (size_t)l[1] * l[0] * 9 + l[1];
    for (auto &l : kPnDec) t += (size_t)l[2] * (l[0] + l[1]) * 9 + l[2];
    return t + 5 * 64 + 5;
}

int cv_round_div(int a, int b) {           // cvRound(a / (double)b): round half to even
    return (int)std::nearbyint((double)a / (double)b);
}

}  // namespace

extern "C" {

size_t pocr_parsenet_num_weight_floats(void) { return pn_num_floats(); }

void pocr_parsenet_destroy(pocr_parsenet *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)locked_device_sync();
    for (auto &l : p->enc) { l.w.release(); l.b.release(); }
    for (auto &l : p->dec) { l.w.release(); l.b.release(); }
    for (DevBuf *b : {&p->head_w, &p->head_b, &p->lut, &p->lines, &p->tiles, &p->wline, &p->ooff, &p->page, &p->small, &p->taps, &p->out}) b->release();
    if (p->pin_taps) (void)locked_host_free(p->pin_taps);
    for (auto &b : p->x) b.release();
    for (auto &b : p->p) b.release();
    for (auto &b : p->y) b.release();
    if (p->pin_in) (void)locked_host_free(p->pin_in);
    if (p->pin_out) (void)locked_host_free(p->pin_out);
    if (p->range_host) (void)locked_host_free(p->range_host);
    p->range.release();
    if (p->shadow) { pocr_parsenet_destroy(p->shadow); p->shadow = nullptr; }
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

int pocr_parsenet_create(const float *weights, size_t n_floats, int device_id, pocr_parsenet **out) {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (!weights) return fail("weights is NULL");
    if (n_floats != pn_num_floats()) return fail("weight blob has %zu floats, the layout network needs %zu", n_floats, pn_num_floats());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail("device_id %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail("device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    pocr_parsenet *p = new pocr_parsenet();
    p->device = device_id;
    auto bail = [&](int rc) { pocr_parsenet_destroy(p); return rc; };
    if (create_front_stream(&p->stream) != hipSuccess) return bail(fail("hipStreamCreate failed"));
    if (hipEventCreate(&p->ev0) != hipSuccess || hipEventCreate(&p->ev1) != hipSuccess) return bail(fail("hipEventCreate failed"));
    hipStream_t st = p->stream;
    WeightCursor cur{weights};
    const int enc_nt[13] = {64, 64, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};
    for (int i = 0; i < 13; ++i) {
        PnLayer &L = p->enc[i];
        L.cin = kPnEnc[i][0]; L.cout = kPnEnc[i][1]; L.nt = enc_nt[i];
        const float *w = cur.take((size_t)L.cout * L.cin * 9), *b = cur.take(L.cout);
        L.cout16 = round_up(L.cout, L.nt) / 16;
        L.b3 = i > 0 && conv_split() != 0;          // e0 (3 input channels, fused uint8 staging) stays on the fp32 kernel
        if (L.b3) {
            auto wsp = build_wsplit(9, L.cin, L.cout16, [&](int co, int ci, int tap) { return w[((size_t)co * L.cin + ci) * 9 + tap]; }, L.cout);
            std::vector<float> bias(L.cout16 * 16, 0.f);
            for (int k = 0; k < L.cout; ++k) bias[k] = b[k];
            if (upload_u16(L.w, wsp, st) || upload(L.b, bias, st)) return bail(1);
            continue;
        }
        std::vector<float> frag;
        if (i == 0) frag = build_wfrag(1, 32, L.cout16, [&](int co, int k, int) { const int tap = k / 3, c = k % 3; return w[((size_t)co * 3 + c) * 9 + tap]; }, 27, L.cout);
        else frag = build_wfrag(9, L.cin, L.cout16, [&](int co, int ci, int tap) { return w[((size_t)co * L.cin + ci) * 9 + tap]; }, L.cin, L.cout);
        std::vector<float> bias(L.cout16 * 16, 0.f);
        for (int k = 0; k < L.cout; ++k) bias[k] = b[k];
        if (upload(L.w, frag, st) || upload(L.b, bias, st)) return bail(1);
    }
    const int dec_nt[6] = {128, 128, 128, 128, 64, 64};
    for (int i = 0; i < 6; ++i) {
        PnLayer &L = p->dec[i];
        L.cin = kPnDec[i][0] + kPnDec[i][1]; L.cout = kPnDec[i][2]; L.nt = dec_nt[i];
        const float *w = cur.take((size_t)L.cout * L.cin * 9), *b = cur.take(L.cout);
        L.cout16 = round_up(L.cout, L.nt) / 16;
        L.b3 = conv_split() != 0;
        std::vector<float> bias(L.cout16 * 16, 0.f);
        for (int k = 0; k < L.cout; ++k) bias[k] = b[k];
        if (L.b3) {
            auto wsp = build_wsplit(9, L.cin, L.cout16, [&](int co, int ci, int tap) { return w[((size_t)co * L.cin + ci) * 9 + tap]; }, L.cout);
            if (upload_u16(L.w, wsp, st) || upload(L.b, bias, st)) return bail(1);
            continue;
        }
        auto frag = build_wfrag(9, L.cin, L.cout16, [&](int co, int ci, int tap) { return w[((size_t)co * L.cin + ci) * 9 + tap]; }, L.cin, L.cout);
        if (upload(L.w, frag, st) || upload(L.b, bias, st)) return bail(1);
    }
    {
        const float *w = cur.take(5 * 64), *b = cur.take(5);
        if (upload(p->head_w, std::vector<float>(w, w + 320), st) || upload(p->head_b, std::vector<float>(b, b + 5), st)) return bail(1);
        // uint8 -> float32 exactly as `tensor.float() * (1/255.)` (torch_parsenet.py:50): the Python double 1/255 is
        // rounded to float32 once, then a float32 multiply - NOT the recogniser's true division by 255
        std::vector<float> lut(256);
        const float inv = (float)(1.0 / 255.0);
        for (int i = 0; i < 256; ++i) lut[i] = (float)i * inv;
        if (upload(p->lut, lut, st)) return bail(1);
    }
    if (conv_split() == 2) p->weights_host.assign(weights, weights + n_floats);      // for the range guard's fall-back network
    *out = p;
    return 0;
}

int64_t pocr_parsenet_range_fallbacks(pocr_parsenet *p) { return p ? p->range_fallbacks : 0; }

int pocr_parsenet_out_shape(int32_t h, int32_t w, int32_t downsample, int32_t *out_h, int32_t *out_w) {
    if (h <= 0 || w <= 0 || downsample < 1 || !out_h || !out_w) return fail("invalid arguments");
    *out_h = downsample == 1 ? h : cv_round_div(h, downsample);
    *out_w = downsample == 1 ? w : cv_round_div(w, downsample);
    return 0;
}

struct AreaTaps { const double *wy, *wx; const int32_t *y0, *x0; int oh, ow, ty, tx; };

static int parsenet_get_maps_impl(pocr_parsenet *p, const uint8_t *img_hwc, int32_t H, int32_t W, int32_t downsample, const AreaTaps *taps, float *out_hw5);

int pocr_parsenet_get_maps(pocr_parsenet *p, const uint8_t *img_hwc, int32_t H, int32_t W, int32_t downsample, float *out_hw5) {
    return parsenet_get_maps_impl(p, img_hwc, H, W, downsample, nullptr, out_hw5);
}

int pocr_parsenet_get_maps_area(pocr_parsenet *p, const uint8_t *img_hwc, int32_t H, int32_t W, int32_t out_h, int32_t out_w,
                                const double *wy, const int32_t *y0, int32_t taps_y, const double *wx, const int32_t *x0, int32_t taps_x,
                                float *out_hw5) {
    if (!wy || !y0 || !wx || !x0) return fail("NULL tap table");
    if (out_h < 1 || out_w < 1 || taps_y < 1 || taps_x < 1 || taps_y > 64 || taps_x > 64) return fail("invalid resample geometry");
    for (int i = 0; i < out_h; ++i) if (y0[i] < 0 || y0[i] >= H) return fail("row tap start %d outside the page", y0[i]);
    for (int i = 0; i < out_w; ++i) if (x0[i] < 0 || x0[i] >= W) return fail("column tap start %d outside the page", x0[i]);
    AreaTaps t{wy, wx, y0, x0, out_h, out_w, taps_y, taps_x};
    return parsenet_get_maps_impl(p, img_hwc, H, W, 1, &t, out_hw5);
}

static int parsenet_get_maps_impl(pocr_parsenet *p, const uint8_t *img_hwc, int32_t H, int32_t W, int32_t downsample, const AreaTaps *taps, float *out_hw5) {
    if (!p) return fail("handle is NULL");
    if (!img_hwc || !out_hw5) return fail("NULL buffer");
    if (H <= 0 || W <= 0 || downsample < 1) return fail("invalid page size / downsample");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = p->stream;
    int h = H, w = W;
    if (downsample > 1) { h = cv_round_div(H, downsample); w = cv_round_div(W, downsample); }
    if (taps) { h = taps->oh; w = taps->ow; }
    if (h < 1 || w < 1) return fail("page too small for this downsample");
    const int Hp = round_up(h, 64), Wp = round_up(w, 64);
    if ((size_t)Hp * Wp * 128 >= 0xffffffffull) return fail("page too large (%d x %d after padding)", Hp, Wp);
    // ---- upload (+ area down-sampling on the device)
    const size_t in_bytes = (size_t)H * W * 3;
    if (in_bytes > p->pin_in_cap) {
        if (p->pin_in) (void)locked_host_free(p->pin_in);
        p->pin_in = nullptr; p->pin_in_cap = 0;
        HIP_TRY(locked_host_malloc(&p->pin_in, in_bytes, hipHostMallocDefault));
        p->pin_in_cap = in_bytes;
    }
    if (p->page.reserve(in_bytes)) return 1;
    HIP_TRY(upload_through_pinned(p->page.p, p->pin_in, img_hwc, in_bytes, st, p->device));
    HIP_TRY(hipEventRecord(p->ev0, st));
    const uint8_t *src = p->page.as<uint8_t>();
    if (taps) {          // fractional factor: separable area weights from the host, applied in float64 (parsenet.hpp)
        const size_t yb = (size_t)h * taps->ty * 8, xb = (size_t)w * taps->tx * 8, y0b = (size_t)h * 4, x0b = (size_t)w * 4;
        const size_t need = yb + xb + y0b + x0b;
        if (p->small.reserve((size_t)h * w * 3) || p->taps.reserve(need)) return 1;
        if (need > p->pin_taps_cap) {
            if (p->pin_taps) (void)locked_host_free(p->pin_taps);
            p->pin_taps = nullptr; p->pin_taps_cap = 0;
            HIP_TRY(locked_host_malloc(&p->pin_taps, need + need / 4, hipHostMallocDefault));
            p->pin_taps_cap = need + need / 4;
        }
        char *pt = static_cast<char *>(p->pin_taps);
        memcpy(pt, taps->wy, yb); memcpy(pt + yb, taps->wx, xb); memcpy(pt + yb + xb, taps->y0, y0b); memcpy(pt + yb + xb + y0b, taps->x0, x0b);
        HIP_TRY(hipMemcpyAsync(p->taps.p, pt, need, hipMemcpyHostToDevice, st));
        const char *dt = static_cast<const char *>(p->taps.p);
        const int total = h * w * 3;
        hipLaunchKernelGGL(area_resample_u8_kernel, dim3((total + 255) / 256), dim3(256), 0, st, src, H, W,
                           reinterpret_cast<const double *>(dt), reinterpret_cast<const int32_t *>(dt + yb + xb), taps->ty,
                           reinterpret_cast<const double *>(dt + yb), reinterpret_cast<const int32_t *>(dt + yb + xb + y0b), taps->tx,
                           p->small.as<uint8_t>(), h, w);
        HIP_TRY(hipGetLastError());
        src = p->small.as<uint8_t>();
    } else
    if (downsample > 1) {
        if (p->small.reserve((size_t)h * w * 3)) return 1;
        const int total = h * w * 3;
        hipLaunchKernelGGL(area_downsample_u8_kernel, dim3((total + 255) / 256), dim3(256), 0, st, src, H, W, downsample, p->small.as<uint8_t>(), h, w);
        HIP_TRY(hipGetLastError());
        src = p->small.as<uint8_t>();
    }
    // ---- buffers: level k has (Hp >> k) x (Wp >> k) pixels
    const int xc[7] = {64, 128, 256, 256, 256, 256, 256}, pc[7] = {0, 64, 128, 256, 256, 256, 256};
    for (int k = 0; k < 7; ++k) {
        const size_t px = (size_t)(Hp >> k) * (Wp >> k);
        if (p->x[k].reserve(px * xc[k] * sizeof(float))) return 1;
        if (k > 0 && p->p[k].reserve(px * pc[k] * sizeof(float))) return 1;
    }
    const int yc[6] = {256, 256, 256, 128, 64, 64};       // y5 .. y0
    for (int i = 0; i < 6; ++i) {
        const int k = 5 - i;
        if (p->y[i].reserve((size_t)(Hp >> k) * (Wp >> k) * yc[i] * sizeof(float))) return 1;
    }
    // ---- e0: conv1_u8_kernel over the zero canvas (one "line" = the page, Hp x Wp)
    {
        const int th = 4, tw = 32, nh = Hp / th, nw = Wp / tw;
        std::vector<PixelTile> tiles((size_t)nh * nw);
        for (int a_ = 0; a_ < nh; ++a_)
            for (int b_ = 0; b_ < nw; ++b_) tiles[(size_t)a_ * nw + b_] = PixelTile{0, (a_ << 16) | b_};
        LineDesc ld{0, w, 0};
        const int32_t wl = Wp;
        const int64_t off0 = 0;
        if (p->tiles.reserve(tiles.size() * sizeof(PixelTile)) || p->lines.reserve(sizeof(LineDesc)) || p->wline.reserve(16) || p->ooff.reserve(16)) return 1;
        {   // the sources are stack / pageable memory: these copies are synchronising calls (g_unsafe_mu, pocr_hip.hip)
            UnsafeLock lock;
            HIP_TRY(hipMemcpyAsync(p->tiles.p, tiles.data(), tiles.size() * sizeof(PixelTile), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(p->lines.p, &ld, sizeof(ld), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(p->wline.p, &wl, sizeof(wl), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(p->ooff.p, &off0, sizeof(off0), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        Conv1Args c1{};
        c1.crops = src; c1.lines = p->lines.as<LineDesc>(); c1.lut = p->lut.as<float>();
        c1.wfrag = p->enc[0].w.as<float>(); c1.bias = p->enc[0].b.as<float>(); c1.y = p->x[0].as<float>();
        c1.tiles = p->tiles.as<PixelTile>(); c1.line_w = p->wline.as<int32_t>(); c1.out_off = p->ooff.as<int64_t>();
        c1.H = Hp; c1.n_ptiles = (int)tiles.size(); c1.src_h = h;
        hipLaunchKernelGGL(conv1_u8_kernel<false>, dim3(c1.n_ptiles), dim3(256), 0, st, c1);
        HIP_TRY(hipGetLastError());
    }
    // f16x2 range guard: the activations of this network stay fp32 in HBM and are split inside every consumer, so what a layer
    // writes must stay inside f16's range like the recogniser's (pocr_hip.hip: range_verdict); a page that leaves it is run
    // again on the bf16x3 kernels of a second network (below)
    const bool guard = conv_split() == 2;
    int n_guarded = 0;
    if (guard) {
        if (!p->range.p) {
            if (p->range.reserve(24 * 8 * sizeof(unsigned))) return 1;
            HIP_TRY(locked_host_malloc(reinterpret_cast<void **>(&p->range_host), 24 * 8 * sizeof(unsigned), hipHostMallocDefault));
        }
        HIP_TRY(hipMemsetAsync(p->range.p, 0, 24 * 8 * sizeof(unsigned), st));
    }
    auto conv = [&](int (*fn)(ConvArgs, hipStream_t), const PnLayer &L, const float *x, const float *x2, int cin_up, float *y, int Hc, int Wc) {
        ConvArgs a{};
        if (guard && n_guarded < 24) a.range_max = p->range.as<unsigned>() + 8 * n_guarded++;
        a.x = x; a.x2 = x2; a.cin_up = cin_up; a.n = 1; a.H = Hc; a.W = Wc; a.Ho = Hc; a.Wo = Wc; a.cin = L.cin;
        a.cout16 = L.cout16; a.cout_valid = L.cout; a.out_stride = L.cout;
        a.wfrag = L.w.as<float>(); a.bias = L.b.as<float>(); a.y = y;
        return fn(a, st);
    };
    // ---- encoder: e{k} at level k (skip x_k), e{k}p pools into level k+1
    int (*enc_fn[13])(ConvArgs, hipStream_t) = {nullptr, conv2_k, conv3_k, conv4_k, conv56_k, pn_pool256_k, conv56_k, pn_pool256_k,
                                                 conv56_k, pn_pool256_k, conv56_k, pn_pool256_k, conv56_k};
    // the same layers on the bf16x3 kernels (conv_bf16x3.hpp): 64->64 + pool, 64->128, 128->128 + pool, ->256, 256->256 + pool
    int (*enc_fn3[13])(ConvArgs, hipStream_t) = {nullptr, conv2_b3, conv3_b3, conv4_b3, conv56_b3, conv4_b3, conv56_b3, conv4_b3,
                                                  conv56_b3, conv4_b3, conv56_b3, conv4_b3, conv56_b3};
    for (int i = 1; i < 13; ++i) if (p->enc[i].b3) enc_fn[i] = enc_fn3[i];
    for (int i = 1; i < 13; ++i) {
        const int k = i / 2;                          // level of the conv's INPUT: e{k}p (odd i) reads x_k, e{k} (even i) reads p_k
        const bool pool = i & 1;
        const float *in = pool ? p->x[k].as<float>() : p->p[k].as<float>();
        float *outp = pool ? p->p[k + 1].as<float>() : p->x[k].as<float>();
        if (conv(enc_fn[i], p->enc[i], in, nullptr, 0, outp, Hp >> k, Wp >> k)) return 1;
    }
    // ---- decoder: y_k = ReLU(conv(cat(up2(y_{k+1}), x_k))), k = 5 .. 0 (up-sampling and concatenation happen in the conv's staging)
    int (*dec_fn[6])(ConvArgs, hipStream_t) = {pn_up_small_k, pn_up_small_k, pn_up_small_k, pn_up_mid_k, pn_up_big_k, pn_up_big_k};
    int (*dec_fn3[6])(ConvArgs, hipStream_t) = {pn_up128_b3, pn_up128_b3, pn_up128_b3, pn_up128_b3, pn_up64_b3, pn_up64_b3};
    for (int i = 0; i < 6; ++i) if (p->dec[i].b3) dec_fn[i] = dec_fn3[i];
    for (int i = 0; i < 6; ++i) {
        const int k = 5 - i;
        const float *up = i == 0 ? p->x[6].as<float>() : p->y[i - 1].as<float>();
        if (conv(dec_fn[i], p->dec[i], up, p->x[k].as<float>(), kPnDec[i][0], p->y[i].as<float>(), Hp >> k, Wp >> k)) return 1;
    }
    // ---- head + crop
    const size_t out_bytes = (size_t)h * w * 5 * sizeof(float);
    if (p->out.reserve(out_bytes)) return 1;
    hipLaunchKernelGGL(parsenet_head_kernel, dim3((h * w + 255) / 256), dim3(256), 0, st, p->y[5].as<float>(), Wp,
                       p->head_w.as<float>(), p->head_b.as<float>(), p->out.as<float>(), h, w);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(p->ev1, st));
    if (out_bytes > p->pin_out_cap) {
        if (p->pin_out) (void)locked_host_free(p->pin_out);
        p->pin_out = nullptr; p->pin_out_cap = 0;
        HIP_TRY(locked_host_malloc(&p->pin_out, out_bytes, hipHostMallocDefault));
        p->pin_out_cap = out_bytes;
    }
    HIP_TRY(hipMemcpyAsync(p->pin_out, p->out.p, out_bytes, hipMemcpyDeviceToHost, st));
    if (guard) HIP_TRY(hipMemcpyAsync(p->range_host, p->range.p, 24 * 8 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (guard)
        for (int k = 0; k < n_guarded; ++k) {
            unsigned m = 0;
            for (int j = 0; j < 8; ++j) m = std::max(m, p->range_host[8 * k + j]);
            if (m >= 0x477fe000u || (m != 0 && m < 0x39000000u)) {
                // plain fp32 in the reference (pero_ocr/layout_engines/torch_parsenet.py:49-53): the page again on bf16x3
                if (p->weights_host.empty()) return fail("internal error: layout network range guard without a retained weight blob");
                SplitScope scope(3);
                if (!p->shadow) {
                    fprintf(stderr, "NOTE: layout network: conv layer %d left the range of the default f16x2 arithmetic (%s); this page and any later "
                                    "such page are re-run on bf16x3 (fp32's range).  POCR_CONV_SPLIT=3 selects bf16x3 for everything.\n",
                            k + 1, m >= 0x477fe000u ? "|x| >= 65504 or not finite" : "its whole activation lies below 2^-13");
                    if (pocr_parsenet_create(p->weights_host.data(), p->weights_host.size(), p->device, &p->shadow)) return 1;
                }
                ++p->range_fallbacks;
                const int rc = parsenet_get_maps_impl(p->shadow, img_hwc, H, W, downsample, taps, out_hw5);
                p->last_ms = p->shadow->last_ms;
                return rc;
            }
        }
    parallel_memcpy(out_hw5, p->pin_out, out_bytes);
    HIP_TRY(hipEventElapsedTime(&p->last_ms, p->ev0, p->ev1));
    return 0;
}

/* GPU time (ms) of the last pocr_parsenet_get_maps between the end of the upload and the end of the head kernel. */
int pocr_parsenet_last_ms(pocr_parsenet *p, float *ms) {
    if (!p || !ms) return fail("NULL argument");
    *ms = p->last_ms;
    return 0;
}

}  // extern "C"
