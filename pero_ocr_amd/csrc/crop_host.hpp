// crop_host.hpp — host side + C ABI of the resident line cropper (included by pocr_hip.hip, which provides DevBuf /
// fail / HIP_TRY).  Replaces EngineLineCropper.crop per line (pero_ocr/core/crop_engine.py:16-30, 54-99, 146-163) by
// three launches per PAGE: the page is uploaded once (helper thread: pageable -> pinned -> DMA in 4 MB pieces on several host threads, behind
// the caller's per-line host work) and stays in HBM for every line; the only host<->device traffic per page besides
// the page itself is ~100 B of spline per line up, 4 B of width per line back, and the crops.
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>

// Crops that stay in HBM (pocr_cropper_crop_resident): the device buffer one crop call filled, detached from the
// cropper so that the next page can be cropped while the recogniser still reads these.  Released buffers go back to the
// cropper's pool (hipMalloc / hipFree synchronise the device: not once per page).
struct CropPool {                              // shared by a cropper and the buffers detached from it (either may die first)
    std::mutex mu;
    std::vector<DevBuf> bufs;
    bool alive = true;
};
struct pocr_crops {
    int device = 0;
    DevBuf buf;
    size_t bytes = 0;
    std::shared_ptr<CropPool> pool;
};

struct pocr_cropper {
    std::shared_ptr<CropPool> pool = std::make_shared<CropPool>();     // crop buffers handed back by pocr_crops_release
    int device = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr;
    DevBuf page, specs, knots, coefs, state, curves, out, grid;
    void *pin_page = nullptr, *pin_out = nullptr, *pin_small = nullptr;
    size_t pin_page_cap = 0, pin_out_cap = 0, pin_small_cap = 0;
    int H = 0, W = 0, C = 0;
    std::thread uploader;
    std::atomic<int> upload_rc{0};
    std::string upload_err;
    int n = 0;                                 // lines of the last measure
    std::vector<pocr::CropState> hstate;
    float last_ms = 0.f;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

static_assert(sizeof(pocr_crop_spec) == sizeof(pocr::CropSpec), "pocr_crop_spec (include/pocr.h) and CropSpec (csrc/crop.hpp) differ");
static_assert(offsetof(pocr_crop_spec, rot) == offsetof(pocr::CropSpec, rot) && offsetof(pocr_crop_spec, n_x) == offsetof(pocr::CropSpec, n_x),
              "pocr_crop_spec layout");

int cropper_join(pocr_cropper *c) {
    if (c->uploader.joinable()) c->uploader.join();
    if (c->upload_rc.load()) return fail("page upload failed: %s", c->upload_err.c_str());
    return 0;
}

int pin_reserve(void **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap) return 0;
    if (*p) { (void)locked_host_free(*p); *p = nullptr; *cap = 0; }
    const size_t want = bytes + bytes / 8;
    HIP_TRY(locked_host_malloc(p, want, hipHostMallocDefault));
    *cap = want;
    return 0;
}

}  // namespace

extern "C" {

int pocr_cropper_create(int device_id, pocr_cropper **out) {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail("device_id %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    auto *c = new pocr_cropper();
    c->device = device_id;
    if (create_front_stream(&c->stream) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return fail("cannot create the cropper's streams");
    }
    *out = c;
    return 0;
}

void pocr_cropper_destroy(pocr_cropper *c) {
    if (!c) return;
    if (c->uploader.joinable()) c->uploader.join();
    std::vector<DevBuf> pooled;
    {   // detached crop buffers may outlive the cropper: they just stop returning to its pool
        std::lock_guard<std::mutex> g(c->pool->mu);
        c->pool->alive = false;
        pooled.swap(c->pool->bufs);
    }
    (void)hipSetDevice(c->device);
    (void)locked_device_sync();
    for (DevBuf *b : {&c->page, &c->specs, &c->knots, &c->coefs, &c->state, &c->curves, &c->out, &c->grid}) b->release();
    for (DevBuf &b : pooled) b.release();
    for (void *p : {c->pin_page, c->pin_out, c->pin_small}) if (p) (void)locked_host_free(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

int pocr_cropper_set_page(pocr_cropper *c, const uint8_t *page_hwc, int32_t H, int32_t W, int32_t C) {
    if (!c || !page_hwc) return fail("NULL pointer");
    if (H <= 0 || W <= 0 || C < 1 || C > 4) return fail("bad page geometry (H %d, W %d, C %d)", H, W, C);
    if (cropper_join(c)) return 1;
    HIP_TRY(hipSetDevice(c->device));
    // no kernel of an earlier page may still read the buffer the upload is about to overwrite
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t bytes = (size_t)H * W * C;
    if (c->page.reserve(bytes) || pin_reserve(&c->pin_page, &c->pin_page_cap, bytes)) return 1;
    c->H = H; c->W = W; c->C = C;
    c->upload_rc.store(0);
    c->uploader = std::thread([c, page_hwc, bytes]() {
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = upload_through_pinned(c->page.p, c->pin_page, page_hwc, bytes, c->copy_stream, c->device);
        if (e == hipSuccess) e = hipStreamSynchronize(c->copy_stream);
        if (e != hipSuccess) { c->upload_err = hipGetErrorString(e); c->upload_rc.store(1); }
    });
    return 0;
}

int pocr_cropper_wait_page(pocr_cropper *c) {
    if (!c) return fail("cropper is NULL");
    return cropper_join(c);
}

int pocr_cropper_measure(pocr_cropper *c, const pocr_crop_spec *specs, int32_t n, const double *knots, int64_t n_knots,
                         const double *coefs, int64_t n_coefs, int32_t *widths, int32_t *status) {
    if (!c || !specs || !widths || !status || (n_knots > 0 && !knots) || (n_coefs > 0 && !coefs)) return fail("NULL pointer");
    if (n <= 0) return fail("no lines");
    for (int i = 0; i < n; ++i) {
        const pocr_crop_spec &s = specs[i];
        const int need_k = s.mode == 0 ? s.n_coef + 4 : 0;
        if ((s.mode != 0 && s.mode != 1) || s.n_coef < (s.mode == 0 ? 4 : 1) || s.coef_off < 0 || s.coef_off + (int64_t)s.n_coef > n_coefs ||
            s.knot_off < 0 || s.knot_off + (int64_t)need_k > n_knots || s.n_x < 0)
            return fail("line %d: bad interpolant description (mode %d, %d coefficients)", i, s.mode, s.n_coef);
    }
    HIP_TRY(hipSetDevice(c->device));
    const size_t sb = (size_t)n * sizeof(pocr::CropSpec), kb = (size_t)std::max<int64_t>(1, n_knots) * 8, cb = (size_t)std::max<int64_t>(1, n_coefs) * 8;
    const size_t stb = (size_t)n * sizeof(pocr::CropState);
    if (c->specs.reserve(sb) || c->knots.reserve(kb) || c->coefs.reserve(cb) || c->state.reserve(stb) ||
        pin_reserve(&c->pin_small, &c->pin_small_cap, sb + kb + cb + stb))
        return 1;
    HIP_TRY(hipStreamSynchronize(c->stream));              // pin_small is reused
    uint8_t *ps = static_cast<uint8_t *>(c->pin_small);
    std::memcpy(ps, specs, sb);
    if (n_knots > 0) std::memcpy(ps + sb, knots, (size_t)n_knots * 8);
    if (n_coefs > 0) std::memcpy(ps + sb + kb, coefs, (size_t)n_coefs * 8);
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(hipMemcpyAsync(c->specs.p, ps, sb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->knots.p, ps + sb, kb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->coefs.p, ps + sb + kb, cb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->state.p, 0, stb, c->stream));
    hipLaunchKernelGGL(pocr::crop_arc_kernel, dim3(n), dim3(256), 0, c->stream, c->specs.as<pocr::CropSpec>(), c->knots.as<double>(),
                       c->coefs.as<double>(), c->state.as<pocr::CropState>());
    HIP_TRY(hipGetLastError());
    pocr::CropState *hs = reinterpret_cast<pocr::CropState *>(ps + sb + kb + cb);
    HIP_TRY(hipMemcpyAsync(hs, c->state.p, stb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->hstate.assign(hs, hs + n);
    c->n = n;
    for (int i = 0; i < n; ++i) { widths[i] = hs[i].width; status[i] = hs[i].status; }
    return 0;
}

static int cropper_crop_impl(pocr_cropper *c, int32_t line_height, const int64_t *crop_off, uint8_t *crops, float *grid_out, int32_t *status,
                             bool to_host, size_t *out_bytes) {
    if (!c || !crop_off || !status) return fail("NULL pointer");
    if (c->n <= 0) return fail("pocr_cropper_measure has not run");
    if (line_height <= 0) return fail("bad line height %d", line_height);
    if (c->H == 0) return fail("pocr_cropper_set_page has not run");
    HIP_TRY(hipSetDevice(c->device));
    const int n = c->n;
    int64_t n_curve = 0, n_out = 0, n_grid = 0;
    int w_max = 0;
    for (int i = 0; i < n; ++i) {
        pocr::CropState &s = c->hstate[i];
        if (crop_off[i] < 0) return fail("line %d: negative offset", i);
        s.curve_off = n_curve; s.out_off = crop_off[i]; s.grid_off = n_grid;
        if (s.status) continue;
        n_curve += (int64_t)4 * s.width;
        n_grid += (int64_t)2 * line_height * s.width;
        n_out = std::max<int64_t>(n_out, crop_off[i] + (int64_t)line_height * s.width * c->C);
        w_max = std::max(w_max, s.width);
    }
    const size_t stb = (size_t)n * sizeof(pocr::CropState);
    if (w_max > 0) {
        if (!to_host && c->out.cap < (size_t)n_out) {      // a pooled buffer that is large enough, if there is one
            std::lock_guard<std::mutex> g(c->pool->mu);
            for (size_t k = 0; k < c->pool->bufs.size(); ++k)
                if (c->pool->bufs[k].cap >= (size_t)n_out) { std::swap(c->out, c->pool->bufs[k]); break; }
        }
        if (c->curves.reserve((size_t)n_curve * 8) || c->out.reserve((size_t)n_out) || (grid_out && c->grid.reserve((size_t)n_grid * 4)) ||
            (to_host && pin_reserve(&c->pin_out, &c->pin_out_cap, (size_t)n_out)))
            return 1;
        pocr::CropState *hs = reinterpret_cast<pocr::CropState *>(c->pin_small);      // measure has synchronised: the block is free
        std::memcpy(hs, c->hstate.data(), stb);
        HIP_TRY(hipMemcpyAsync(c->state.p, hs, stb, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(pocr::crop_columns_kernel, dim3((w_max + 255) / 256, n), dim3(256), 0, c->stream, c->specs.as<pocr::CropSpec>(),
                           c->knots.as<double>(), c->coefs.as<double>(), c->state.as<pocr::CropState>(), c->curves.as<double>());
        HIP_TRY(hipGetLastError());
        if (cropper_join(c)) return 1;                     // the page must have landed before the pixel kernel
        hipLaunchKernelGGL(pocr::remap_spec_u8_kernel, dim3((line_height * w_max + 255) / 256, n), dim3(256), 0, c->stream, c->page.as<uint8_t>(),
                           c->H, c->W, c->C, c->specs.as<pocr::CropSpec>(), c->state.as<pocr::CropState>(), c->curves.as<double>(), line_height,
                           c->out.as<uint8_t>(), grid_out ? c->grid.as<float>() : (float *)nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(hs, c->state.p, stb, hipMemcpyDeviceToHost, c->stream));
        if (to_host) HIP_TRY(hipMemcpyAsync(c->pin_out, c->out.p, (size_t)n_out, hipMemcpyDeviceToHost, c->stream));
        if (grid_out) HIP_TRY(hipMemcpyAsync(grid_out, c->grid.p, (size_t)n_grid * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1);
        for (int i = 0; i < n; ++i) c->hstate[i].status = hs[i].status;
        if (crops && to_host) std::memcpy(crops, c->pin_out, (size_t)n_out);
    } else if (cropper_join(c)) {
        return 1;
    }
    for (int i = 0; i < n; ++i) status[i] = c->hstate[i].status;
    if (out_bytes) *out_bytes = (size_t)n_out;
    return 0;
}

int pocr_cropper_crop(pocr_cropper *c, int32_t line_height, const int64_t *crop_off, uint8_t *crops, float *grid_out, int32_t *status) {
    return cropper_crop_impl(c, line_height, crop_off, crops, grid_out, status, true, nullptr);
}

int pocr_cropper_crop_resident(pocr_cropper *c, int32_t line_height, const int64_t *crop_off, int32_t *status, pocr_crops **out) {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    size_t bytes = 0;
    if (cropper_crop_impl(c, line_height, crop_off, nullptr, nullptr, status, false, &bytes)) return 1;
    auto *k = new pocr_crops();
    k->device = c->device; k->pool = c->pool; k->bytes = bytes;
    std::swap(k->buf, c->out);                         // the cropper allocates (or takes from its pool) another one next time
    *out = k;
    return 0;
}

void pocr_crops_release(pocr_crops *k) {
    if (!k) return;
    if (k->pool) {
        std::lock_guard<std::mutex> g(k->pool->mu);
        if (k->pool->alive && k->buf.p && k->pool->bufs.size() < 8) { k->pool->bufs.emplace_back(); std::swap(k->pool->bufs.back(), k->buf); }
    }
    (void)hipSetDevice(k->device);
    k->buf.release();
    delete k;
}

int64_t pocr_crops_bytes(const pocr_crops *k) { return k ? (int64_t)k->bytes : 0; }

int pocr_crops_read(const pocr_crops *k, int64_t offset, int64_t nbytes, uint8_t *out) {
    if (!k || !out) return fail("NULL pointer");
    if (offset < 0 || nbytes < 0 || (size_t)(offset + nbytes) > k->bytes) return fail("range [%lld, +%lld) outside the %zu crop bytes", (long long)offset, (long long)nbytes, k->bytes);
    if (nbytes == 0) return 0;
    HIP_TRY(hipSetDevice(k->device));
    HIP_TRY(locked_memcpy(out, static_cast<const uint8_t *>(k->buf.p) + offset, (size_t)nbytes, hipMemcpyDeviceToHost));
    return 0;
}

int pocr_slot_stage_resident(pocr_engine *e, int32_t slot, const pocr_crops *const *crops_of_line, const int64_t *crop_offsets,
                             const int32_t *widths, const int32_t *w_pads, int32_t n, int32_t pad_left) {
    if (check_slot(e, slot)) return 1;
    if (!crops_of_line || !crop_offsets || !widths || !w_pads) return fail("NULL input pointer");
    if (n <= 0) return fail("n must be positive (got %d)", n);
    const int H = e->cfg.height;
    const uint8_t *base = nullptr;
    std::vector<int64_t> rel((size_t)n);
    for (int i = 0; i < n; ++i) {
        const pocr_crops *k = crops_of_line[i];
        if (!k || !k->buf.p) return fail("line %d: no resident crop buffer", i);
        if (k->device != e->device) return fail("line %d: its crops live on device %d, the engine on device %d", i, k->device, e->device);
        if (widths[i] < 0 || crop_offsets[i] < 0 || (size_t)crop_offsets[i] + (size_t)H * widths[i] * 3 > k->bytes)
            return fail("line %d: crop [%d x %d x 3] at offset %lld lies outside its %zu-byte buffer", i, H, widths[i], (long long)crop_offsets[i], k->bytes);
        if (!base) base = static_cast<const uint8_t *>(k->buf.p);
        rel[i] = (static_cast<const uint8_t *>(k->buf.p) - base) + crop_offsets[i];       // lines of several buffers: offsets of either sign
    }
    return stage_ragged_impl(e, slot, nullptr, rel.data(), widths, w_pads, n, pad_left, nullptr, base);
}

int pocr_cropper_read_curves(pocr_cropper *c, double *out, int64_t cap) {
    if (!c || !out) return fail("NULL pointer");
    HIP_TRY(hipSetDevice(c->device));
    int64_t total = 0;
    for (int i = 0; i < c->n; ++i) if (!c->hstate[i].status || c->hstate[i].curve_off + 4 * (int64_t)c->hstate[i].width > total)
        total = std::max<int64_t>(total, c->hstate[i].curve_off + 4 * (int64_t)c->hstate[i].width);
    if (total > cap) return fail("curve buffer holds %lld doubles, caller offers %lld", (long long)total, (long long)cap);
    if (total > 0) HIP_TRY(locked_memcpy(out, c->curves.p, (size_t)total * 8, hipMemcpyDeviceToHost));
    return 0;
}

const uint8_t *pocr_cropper_pinned_crops(pocr_cropper *c) { return c ? static_cast<const uint8_t *>(c->pin_out) : nullptr; }

float pocr_cropper_last_ms(pocr_cropper *c) { return c ? c->last_ms : 0.f; }

}  // extern "C"
