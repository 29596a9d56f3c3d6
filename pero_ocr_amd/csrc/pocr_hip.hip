// pocr_hip.hip — C ABI (include/pocr.h) + host-side orchestration of the gfx950 kernels.
// One engine = one GPU, one HIP stream.  No CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <atomic>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <mutex>
#include <unordered_set>
#include <map>
#include <tuple>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pocr.h"
#include "conv_igemm.hpp"
#include "conv1_u8.hpp"
#include "conv_bf16x3.hpp"
#include "conv_rows.hpp"
#include "gemm_f16x2.hpp"
#include "ctc.hpp"
#include "encoder.hpp"
#include "decoder.hpp"
#include "crop.hpp"
#include "lstm.hpp"
#include "lstm_resident.hpp"
#include "sparsify.hpp"
#include "comm.hpp"
#include "parsenet.hpp"

using namespace pocr;

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// Calls that are "unsafe" while another thread of the process captures a stream into a hipGraph (allocation, free, the
// synchronising copies and device-wide synchronisation) and the capture itself exclude each other: a page stream runs the
// layout network and the cropper on a helper thread while the recogniser's thread may be capturing its recurrence
// (hipStreamCaptureModeThreadLocal permits that on paper; the lock is cheap insurance - it costs nothing in the steady state,
// where no buffer grows and every graph is cached).  Under the lock: hipMalloc / hipFree (DevBuf, the RCCL staging buffers),
// hipHostMalloc / hipHostFree, hipMemcpy, hipDeviceSynchronize, copies from pageable memory, stream capture + instantiation.
// Deliberately NOT under it: hipStreamSynchronize / hipEventSynchronize on a stream of the calling thread's own object
// (slot, cropper, layout network), hipMemcpyAsync between pinned and device memory, and kernel launches - none of them is
// on HIP's list of calls that invalidate another thread's thread-local capture, and they are the steady state of a page stream.
std::recursive_mutex g_unsafe_mu;
struct UnsafeLock { std::lock_guard<std::recursive_mutex> g{g_unsafe_mu}; };
inline hipError_t locked_host_malloc(void **p, size_t n, unsigned flags) { UnsafeLock l; return hipHostMalloc(p, n, flags); }
inline hipError_t locked_host_free(void *p) { UnsafeLock l; return hipHostFree(p); }
inline hipError_t locked_device_sync() { UnsafeLock l; return hipDeviceSynchronize(); }

// A page (36 MB at 4k x 3k) from pageable memory to the device: `n_threads` host threads (the caller is one of them) take 4 MB
// pieces in turn, copy each into the pinned buffer and queue its DMA on `st` - the pieces are independent, so neither their order on
// the stream nor which thread queued them matters, and the DMA of a piece runs behind the host copies of the next ones.  One
// thread moves ~12 GB/s out of pageable memory, so a lone memcpy in front of a lone DMA (rounds 1-3) cost a 4k x 3k page 3.6 ms.
inline int upload_threads() {
    static const int n = [] { const char *e = getenv("POCR_UPLOAD_THREADS"); int v = e ? atoi(e) : 4; return v < 1 ? 1 : v > 16 ? 16 : v; }();
    return n;
}
inline hipError_t upload_through_pinned(void *dev, void *pin, const void *src, size_t bytes, hipStream_t st, int device) {
    const size_t piece = (size_t)4 << 20;
    const size_t n_pieces = (bytes + piece - 1) / piece;
    const int n_threads = (int)std::min<size_t>((size_t)upload_threads(), std::max<size_t>(n_pieces, 1));
    std::atomic<size_t> next{0};
    std::atomic<int> err{(int)hipSuccess};
    auto work = [&](bool set_device) {
        if (set_device) { hipError_t e = hipSetDevice(device); if (e != hipSuccess) { err.store((int)e); return; } }
        for (size_t i = next.fetch_add(1); i < n_pieces && err.load() == (int)hipSuccess; i = next.fetch_add(1)) {
            const size_t o = i * piece, nb = std::min(piece, bytes - o);
            std::memcpy(static_cast<uint8_t *>(pin) + o, static_cast<const uint8_t *>(src) + o, nb);
            hipError_t e = hipMemcpyAsync(static_cast<uint8_t *>(dev) + o, static_cast<uint8_t *>(pin) + o, nb, hipMemcpyHostToDevice, st);
            if (e != hipSuccess) err.store((int)e);
        }
    };
    std::vector<std::thread> helpers;
    for (int t = 1; t < n_threads; ++t) helpers.emplace_back(work, true);
    work(false);
    for (std::thread &h : helpers) h.join();
    return (hipError_t)err.load();
}

// Host copy of a large result (the layout maps: 15.7 MB per 4k x 3k page, into a fresh numpy array whose pages are touched for the
// first time) split over the same helper threads.
inline void parallel_memcpy(void *dst, const void *src, size_t bytes) {
    const size_t piece = (size_t)2 << 20;
    const size_t n_pieces = (bytes + piece - 1) / piece;
    const int n_threads = (int)std::min<size_t>((size_t)upload_threads(), std::max<size_t>(n_pieces, 1));
    if (n_threads <= 1) { std::memcpy(dst, src, bytes); return; }
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t i = next.fetch_add(1); i < n_pieces; i = next.fetch_add(1)) {
            const size_t o = i * piece;
            std::memcpy(static_cast<uint8_t *>(dst) + o, static_cast<const uint8_t *>(src) + o, std::min(piece, bytes - o));
        }
    };
    std::vector<std::thread> helpers;
    for (int t = 1; t < n_threads; ++t) helpers.emplace_back(work);
    work();
    for (std::thread &h : helpers) h.join();
}

// The streams of the page front (layout network, cropper): ordinary priority.  In a page stream their kernels share the GPU with the
// recogniser's convolutions of EARLIER pages.  Greatest priority for them (and a third front pair) was measured in rounds 4 and 6 on
// the c5 stream - 72.9-75.6 pages/s as shipped, 73.6-74.7 with high priority, 74.1-75.1 with three fronts, 74.1-74.4 with both
// (profiles/r06_c5_front_ab.txt): the stream is bound by the GPU's work per page, not by the front's latency - and removed.
inline hipError_t create_front_stream(hipStream_t *st) { return hipStreamCreateWithFlags(st, hipStreamNonBlocking); }
inline hipError_t locked_memcpy(void *d, const void *s_, size_t n, hipMemcpyKind k) { UnsafeLock l; return hipMemcpy(d, s_, n, k); }
inline hipError_t locked_memcpy2d(void *d, size_t dp, const void *s_, size_t sp, size_t w, size_t h, hipMemcpyKind k) { UnsafeLock l; return hipMemcpy2D(d, dp, s_, sp, w, h, k); }

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        UnsafeLock lock;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        // debugging aid: no kernel may depend on what fresh memory holds.  POCR_POISON=1, or "lo:hi" = only allocations of lo..hi bytes
        static const char *poison = getenv("POCR_POISON");
        if (poison) {
            size_t lo = 0, hi = ~(size_t)0;
            if (strchr(poison, ':')) { lo = strtoull(poison, nullptr, 10); hi = strtoull(strchr(poison, ':') + 1, nullptr, 10); }
            if (bytes >= lo && bytes <= hi) { HIP_TRY(hipMemset(p, 0xFF, want)); HIP_TRY(locked_device_sync()); }
        }
        return 0;
    }
    void release() { UnsafeLock lock; if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// (cin, cout, act, poolh, poolw) — same table as pero_ocr_amd/netspec.py CONV_PLAN
struct ConvLayer { int cin, cout, act, ph, pw; };
const ConvLayer kConvPlan[9] = {
    {3, 64, ACT_RELU, 1, 1},    {64, 64, ACT_RELU, 2, 2},   {64, 128, ACT_RELU, 1, 1},
    {128, 128, ACT_RELU, 2, 2}, {128, 256, ACT_RELU, 1, 1}, {256, 256, ACT_RELU, 1, 1},
    {256, 256, ACT_RELU, 2, 1}, {256, 512, ACT_LEAKY, 1, 1}, {512, 512, ACT_LEAKY, 1, 1},
};
const float kBnEps = 1e-5f;
// f16x2 range guard: sets of 8 words (conv_igemm.hpp: range_publish).  Sets 0..8 = the activations of conv1..conv9, 9 = the
// aggregated features, 10 = everything else an f16x2 GEMM consumes (LayerNorm / attention / feed-forward outputs).
constexpr int kRangeSets = 16, kRangeWords = kRangeSets * 8, kRangeOther = 10;

// fragment-order weights: wfrag[tap][cin/16][cout16][lane][j]
//   = W(cout = 16*s + (lane & 15), cin = 16*g + 4*(lane >> 4) + j, tap), zero outside the valid range.
std::vector<float> build_wfrag(int ntaps, int cin_pad, int cout16,
                               const std::function<float(int, int, int)> &W, int cin_valid, int cout_valid) {
    std::vector<float> out((size_t)ntaps * (cin_pad / 16) * cout16 * 256, 0.f);
    size_t o = 0;
    for (int tap = 0; tap < ntaps; ++tap)
        for (int g = 0; g < cin_pad / 16; ++g)
            for (int s = 0; s < cout16; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j, ++o) {
                        const int co = 16 * s + (lane & 15), ci = 16 * g + 4 * (lane >> 4) + j;
                        if (co < cout_valid && ci < cin_valid) out[o] = W(co, ci, tap);
                    }
    return out;
}

template <class Kern>
int launch_conv(Kern kern, int TH, int TW, int NT, int nthreads, ConvArgs a, hipStream_t st) {
    if (!a.tiles) {
        a.tiles_w = (a.Wo + TW - 1) / TW;
        a.tiles_h = (a.Ho + TH - 1) / TH;
    }
    a.tiles_n = (a.cout16 * 16) / NT;
    // Channel tiles per XCD (ConvArgs::xcd_g): one.  That keeps 1 / tiles_n of a layer's weights in each L2 and lets tiles_n XCDs fetch
    // the same halo tile; two per XCD was measured: same time, and the HBM-side read bytes of conv9 do NOT drop (2.34 -> 2.14 GB per
    // launch: what the halo re-reads save, the weight stream - then 4.7 MB per XCD, over the 4 MB L2 - costs;
    // profiles/r03_xcd_mapping_experiment.txt).  The kernels' tile map still takes the group size as an argument.
    if (a.xcd_g == 0) a.xcd_g = 1;
    const size_t blocks = conv_grid_blocks(a);
    if (blocks == 0) return 0;
    if (blocks > 0x7fffffffull) return fail("conv grid too large (%zu blocks)", blocks);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(nthreads), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// conv_rows.hpp: the 3x3 f16x2 / P2 layers with the result layout [channel][pixel] and the output tile leaving through LDS as
// whole lines; same bits as the kernels of conv_bf16x3.hpp.  The kernel can also run PERSISTENTLY - a grid of two workgroups per
// CU, each walking the blocks blockIdx.x, + gridDim.x, ... with the next tile's first chunk requested under the epilogue (needs an
// even number of 32-channel chunks and a mapping that keeps a workgroup on one channel tile: 1 tile, or 2 / 4 / 8 with a grid that
// is a multiple of 64).  Which layer takes which form was measured on a c2 chunk alone (profiles/r05_conv_rows.txt; ms, old kernel /
// rows / rows persistent): conv3 0.657 / 0.603 / 0.555, conv4 0.95 / 0.92 / 0.97, conv5 0.544 / 0.507 / 0.527, conv6 0.943 / 0.926 /
// 0.962, conv7 0.92 / 0.945 / 0.966 (0.96 either way in the evidence run, within the boxes' spread), conv8 0.977 / 0.974 / 0.972, conv9 1.815 / 1.79 / 1.83 - the two workgroups of a CU already cover
// each other's prologue, so persistence pays only where the prologue is a third of a tile (conv3).
// POCR_CONV_ROWS=0: conv3 .. conv9 on the conv_bf16x3.hpp kernels they replace; POCR_CONV_PERSIST_MASK: bit i = conv(i + 1) in the
// persistent form (default 0x4: conv3) - both are the A / B switches of test_conv_rows_kernels_are_bit_identical_to_the_one_tile_kernels.
thread_local int g_conv_layer = -1;                     // index of the layer being launched (run_network)
inline int conv_rows_mode() {                           // 0: conv_bf16x3.hpp, 1: conv_rows.hpp, 2: conv_rows.hpp persistent
    static const bool rows_on = [] { const char *e = getenv("POCR_CONV_ROWS"); return !(e && atoi(e) == 0); }();
    static const int pers_mask = [] { const char *m = getenv("POCR_CONV_PERSIST_MASK"); return m ? (int)strtol(m, nullptr, 0) : 0x4; }();
    const int l = g_conv_layer;
    if (!rows_on) return 0;
    if (l < 2 || l > 8) return 1;                       // (conv3 .. conv9 are layers 2 .. 8; other callers: the plain form)
    return ((pers_mask >> l) & 1) ? 2 : 1;
}
template <class Kern>
int launch_conv_rows(Kern kern, int TH, int TW, int NT, ConvArgs a, hipStream_t st) {
    if (!a.tiles) {
        a.tiles_w = (a.Wo + TW - 1) / TW;
        a.tiles_h = (a.Ho + TH - 1) / TH;
    }
    a.tiles_n = (a.cout16 * 16) / NT;
    if (a.xcd_g == 0) a.xcd_g = 1;
    const size_t blocks = conv_grid_blocks(a);
    if (blocks == 0) return 0;
    if (blocks > 0x7fffffffull) return fail("conv grid too large (%zu blocks)", blocks);
    static const int n_cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int tn = a.tiles_n;
    const bool one_tile = tn == 1 || (tn >= 2 && tn <= 8 && (tn & (tn - 1)) == 0);
    size_t grid = blocks;
    if (one_tile && ((a.cin / 32) & 1) == 0) {
        const size_t cap = conv_rows_mode() == 2 ? (size_t)(n_cus * 2) / 64 * 64 : 0;
        if (cap >= 64 && cap < blocks) grid = cap;
    }
    a.nblocks = (int32_t)blocks;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// tile configurations (KH,KW,PADH,PADW, TH,MW,NS,NWAVE,KC, POOLH,POOLW, ACT,BN, STAGER, PIPE)
// PI = PIPE_INTERLEAVED (unrolled taps, MFMA-interleaved staging, see conv_igemm.hpp); PP = PIPE_PLAIN.
#define POCR_CONV(name, KH, KW, PH, PW, TH, MW, NS, NWAVE, KC, POOLH, POOLW, ACT, BN, STG, PIPE)                       \
    int name(ConvArgs a, hipStream_t st) {                                                                           \
        return launch_conv(conv_igemm_kernel<KH, KW, PH, PW, TH, MW, NS, NWAVE, KC, POOLH, POOLW, ACT, BN, STG, PIPE>, \
                           TH, 16 * MW, NS * NWAVE * 16, NWAVE * 64, a, st);                                        \
    }
//                 KH KW P  P  TH MW NS NW KC PH PW
POCR_CONV(conv2_k,   3, 3, 1, 1, 4, 4, 1, 4, 16, 2, 2, ACT_RELU, false, STAGE_F32_NHWC, PIPE_DEEP)          // 64->64   + pool 2x2
POCR_CONV(conv3_k,   3, 3, 1, 1, 4, 2, 2, 4, 16, 1, 1, ACT_RELU, false, STAGE_F32_NHWC, PIPE_INTERLEAVED)   // 64->128
POCR_CONV(conv4_k,   3, 3, 1, 1, 4, 2, 2, 4, 16, 2, 2, ACT_RELU, false, STAGE_F32_NHWC, PIPE_INTERLEAVED)   // 128->128 + pool 2x2
POCR_CONV(conv56_k,  3, 3, 1, 1, 10, 1, 2, 4, 16, 1, 1, ACT_RELU, false, STAGE_F32_NHWC, PIPE_DEEP)         // ->256
POCR_CONV(conv7_k,   3, 3, 1, 1, 10, 1, 2, 4, 16, 2, 1, ACT_RELU, false, STAGE_F32_NHWC, PIPE_INTERLEAVED)  // 256->256 + pool 2x1
POCR_CONV(conv8_k,   3, 3, 1, 1, 5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, false, STAGE_F32_NHWC, PIPE_DEEP)         // 256->512
POCR_CONV(conv9_k,   3, 3, 1, 1, 5, 1, 4, 4, 16, 1, 1, ACT_LEAKY, true, STAGE_F32_NHWC, PIPE_DEEP)          // 512->512 + BN; weight tile requested two steps ahead (+2 %)
POCR_CONV(agg4_k,    4, 1, 0, 0, 1, 3, 4, 4, 16, 1, 1, ACT_LEAKY, false, STAGE_F32_NHWC, PIPE_INTERLEAVED)
POCR_CONV(agg5_k,    5, 1, 0, 0, 1, 3, 4, 4, 16, 1, 1, ACT_LEAKY, false, STAGE_F32_NHWC, PIPE_INTERLEAVED)
POCR_CONV(agg6_k,    6, 1, 0, 0, 1, 3, 4, 4, 16, 1, 1, ACT_LEAKY, false, STAGE_F32_NHWC, PIPE_PLAIN)
POCR_CONV(agg8_k,    8, 1, 0, 0, 1, 3, 4, 4, 16, 1, 1, ACT_LEAKY, false, STAGE_F32_NHWC, PIPE_PLAIN)
POCR_CONV(gemm128_k, 1, 1, 0, 0, 1, 8, 2, 4, 16, 1, 1, ACT_NONE, false, STAGE_F32_NHWC, PIPE_DEEP)   // rows x 128 cols per WG
POCR_CONV(gemm64_k,  1, 1, 0, 0, 1, 8, 1, 4, 16, 1, 1, ACT_NONE, false, STAGE_F32_NHWC, PIPE_DEEP)   // rows x 64 cols per WG
POCR_CONV(gemm128_relu_k, 1, 1, 0, 0, 1, 8, 2, 4, 16, 1, 1, ACT_RELU, false, STAGE_F32_NHWC, PIPE_DEEP)   // FFN first linear
const int kConvNT[9] = {64, 64, 128, 128, 128, 128, 128, 256, 256};
// The same layers on the bf16 matrix pipe with fp32-level accuracy (conv_bf16x3.hpp: three-way exact operand split, six
// bf16 MFMAs per product block); tile configurations from tools/conv_bench_bf16.hip.  (TH, MW, NS, WM, POOLH, POOLW, ACT, BN)
// Operand split of the split-precision kernels, process-wide (weights are laid out for it at creation): 2 = f16x2 (two f16
// planes, THREE MFMAs per 32-deep product block: ceiling 2500 / 3 = 833 TFLOP/s of algorithmic fp32 FLOPs; the default),
// 3 = bf16x3 (three bf16 planes, six MFMAs: 416.7; POCR_CONV_SPLIT=3), 0 = fp32 MFMA kernels (POCR_CONV_FP32=1: 157.3).
// The fall-back engine of the f16x2 range guard (below: run_fallback) is a second engine of the same process that runs on
// bf16x3: the calls made on its behalf set this thread-local override.
thread_local int g_split_tls = -1;
struct SplitScope {
    int prev;
    explicit SplitScope(int v) : prev(g_split_tls) { g_split_tls = v; }
    ~SplitScope() { g_split_tls = prev; }
};
int conv_split() {
    if (g_split_tls >= 0) return g_split_tls;
    static const int mode = [] {
        if (const char *env = getenv("POCR_CONV_FP32")) if (atoi(env) != 0) return 0;
        if (const char *env = getenv("POCR_CONV_SPLIT")) return atoi(env) == 3 ? 3 : 2;
        return 2;
    }();
    return mode;
}
// MINW2: workgroups per CU the f16x2 instantiation is budgeted for (launch bounds)
#define POCR_CONV3(name, TH, MW, NS, WM, POOLH, POOLW, ACT, BN, MINW, BDIR, MINW2)                                 \
    int name(ConvArgs a, hipStream_t st) {                                                                         \
        if (conv_split() == 2)                                                                                     \
            return launch_conv(conv3x3_bf16x3_kernel<TH, MW, NS, WM, POOLH, POOLW, ACT, BN, MINW2, BDIR, 3, 3, 1, 1, false, 2>, \
                               TH, 16 * MW, NS * (4 / WM) * 16, 256, a, st);                                       \
        return launch_conv(conv3x3_bf16x3_kernel<TH, MW, NS, WM, POOLH, POOLW, ACT, BN, MINW, BDIR>, TH, 16 * MW,   \
                           NS * (4 / WM) * 16, 256, a, st);                                                        \
    }
// measured per layer in profiles/r02_conv_bf16x3_bench.txt; BDIR = weights straight from L2 into registers (no LDS tile):
// wins wherever each wave owns its channels (WM 1) and the accumulators leave room for three weight sets
// the low-K layers (2-4 chunks of K) want SMALL tiles: two or three workgroups per CU cover each other's prologue / epilogue
// (MFMA busy 44-52 % with one 4x64 / 4x32 workgroup per CU; -10 ... -16 % with these)
POCR_CONV3(conv2_b3,  4, 2, 2, 2, 2, 2, ACT_RELU, false, 2, false, 2)   // 64->64 + pool 2x2: 4x32 px, NT 64, waves 2 (pixels) x 2 (channels)
POCR_CONV3(conv3_b3,  5, 1, 2, 1, 1, 1, ACT_RELU, false, 2, true, 2)    // 64->128: 5x16 px, NT 128
POCR_CONV3(conv4_b3,  4, 1, 2, 1, 2, 2, ACT_RELU, false, 2, true, 2)    // 128->128 + pool 2x2: 4x16 px
POCR_CONV3(conv56_b3, 5, 1, 2, 1, 1, 1, ACT_RELU, false, 2, true, 2)    // ->256: 5x16 px, NT 128, two workgroups per CU
POCR_CONV3(conv7_b3,  2, 2, 2, 1, 2, 1, ACT_RELU, false, 2, true, 2)    // 256->256 + pool 2x1: 2x32 px (the pool needs an even tile height)
POCR_CONV3(conv8_b3,  5, 1, 2, 1, 1, 1, ACT_LEAKY, false, 2, true, 2)   // 256->512: 5x16 px, NT 128, two workgroups per CU
POCR_CONV3(conv9_b3,  5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2, true, 2)    // 512->512 + BN
// the aggregation conv (AH x 1, no padding) and the GEMM-mode layers (1 x 1: LSTM input projections, encoder linears) on the
// same kernel: weights through LDS, 48 pixels x 256 channels / 128 rows x 128 columns per workgroup
#define POCR_CONV3G(name, TH, MW, NS, WM, ACT, MINW, KH, BDIR)                                                      \
    int name(ConvArgs a, hipStream_t st) {                                                                         \
        if (conv_split() == 2)                                                                                     \
            return launch_conv(conv3x3_bf16x3_kernel<TH, MW, NS, WM, 1, 1, ACT, false, MINW, BDIR, KH, 1, 0, 0, false, 2>, TH, \
                               16 * MW, NS * (4 / WM) * 16, 256, a, st);                                           \
        return launch_conv(conv3x3_bf16x3_kernel<TH, MW, NS, WM, 1, 1, ACT, false, MINW, BDIR, KH, 1, 0, 0>, TH,    \
                           16 * MW, NS * (4 / WM) * 16, 256, a, st);                                               \
    }
// f16x2 with PRE-SPLIT activations (conv_bf16x3.hpp "P2"): the recogniser's conv stack in the default mode.  conv1 (inside conv2's prologue by default) writes
// the two-plane layout, conv2 .. conv9 read and write it (stager = 16-byte copies), the aggregation conv reads it and
// writes fp32 features for the sequence model.
#define POCR_CONVP(name, TH, MW, NS, WM, POOLH, POOLW, ACT, BN, MINW, BDIR)                                        \
    int name(ConvArgs a, hipStream_t st) {                                                                         \
        if constexpr (BDIR) {                                                                                      \
            if constexpr (NS == 1) {            /* (the persistent form of the 128-channel tiles does not fit the registers) */ \
                if (conv_rows_mode() == 2 && !a.x2)                                                                \
                    return launch_conv_rows(conv3x3_rows_kernel<TH, MW, NS, WM, POOLH, POOLW, ACT, BN, MINW, true>, TH, 16 * MW, \
                                            NS * (4 / WM) * 16, a, st);                                            \
            }                                                                                                      \
            if (conv_rows_mode() >= 1 && !a.x2)                                                                    \
                return launch_conv_rows(conv3x3_rows_kernel<TH, MW, NS, WM, POOLH, POOLW, ACT, BN, MINW, false>, TH, 16 * MW, \
                                        NS * (4 / WM) * 16, a, st);                                                \
        }                                                                                                          \
        return launch_conv(conv3x3_bf16x3_kernel<TH, MW, NS, WM, POOLH, POOLW, ACT, BN, MINW, BDIR, 3, 3, 1, 1, false, 2, true, true>, \
                           TH, 16 * MW, NS * (4 / WM) * 16, 256, a, st);                                           \
    }
// Tiles for the halo-row streaming main loop (conv_bf16x3.hpp ROWS; tools/conv_ablate.hip, profiles/r03_conv_rowstream.txt):
// 10 x 16 pixels x 64 channels per workgroup (each wave 160 pixels x 16 channels: ten row fragments per weight fragment, so the
// weight stream from L2 halves against the 80 x 32 wave tile) wherever the image is at least ten rows high - a tile then covers
// whole columns of the 10-row layers and no halo row is staged twice; 5 x 16 x 128 stays best for the 5-row layers.
POCR_CONVP(conv2_p2,  10, 1, 1, 1, 2, 2, ACT_RELU, false, 2, true)
POCR_CONVP(conv3_p2,  10, 1, 1, 1, 1, 1, ACT_RELU, false, 2, true)
POCR_CONVP(conv4_p2,  10, 1, 1, 1, 2, 2, ACT_RELU, false, 2, true)
POCR_CONVP(conv56_p2, 10, 1, 1, 1, 1, 1, ACT_RELU, false, 2, true)
POCR_CONVP(conv7_p2,  10, 1, 1, 1, 2, 1, ACT_RELU, false, 2, true)
POCR_CONVP(conv8_p2,  5, 1, 2, 1, 1, 1, ACT_LEAKY, false, 2, true)
// conv2 with conv1 computed in its prologue (conv_bf16x3.hpp FUSE1): the recogniser's default; conv1's activation is then never written
int conv2_p2_fused(ConvArgs a, hipStream_t st) {
    return launch_conv(conv3x3_bf16x3_kernel<10, 1, 1, 1, 2, 2, ACT_RELU, false, 2, true, 3, 3, 1, 1, false, 2, true, true, true>, 10, 16, 64, 256, a, st);
}
int conv2_p2_fused8(ConvArgs a, hipStream_t st) {      // 8 x 16 pixels, three workgroups per CU (52 KB of LDS each): networks without a recurrence (pocr_create)
    return launch_conv(conv3x3_bf16x3_kernel<8, 1, 1, 1, 2, 2, ACT_RELU, false, 3, true, 3, 3, 1, 1, false, 2, true, true, true>, 8, 16, 64, 256, a, st);
}
// (the round-2 tiles - 4 x 32 / 5 x 16 x 128 / 4 x 16 / 2 x 32 pixels for conv2 .. conv7 - lost to these on every layer and were removed in round 6:
// profiles/r03_conv_rowstream.txt, r05_conv_rows.txt)
POCR_CONVP(conv9_p2,  5, 1, 2, 1, 1, 1, ACT_LEAKY, true, 2, true)
#define POCR_CONVPG(name, TH, MW, NS, WM, ACT, MINW, KH, BDIR)                                                      \
    int name(ConvArgs a, hipStream_t st) {                                                                         \
        return launch_conv(conv3x3_bf16x3_kernel<TH, MW, NS, WM, 1, 1, ACT, false, MINW, BDIR, KH, 1, 0, 0, false, 2, true, false>, TH, \
                           16 * MW, NS * (4 / WM) * 16, 256, a, st);                                               \
    }
// 48 pixels x 128 channels, weights straight from L2, two workgroups per CU: 0.44 ms per c2 launch against 0.65 with the
// 48 x 256 LDS-weights configuration of the bf16x3 build and 0.57 / 0.51 / 0.78 for NT 256 / three workgroups / 80-pixel tiles
POCR_CONVPG(agg4_p2, 1, 3, 2, 1, ACT_LEAKY, 2, 4, true)
POCR_CONVPG(agg5_p2, 1, 3, 2, 1, ACT_LEAKY, 2, 5, true)
POCR_CONVPG(agg6_p2, 1, 3, 2, 1, ACT_LEAKY, 2, 6, true)
POCR_CONVPG(agg8_p2, 1, 3, 2, 1, ACT_LEAKY, 2, 8, true)
// decoder convs of the layout network: virtual cat(up2(x), skip) input
#define POCR_CONV3U(name, TH, MW, NS, WM, MINW, BDIR)                                                               \
    int name(ConvArgs a, hipStream_t st) {                                                                         \
        if (conv_split() == 2)                                                                                     \
            return launch_conv(conv3x3_bf16x3_kernel<TH, MW, NS, WM, 1, 1, ACT_RELU, false, MINW, BDIR, 3, 3, 1, 1, true, 2>, \
                               TH, 16 * MW, NS * (4 / WM) * 16, 256, a, st);                                       \
        return launch_conv(conv3x3_bf16x3_kernel<TH, MW, NS, WM, 1, 1, ACT_RELU, false, MINW, BDIR, 3, 3, 1, 1, true>, \
                           TH, 16 * MW, NS * (4 / WM) * 16, 256, a, st);                                           \
    }
// decoder layers with 256 / 128 output channels: 5 x 16 pixels x 64 channels, weights straight from L2 (f16x2: halo-row streaming);
// the 128-channel tile of round 3 kept 8 B of scratch per lane in its f16x2 build and is 3 % slower (profiles/r04_parsenet_up64.txt)
#define POCR_CONV3U2(name, TH, MW, NS, WM, MINW, BDIR, NS3)                                                         \
    int name(ConvArgs a, hipStream_t st) {                                                                         \
        if (conv_split() == 2)                                                                                     \
            return launch_conv(conv3x3_bf16x3_kernel<TH, MW, NS, WM, 1, 1, ACT_RELU, false, MINW, BDIR, 3, 3, 1, 1, true, 2>, \
                               TH, 16 * MW, NS * (4 / WM) * 16, 256, a, st);                                       \
        return launch_conv(conv3x3_bf16x3_kernel<TH, MW, NS3, WM, 1, 1, ACT_RELU, false, MINW, BDIR, 3, 3, 1, 1, true>, \
                           TH, 16 * MW, NS3 * (4 / WM) * 16, 256, a, st);                                          \
    }
POCR_CONV3U2(pn_up128_b3, 5, 1, 1, 1, 2, true, 2)   // f16x2: NT 64; bf16x3: NT 128 as before
// d1 / d0 (64 output channels): 10 x 16 pixels, weights straight from L2, halo-row streaming - 230 registers, no scratch; 1.86 ms per
// 4k x 3k page against 2.62 with the 4 x 64-pixel tile of round 3, which spilled 456 B per lane (profiles/r04_parsenet_up64.txt)
POCR_CONV3U(pn_up64_b_b3, 10, 1, 1, 1, 2, true)
POCR_CONV3U(pn_up64_a_b3, 4, 2, 2, 2, 2, false)     // 4 x 32 pixels, waves 2 (pixels) x 2 (channels), LDS weights (the bf16x3 build: row streaming is f16x2 only): 198 registers
int pn_up64_b3(ConvArgs a, hipStream_t st) { return conv_split() == 2 ? pn_up64_b_b3(a, st) : pn_up64_a_b3(a, st); }
POCR_CONV3G(agg4_b3, 1, 3, 4, 1, ACT_LEAKY, 1, 4, false)
POCR_CONV3G(agg5_b3, 1, 3, 4, 1, ACT_LEAKY, 1, 5, false)
POCR_CONV3G(agg6_b3, 1, 3, 2, 1, ACT_LEAKY, 1, 6, false)
POCR_CONV3G(agg8_b3, 1, 3, 2, 1, ACT_LEAKY, 1, 8, false)
POCR_CONV3G(gemm128_b3, 1, 8, 4, 2, ACT_NONE, 2, 1, false)
POCR_CONV3G(gemm128_relu_b3, 1, 8, 4, 2, ACT_RELU, 2, 1, false)
// (weights straight from L2 - BDIR - was measured slower for these: agg 1.04 vs 0.85 ms, c4 encoder 5.28 vs 5.11 ms)
const int kConvNT3[9] = {64, 64, 128, 128, 128, 128, 128, 128, 128};
// pixel-tile shape (TH, 16*MW) of conv1..conv9 and of the aggregation conv - the same numbers as in the tables above
// (kConvTH3: the bf16x3 configurations, which tile conv5 / conv6 differently)
const int kConvTH[10] = {4, 4, 4, 4, 10, 10, 10, 5, 5, 1};
const int kConvTH3[10] = {4, 4, 5, 4, 5, 5, 2, 5, 5, 1};
const int kConvTW[10] = {32, 64, 32, 32, 16, 16, 16, 16, 16, 48};
const int kConvTW3[10] = {32, 32, 16, 16, 16, 16, 32, 16, 16, 48};
const int kConvTHP[10] = {4, 10, 10, 10, 10, 10, 10, 5, 5, 1};       // the P2 configurations (POCR_CONVP)
const int kConvTWP[10] = {32, 16, 16, 16, 16, 16, 16, 16, 16, 48};
inline int conv_tile_h(bool p2, bool b3, int k, bool conv2_tile8 = false) { return p2 ? (k == 1 && conv2_tile8 ? 8 : kConvTHP[k]) : b3 ? kConvTH3[k] : kConvTH[k]; }
inline int conv_tile_w(bool p2, bool b3, int k) { return p2 ? kConvTWP[k] : b3 ? kConvTW3[k] : kConvTW[k]; }
// input width level of each conv (0: W_pad, 1: W_pad/2, 2: (W_pad/2)/2) and of its output
const int kConvLvlIn[10] = {0, 0, 1, 1, 2, 2, 2, 2, 2, 2};
const int kConvLvlOut[10] = {0, 1, 1, 2, 2, 2, 2, 2, 2, 2};
const int kAggNT = 256, kProjNT = 128, kHeadNT = 64, kSkinnyNT = 64;
const int kConvWaitLayer = 4;        // index into kConvPlan (conv5): see run_network
const int kLstmDecayAfter = 64;      // resident launches in a row without a hand-off timeout that halve the back-off again (sync_and_guard)

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// wsplit[tap][cin/32][cout16][plane][lane][8 x 16 bit] = plane of W(co = 16 s + (lane & 15), ci = 32 g + 8 (lane >> 4) + j, tap),
// zero outside cout_valid.  conv_split() == 3: hi / mid / lo of the exact bf16 truncation split; == 2: h = f16(w) and
// l = f16((w - h) * 2^11), both rounded to nearest (conv_bf16x3.hpp)
std::atomic<bool> g_f16_weight_overflow{false};      // a weight beyond f16's range met the f16x2 split (checked by the create calls)
inline void split_weight(float wv, int split, uint16_t *planes) {
    if (split == 2) {
        if (!(std::fabs(wv) <= 65504.0f)) g_f16_weight_overflow.store(true);
        const _Float16 h = (_Float16)wv;
        const _Float16 l = (_Float16)((wv - (float)h) * 2048.0f);
        memcpy(&planes[0], &h, 2); memcpy(&planes[1], &l, 2);
        return;
    }
    uint32_t wb; memcpy(&wb, &wv, 4);
    const uint32_t hb = wb & 0xffff0000u; float hf; memcpy(&hf, &hb, 4);
    const float r1 = wv - hf; uint32_t r1b; memcpy(&r1b, &r1, 4);
    const uint32_t mb = r1b & 0xffff0000u; float mf; memcpy(&mf, &mb, 4);
    const float r2 = r1 - mf; uint32_t r2b; memcpy(&r2b, &r2, 4);
    planes[0] = (uint16_t)(hb >> 16); planes[1] = (uint16_t)(mb >> 16); planes[2] = (uint16_t)(r2b >> 16);
}
std::vector<uint16_t> build_wsplit(int ntaps, int cin, int cout16, const std::function<float(int, int, int)> &W, int cout_valid) {
    const int split = conv_split();
    std::vector<uint16_t> wsp((size_t)ntaps * (cin / 32) * cout16 * split * 64 * 8);
    size_t o = 0;
    for (int tap = 0; tap < ntaps; ++tap)
        for (int g = 0; g < cin / 32; ++g)
            for (int sg = 0; sg < cout16; ++sg) {
                uint16_t part[3][64][8];
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 16 * sg + (lane & 15), ci = 32 * g + 8 * (lane >> 4) + j;
                        uint16_t pl[3] = {0, 0, 0};
                        split_weight(co < cout_valid ? W(co, ci, tap) : 0.f, split, pl);
                        part[0][lane][j] = pl[0]; part[1][lane][j] = pl[1]; part[2][lane][j] = pl[2];
                    }
                memcpy(&wsp[o], part, (size_t)split * 64 * 8 * sizeof(uint16_t));
                o += (size_t)split * 64 * 8;
            }
    return wsp;
}

}  // namespace

// Per-chunk state.  An engine owns POCR_NUM_SLOTS of these, each with its own HIP stream and
// activation buffers, so that chunk k+1 (conv backbone, MFMA-bound, fills the chip) overlaps the
// latency-bound tail of chunk k (2*T*L serial LSTM step launches, D2H, host decode).
struct Slot {
    hipStream_t stream = nullptr;        // uploads + conv backbone
    hipEvent_t conv_part = nullptr;      // recorded behind conv layer kConvWaitLayer of a launch: the NEXT launch's backbone may start then (run_network)
    bool conv_part_valid = false;
    hipStream_t seq_stream = nullptr;    // sequence model, head, CTC, D2H: high priority, so its short
                                         // latency-bound kernels are dispatched ahead of the other slot's conv workgroups
    // staged chunk
    DevBuf crops, lines;
    const uint8_t *crops_ext = nullptr;   // staged from device-resident crops (pocr_slot_stage_resident): conv1 reads them in place
    void *host_in = nullptr;         // pinned staging for the crop pool
    size_t host_in_cap = 0;
    int n = 0, w_pad = 0;            // w_pad = widest padded row of the staged lines
    bool staged = false, in_flight = false, want_logits = false, want_argmax = false;
    // Geometry of the staged lines.  Every line keeps the padded width of its reference chunk
    // (line_ocr_engine.py:79-90, 121-123): that width is part of the numerical contract, the grouping
    // of lines into device launches is not - all kernels work from these per-line tables.
    DevBuf geom;                     // one device blob, sub-allocated below
    std::vector<char> geom_host;
    const int32_t *g_lvl_w[3]{};     // [n] widths at the three levels
    const int64_t *g_act_off[9]{};   // [n+1] element offset of line i in the output of conv k
    const int64_t *g_feat_off = nullptr;   // [n+1] row_off * E
    const int32_t *g_line_T = nullptr;     // [n] frames of line i (= g_lvl_w[2], stable address)
    const int32_t *g_row_off = nullptr;    // [n+1] first frame (row) of line i in the sequence tensors
    const int32_t *g_slice_T = nullptr;    // [npad/16] longest line of each 16-line slice
    const int32_t *g_row_t = nullptr;      // [rows] frame index of every row inside its line
    const int32_t *g_row_line = nullptr;   // [rows] line of every row (the aggregation conv as a gathered GEMM, gemm_f16x2.hpp)
    const PixelTile *g_tiles[10]{};
    int g_ntiles[10]{};
    const FillSeg *g_fill = nullptr; // constant padding columns written by pad_fill_kernel instead of being convolved
    int g_nfill = 0;
    int64_t act_elems[9]{};          // total elements of every conv output
    int rows = 0, t_max = 0;         // sum of T_i, max T_i
    // activations
    DevBuf act[9], feat, xproj, hbuf, cbuf, logits, best, labels, lens;
    std::vector<DevBuf> lstm_y, sa_y;
    DevBuf sa_x, sa_x1, sa_qkv, sa_att, sa_tmp, sa_ff;
    // f16x2 range guard: [kRangeSets][8] words, the largest |value| every producer of an f16x2 operand wrote in this launch
    DevBuf range;
    unsigned *range_host = nullptr;    // pinned copy, read at collect time
    bool guard_checked = false, redirect = false;      // redirect: this launch was re-run on the fall-back engine; its results live in shadow->slot[k]
    std::vector<int64_t> st_off;       // what was staged (per line): crop offset, width, padded width, left padding - the fall-back engine stages the same lines
    std::vector<int32_t> st_w, st_wpad, st_padl, sp_rows_host;
    bool feat_is_p2 = false;           // this launch's aggregated features are stored in the P2 layout (pocr_debug_read converts)
    int lstm_p2_layers = 0;            // this launch's BiLSTM layers 0 .. lstm_p2_layers - 1 wrote their output in the P2 layout
    DevBuf sa_xp2, sa_x1p2, feat_p2;   // P2 (pre-split) copies that feed the persistent GEMM (gemm_f16x2.hpp): LayerNorm outputs; the style-embedded features
    int act_h[9]{}, act_w[9]{}, act_c[9]{};
    // host pinned staging for the outputs
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    // device-side CSC sparsification of the logits (sparsify.hpp)
    bool want_sparse = false;
    float sp_thr = 1e-4f;
    DevBuf sp_rowstat, sp_colcount, sp_line_nnz, sp_line_off, sp_indptr, sp_data, sp_indices, sp_rows, sp_conf, sp_bid, sp_bp;   // sp_colcount: [n][frame blocks][C]
    bool sp_has_rows = false;
    size_t sp_spec = 0;              // entries copied back speculatively at launch time
    void *sp_pinned = nullptr;       // [line_off (n+1) int64 | indptr n*(C+1) int32] then data | indices at collect time
    size_t sp_pinned_cap = 0;
    // profiling
    hipEvent_t ev[POCR_NUM_STAGES + 1]{};
    hipEvent_t ev_conv_end = nullptr;         // end of the aggregation conv on the conv stream (the sequence stage may start later: deferred)
    float stage_ms[POCR_NUM_STAGES]{};
    bool have_ms = false;
    // BiLSTM recurrence as replayable hipGraphs: key (layer, T, slice bucket) -> 2 memsets + T step launches.
    // Node parameters hold this slot's buffer addresses; any re-allocation of those buffers flushes the cache.
    std::map<std::tuple<int, int, int>, hipGraphExec_t> lstm_graphs;
    DevBuf lstm_dims;                // device {n, npad} read by the replayed step kernels
    int32_t *lstm_dims_host = nullptr;   // pinned source of that copy
    DevBuf lstm_sync;                    // resident recurrence (lstm_resident.hpp): [clusters][32] sync words, then 4 error / diagnostic words
    size_t lstm_err_off = 0;             // index (uint32) of the error words inside lstm_sync
    bool lstm_resident_used = false;     // this launch ran the resident kernel: collect checks the error word
    bool lstm_judged = false;            // ... and sync_and_guard has counted its outcome for the back-off (once per launch)
    bool lstm_force_step = false;        // the repeat of a launch whose resident recurrence timed out: step kernels
    uint32_t *lstm_err_host = nullptr;   // pinned copy of the error words
    DevBuf nf_flag;                      // set by frame_argmax_kernel when a winning logit is NaN / inf
    int32_t *nf_host = nullptr;          // pinned copy, read at collect time
    size_t h_stride = 0;             // floats between the two h ping-pong buffers (capacity-based, stable)
    const void *graph_geom = nullptr;    // address of `seqgeom` the cached graphs were captured with
    // Sequence-part tables (frames per line, first row, per-slice maximum) live in their own buffer with a
    // capacity-based layout, so their addresses - baked into the cached LSTM graphs - stay put while the
    // number of staged lines varies: [line_T: cap] [row_off: cap + 16] [slice_T: cap / 16 + 16].
    DevBuf seqgeom;
    int sg_cap = 0;
    std::vector<int32_t> sg_host;
    size_t h_lvl2_off = 0;           // byte offset of the per-line frame counts inside geom_host
    // recorded after the conv backbone of a launch: the next launch (on another slot) starts its own
    // MFMA-bound backbone only then, so backbones run one after the other at full speed and only the
    // latency-bound sequence tail of the previous chunk shares the chip with them
    hipEvent_t conv_done = nullptr;
    bool conv_done_valid = false;

    // sequence-to-sequence decoding (POCR_ARCH_S2S, decoder.hpp)
    std::vector<DevBuf> s2s_kv, s2s_cache;   // per decoder layer: projected encoder output [rows][2E]; self cache [S_cap][n][3E]
    DevBuf s2s_x, s2s_x1, s2s_x2, s2s_t, s2s_ctx, s2s_q, s2s_ff, s2s_logits, s2s_tokens, s2s_state, s2s_tables;
    std::vector<int32_t> s2s_batch_first, s2s_limit, s2s_steps, s2s_wpads;
    int s2s_batches = 0, s2s_cap = 0, s2s_smax = 0;
    bool s2s_launched = false, s2s_decoded = false, s2s_want_logits = false;
    void *s2s_pinned = nullptr;      // [tokens n*S_cap int32 | logits n*s_max*C float]
    size_t s2s_pinned_cap = 0;
    int32_t *s2s_flags = nullptr;    // pinned: [remaining per polled block (64) | steps (n_batches)]
    size_t s2s_flags_cap = 0;
    hipEvent_t s2s_ev[2]{};
};

struct pocr_engine {
    pocr_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;    // set-up stream (weight uploads)
    // weights (device)
    DevBuf conv_w[9], conv_b[9], bn_scale, bn_shift, agg_w, agg_b, head_w, head_b, lut;
    std::vector<float> embed_table;  // style embeddings [embed_num + 1][2E] (host copy); embed_ss = (1 + scale | shift) of the chosen row
    DevBuf embed_ss;
    int embed_id = -1;
    std::unordered_set<const void *> b3_weights;   // weight buffers laid out for the bf16x3 kernels (wsplit): the GEMM-mode / aggregation launches ask
    std::vector<DevBuf> proj_w, proj_b, whh;       // per LSTM layer
    std::vector<DevBuf> whh2;                      // per LSTM layer: W_hh as f16x2 fragments (lstm.hpp: lstm_gemm_f16x2); empty = fp32 MFMA recurrence
    // self-attention encoder (POCR_ARCH_SA): per layer in_proj, out_proj, lin1, lin2 (fragment order) + LN params
    struct SaLayer { DevBuf w_in, b_in, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b; };
    std::vector<SaLayer> sa;
    DevBuf sa_nw, sa_nb, pe;
    int pe_rows = 0;
    // decoder (POCR_ARCH_S2S): DecoderLayer weights (transformer.py:388-411); in_proj of the memory attention is
    // split into its query rows [0, E) and key/value rows [E, 3E) (cached_forward :237-247, :259-268)
    struct DecLayer {
        DevBuf ws_in, bs_in, ws_out, bs_out, wc_q, bc_q, wc_kv, bc_kv, wc_out, bc_out, w1, b1, w2, b2;
        DevBuf n1w, n1b, n2w, n2b, n3w, n3b;
    };
    std::vector<DecLayer> dec;
    DevBuf dec_embed, dec_out_w, dec_out_b;
    int dec_out_cout16 = 0;
    int lstm_capacity[3][3] = {{-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};   // [KPW 1/2/4][SL 1/2/4]: workgroups of that instantiation of the resident recurrence the chip holds at once (occupancy query, first use)
    bool lstm_resident = true;       // one launch per BiLSTM layer with the hidden state handed over inside an XCD (lstm_resident.hpp); POCR_LSTM_RESIDENT=0: one launch per step
    // A hand-off timeout (a cluster that was not fully resident: other tenants on the chip) sends the NEXT lstm_skip launches to the
    // step kernels, then the resident path is tried again; every further timeout doubles the pause (4 .. 256 launches).
    // The pause decays: every kLstmDecayAfter resident launches in a row that hand over in time halve it again (below 4: gone), so a
    // long-lived server pays for a handful of sporadic timeouts once, not for the rest of its life.  The counters are
    // touched by run_network / sync_and_guard of several slots, which may run on worker threads (decoding loops): atomics.
    std::atomic<int> lstm_skip{0}, lstm_skip_len{0}, lstm_ok_run{0};
    std::atomic<int64_t> lstm_timeouts{0};       // launches repeated on the step kernels after a timeout (pocr_lstm_timeouts)
    int lstm_spin_limit = 1 << 22;   // POCR_LSTM_SPIN_LIMIT at creation (tests force the timeout path with a tiny limit)
    bool warned_nonfinite = false, warned_placement = false, warned_range = false;
    bool fuse12 = false;             // conv1 inside conv2's prologue (P2 only; POCR_NO_FUSE12=1: conv1 as its own launch, its activation in HBM)
    bool conv2_tile8 = false;        // the fused conv1+2 kernel as 8 x 16 tiles, three workgroups per CU (networks without a recurrence)
    DevBuf conv1_w2;                 // conv1's weights as f16x2 fragments (Conv1Args::w1x2)
    std::atomic<pocr_engine *> shadow{nullptr};   // f16x2 range guard: the same network on bf16x3 (fp32's range); built by shadow_builder behind pocr_create
                                     // (POCR_FALLBACK_EAGER=0: when a launch first leaves f16's range)
    std::mutex shadow_mu;            // creation of / launches on the fall-back engine (decoding loops of several slots run on worker threads)
    std::thread shadow_builder;      // holds shadow_mu while it builds: a fall-back that comes earlier waits for it instead of building twice
    std::atomic<int> shadow_state{0};        // 0 not asked for, 1 being built in the background, 2 there, -1 the background build failed (built again on demand, where the error can be reported)
    std::vector<float> weights_host; // the weight blob (kept for the fall-back engine; f16x2 engines only)
    int64_t range_fallbacks = 0;     // launches re-run on the fall-back engine
    bool is_shadow = false;
    int n_cus = 256;                 // compute units (grid of the persistent GEMM)
    bool gemm2 = false;              // GEMM-shaped layers on the persistent 256 x 128 kernel with P2 inputs (gemm_f16x2.hpp; needs p2; POCR_NO_GEMM2=1: conv3x3_bf16x3_kernel's GEMM mode on fp32 activations)
    bool p2 = false;                 // f16x2 with pre-split activations between conv1 and the aggregation conv (conv_bf16x3.hpp "P2"; POCR_NO_P2=1: split in every consumer)
    DevBuf head_w2, head_b2;         // the output layer's weights in the wsplit layout + its bias padded to 128 columns: the head on the persistent GEMM
    int head2_cout16 = 0;
    bool head_fp32 = false;          // POCR_HEAD_FP32 at creation: the head on the fp32-MFMA GEMM (A / B test of the f16x2 head)
    bool bf16x3 = true;              // conv2..conv9 on the bf16 matrix pipe with the exact 3-way operand split (POCR_CONV_FP32=1: fp32 MFMA)
    DevBuf cconst[9];                // per conv layer: the output column [H_out][cout] far inside zero padding
    bool pad_skip = false;           // skip + fill constant padding tiles (POCR_NO_PAD_SKIP=1 turns it off)
    bool cconst_ready = false;       // the constants are computed when a launch first needs them
    int conv_cout16[9]{};
    int agg_cout16 = 0, head_cout16 = 0, proj_cout16 = 0;
    Slot slot[POCR_NUM_SLOTS + 1];   // the last one is internal (padding-column constants), not reachable through the API
    int last_slot = 0;               // slot of the most recent launch (stage timings / debug taps)
    size_t sp_prev_total = 0;        // kept entries of the most recent sparse launch (sizes the next speculative copy)
    int sp_prev_rows = 0;            // ... and its frames
    bool use_graphs = true;          // replay the LSTM recurrence from captured hipGraphs (POCR_NO_GRAPHS=1 disables)
    bool profiling = false;
    Comm comm;                       // RCCL communicator of the multi-GPU path (comm.hpp); inactive on a single GPU
};

namespace {

int upload_u16(DevBuf &b, const std::vector<uint16_t> &v, hipStream_t st) {
    if (b.reserve(v.size() * sizeof(uint16_t))) return 1;
    HIP_TRY(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int upload(DevBuf &b, const std::vector<float> &v, hipStream_t st) {
    if (b.reserve(v.size() * sizeof(float))) return 1;
    HIP_TRY(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int check_cfg(const pocr_config *c) {
    if (!c) return fail("config is NULL");
    if (c->abi_version != POCR_ABI_VERSION) return fail("ABI version mismatch: got %d, library is %d", c->abi_version, POCR_ABI_VERSION);
    if (c->height <= 0 || c->height % 8) return fail("height must be a positive multiple of 8 (got %d)", c->height);
    const int ah = c->height / 8;
    if (ah != 4 && ah != 5 && ah != 6 && ah != 8) return fail("unsupported height %d (aggregation height %d; built for 32/40/48/64)", c->height, ah);
    if (c->num_classes < 2) return fail("num_classes must be >= 2");
    if (c->embed_num < 0 || (c->embed_num > 0 && c->arch == POCR_ARCH_S2S)) return fail("embed_num must be >= 0 (0 for the seq2seq engine)");
    if (c->conv_out <= 0 || c->conv_out % 16) return fail("conv_out must be a positive multiple of 16");
    if (c->arch == POCR_ARCH_SA || c->arch == POCR_ARCH_S2S) {
        if (c->arch == POCR_ARCH_S2S && c->dec_layers < 1) return fail("dec_layers must be >= 1");
        if (c->arch == POCR_ARCH_S2S && c->num_classes < 3) return fail("a seq2seq model needs >= 3 classes (symbol, boundary, ignore)");
        if (c->sa_layers < 1) return fail("sa_layers must be >= 1");
        if (c->sa_heads < 1 || c->conv_out % c->sa_heads) return fail("conv_out must be divisible by sa_heads");
        const int d = c->conv_out / c->sa_heads;
        if (d != 32 && d != 64 && d != 128) return fail("head dim conv_out/sa_heads must be 32, 64 or 128 (got %d)", d);
        if (c->sa_ff <= 0 || c->sa_ff % 16) return fail("sa_ff must be a positive multiple of 16");
        if (c->conv_out > 1024) return fail("conv_out must be <= 1024 for the self-attention encoder");
    } else if (c->arch == POCR_ARCH_BLSTM) {
        if (c->lstm_hidden <= 0 || c->lstm_hidden % 16) return fail("lstm_hidden must be a positive multiple of 16");
        if (c->lstm_layers < 1) return fail("lstm_layers must be >= 1");
    } else {
        return fail("unknown arch id %d", c->arch);
    }
    return 0;
}

struct WeightCursor {
    const float *p;
    const float *take(size_t n) { const float *r = p; p += n; return r; }
};

// feat [rows][E] <- feat * ss[c] + ss[E + c]: two roundings per element like torch's `f * (1 + s) + b` (no FMA contraction)
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void style_embed_kernel(float *feat, const float *ss, size_t total, int E) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * 1024) {
        f32x4 v = *reinterpret_cast<f32x4 *>(feat + i);
        const int c = (int)(i % (size_t)E);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + c), sh = *reinterpret_cast<const f32x4 *>(ss + E + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float m = v[j] * sc[j]; v[j] = m + sh[j]; }
        *reinterpret_cast<f32x4 *>(feat + i) = v;
    }
}
#pragma clang fp contract(fast)

__global__ __launch_bounds__(256) void zero_fill_kernel(f32x4 *p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

int launch_pad_fill(pocr_engine *e, Slot &s, hipStream_t st, int only_layer, int skip_layer) {
    if (s.g_nfill <= 0) return 0;
    FillArgs fa{};
    int hh = e->cfg.height;
    for (int i = 0; i < 9; ++i) {
        hh /= kConvPlan[i].ph;
        fa.act[i] = s.act[i].as<float>(); fa.cvec[i] = e->cconst[i].as<float>(); fa.out_off[i] = s.g_act_off[i];
        fa.lvl_out[i] = kConvLvlOut[i]; fa.h_out[i] = hh; fa.cout[i] = kConvPlan[i].cout;
    }
    for (int k = 0; k < 3; ++k) fa.lvl_w[k] = s.g_lvl_w[k];
    fa.segs = s.g_fill;
    fa.only_layer = only_layer; fa.skip_layer = skip_layer;
    hipLaunchKernelGGL(pad_fill_kernel, dim3(s.g_nfill, e->cfg.height), dim3(256), 0, st, fa);
    HIP_TRY(hipGetLastError());
    return 0;
}

// conv1 as its own launch (every mode but the fused default; pocr_debug_read(0) runs it on demand in the fused mode)
int launch_conv1(pocr_engine *e, Slot &s, hipStream_t st, unsigned *range = nullptr) {
    Conv1Args c1{};
    c1.range_max = range;
    c1.crops = s.crops_ext ? s.crops_ext : s.crops.as<uint8_t>(); c1.lines = s.lines.as<LineDesc>(); c1.lut = e->lut.as<float>();
    c1.wfrag = e->conv_w[0].as<float>(); c1.bias = e->conv_b[0].as<float>(); c1.y = s.act[0].as<float>();
    c1.w1x2 = e->conv1_w2.p;
    c1.tiles = s.g_tiles[0]; c1.line_w = s.g_lvl_w[0]; c1.out_off = s.g_act_off[0];
    c1.H = e->cfg.height; c1.n_ptiles = s.g_ntiles[0];
    if (c1.n_ptiles > 0) {
        const bool x2 = conv_split() == 2;           // f16x2 builds: conv1 in the same arithmetic as the fused prologue
        if (e->p2) hipLaunchKernelGGL((conv1_u8_kernel<true, true>), dim3(c1.n_ptiles), dim3(256), 0, st, c1);
        else if (x2) hipLaunchKernelGGL((conv1_u8_kernel<false, true>), dim3(c1.n_ptiles), dim3(256), 0, st, c1);
        else hipLaunchKernelGGL((conv1_u8_kernel<false, false>), dim3(c1.n_ptiles), dim3(256), 0, st, c1);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// The persistent P2-input GEMM (gemm_f16x2.hpp) for a [rows][cin] x [cin][cout] product whose weights are f16x2 wsplit fragments
// with cout16 groups of 16 columns.  `can_gemm2` tells whether the shape fits (else: conv3x3_bf16x3_kernel's GEMM mode).
inline bool can_gemm2(const pocr_engine *e, int cin, int cout16, int nk, int cout_valid, bool p2out) {
    return e->gemm2 && cin % 32 == 0 && cout16 % (kGemmBN / 16) == 0 && cout16 * 16 <= kGemmBiasMax && nk >= 3 &&
           (!p2out || cout_valid % 32 == 0);
}
inline void gemm2_shape(const pocr_engine *e, GemmP2Args &g, int rows, int cout16) {
    g.mt_total = (rows + kGemmBM - 1) / kGemmBM;
    g.nt_total = cout16 * 16 / kGemmBN;
    // column tiles per XCD block: every XCD sees every column tile (measured: 2 / 4 / 8-tile groups change nothing, profiles/r04_gemm_dma_xcd_map.txt)
    g.nb = g.nt_total;
    (void)e;
}
template <int ACT, bool P2OUT, bool GATHER>
int launch_gemm2(const pocr_engine *e, GemmP2Args g, hipStream_t st) {
    if (g.M <= 0) return 0;
    const int grid = gemm_f16x2_grid(g.M, g.N16 * 16, e->n_cus);
    hipLaunchKernelGGL((gemm_f16x2_kernel<ACT, P2OUT, GATHER>), dim3(grid), dim3(kGemmThreads), 0, st, g);
    HIP_TRY(hipGetLastError());
    return 0;
}

int run_network(pocr_engine *e, Slot &s) {
    const pocr_config &c = e->cfg;
    hipStream_t st = s.stream;          // switches to s.seq_stream after the backbone
    const int n = s.n, H = c.height;
    const bool prof = e->profiling;
    auto mark = [&](int i) { if (prof) (void)hipEventRecord(s.ev[i], st); };
    const int T = s.t_max, E = c.conv_out, AH = H / 8, rows = s.rows;

    // ---- conv stack.  If the previous launch ran on another slot, this backbone starts behind that one's conv5
    // (kConvWaitLayer): the first layers of this launch then share the chip with conv6 .. conv9 + aggregation of the one
    // ahead and fill the ends of its kernels - every kernel of a stream leaves the CUs half empty while its last round of
    // workgroups finishes (conv6 .. conv9: 18 rounds of 45-95 us tiles) and the next one cannot start before it has.
    // Measured on one box (profiles/r04_backbone_overlap.txt): c2 9.73 -> 9.39 ms per step, c4 +1.2 %, c3 / c5 unchanged;
    // no wait at all is as good on c2 and costs the c3 stream 3 %; behind the WHOLE backbone (rounds 1-3) two backbones never share
    // the chip.
    constexpr int conv_wait_layer = kConvWaitLayer;
    {
        Slot &prev = e->slot[e->last_slot];
        if (&prev != &s && prev.conv_done_valid) {
            if (prev.conv_part_valid) HIP_TRY(hipStreamWaitEvent(st, prev.conv_part, 0));
            else HIP_TRY(hipStreamWaitEvent(st, prev.conv_done, 0));
        }
    }
    s.conv_part_valid = false;
    int h = H;
    for (int i = 0; i < 9; ++i)
        if (s.act[i].reserve((size_t)s.act_elems[i] * sizeof(float))) return 1;
    // f16x2 range guard: the words of this launch start at zero
    const bool guard = conv_split() == 2;
    if (guard) {
        if (!s.range.p) {
            if (s.range.reserve(kRangeWords * sizeof(unsigned))) return 1;
            HIP_TRY(locked_host_malloc(reinterpret_cast<void **>(&s.range_host), kRangeWords * sizeof(unsigned), hipHostMallocDefault));
        }
        HIP_TRY(hipMemsetAsync(s.range.p, 0, kRangeWords * sizeof(unsigned), st));
    }
    s.guard_checked = !guard; s.redirect = false;
    auto rset = [&](int k) -> unsigned * { return guard ? s.range.as<unsigned>() + 8 * k : nullptr; };
    // constant padding columns of all nine layers in one launch (conv_igemm.hpp: pad_fill_kernel); conv1's activation does not
    // exist in the fused mode (pocr_debug_read(0) fills and computes it on demand)
    if (launch_pad_fill(e, s, st, -1, e->fuse12 ? 0 : -1)) return 1;
    for (int i = 0; i < 9; ++i) {
        const ConvLayer &L = kConvPlan[i];
        ConvArgs a{};
        a.H = h; a.Ho = h;
        a.tiles = s.g_tiles[i]; a.n_ptiles = s.g_ntiles[i];
        a.line_w = s.g_lvl_w[kConvLvlIn[i]];
        a.in_off = i ? s.g_act_off[i - 1] : s.g_act_off[0];      // conv1 reads the u8 crops, not an fp32 image
        a.out_off = s.g_act_off[i];
        a.cout16 = e->conv_cout16[i]; a.cout_valid = L.cout; a.out_stride = L.cout;
        a.wfrag = e->conv_w[i].as<float>(); a.bias = e->conv_b[i].as<float>();
        a.y = s.act[i].as<float>();
        a.range_max = rset(i); a.f1_range = rset(0);
        mark(i);
        int rc = 0;
        if (i == 0) {
            if (!e->fuse12) rc = launch_conv1(e, s, st, rset(0));
        } else {
            a.x = s.act[i - 1].as<float>(); a.cin = L.cin;
            if (i == 8) { a.bn_scale = e->bn_scale.as<float>(); a.bn_shift = e->bn_shift.as<float>(); }
            if (e->p2) {
                g_conv_layer = i;                       // (conv_rows_mode: which form of the kernel this layer takes)
                switch (i) {
                    case 1:
                        if (e->fuse12) {
                            a.f1_crops = s.crops_ext ? s.crops_ext : s.crops.as<uint8_t>(); a.f1_lines = s.lines.as<LineDesc>();
                            a.f1_lut = e->lut.as<float>(); a.f1_w = e->conv1_w2.p; a.f1_bias = e->conv_b[0].as<float>(); a.f1_src_h = 0;
                            rc = e->conv2_tile8 ? conv2_p2_fused8(a, st) : conv2_p2_fused(a, st);
                        } else {
                            rc = conv2_p2(a, st);
                        }
                        break;
                    case 2: rc = conv3_p2(a, st); break;
                    case 3: rc = conv4_p2(a, st); break;
                    case 4: case 5: rc = conv56_p2(a, st); break;
                    case 6: rc = conv7_p2(a, st); break;
                    case 7: rc = conv8_p2(a, st); break;
                    default: rc = conv9_p2(a, st); break;
                }
            } else if (e->bf16x3) {
                switch (i) {
                    case 1: rc = conv2_b3(a, st); break;
                    case 2: rc = conv3_b3(a, st); break;
                    case 3: rc = conv4_b3(a, st); break;
                    case 4: case 5: rc = conv56_b3(a, st); break;
                    case 6: rc = conv7_b3(a, st); break;
                    case 7: rc = conv8_b3(a, st); break;
                    default: rc = conv9_b3(a, st); break;
                }
            } else {
            switch (i) {
                case 1: rc = conv2_k(a, st); break;
                case 2: rc = conv3_k(a, st); break;
                case 3: rc = conv4_k(a, st); break;
                case 4: case 5: rc = conv56_k(a, st); break;
                case 6: rc = conv7_k(a, st); break;
                case 7: rc = conv8_k(a, st); break;
                default: rc = conv9_k(a, st); break;
            }
            }
        }
        if (rc) return rc;
        h /= L.ph;
        s.act_h[i] = h; s.act_c[i] = L.cout;
        if (i == conv_wait_layer) { HIP_TRY(hipEventRecord(s.conv_part, st)); s.conv_part_valid = true; }
    }
    // ---- aggregation conv: line i [H/8][T_i][512] -> rows row_off[i] .. of feat [rows][E]
    {
        if (s.feat.reserve((size_t)rows * E * sizeof(float))) return 1;
        ConvArgs a{};
        a.x = s.act[8].as<float>(); a.H = AH; a.Ho = 1; a.cin = 512;
        a.tiles = s.g_tiles[9]; a.n_ptiles = s.g_ntiles[9];
        a.line_w = s.g_lvl_w[2]; a.in_off = s.g_act_off[8]; a.out_off = s.g_feat_off;
        a.cout16 = e->agg_cout16; a.cout_valid = E; a.out_stride = E;
        a.wfrag = e->agg_w.as<float>(); a.bias = e->agg_b.as<float>(); a.y = s.feat.as<float>();
        a.range_max = rset(9);
        mark(POCR_STAGE_AGG);
        int rc;
        // the features feed a projection GEMM directly (BiLSTM, no style embedding): written in the P2 layout, once
        s.feat_is_p2 = false;
        if (e->p2 && can_gemm2(e, 512, e->agg_cout16, AH * 16, E, false)) {
            GemmP2Args g{};
            g.a = s.act[8].p; g.w = e->agg_w.p; g.bias = e->agg_b.as<float>(); g.y = s.feat.p;
            g.M = rows; g.nk = AH * 16; g.N16 = e->agg_cout16; g.n_valid = E; g.ldy = E;
            gemm2_shape(e, g, rows, e->agg_cout16);
            g.row_line = s.g_row_line; g.row_t = s.g_row_t; g.line_w = s.g_lvl_w[2]; g.in_off = s.g_act_off[8];
            g.cpt = 16; g.ntap = AH; g.cin = 512; g.range_flag = rset(9);
            s.feat_is_p2 = c.arch == POCR_ARCH_BLSTM && c.embed_num == 0 && E % 32 == 0 && can_gemm2(e, E, e->proj_cout16, E / 32, 8 * c.lstm_hidden, false);
            rc = s.feat_is_p2 ? launch_gemm2<ACT_LEAKY, true, true>(e, g, st) : launch_gemm2<ACT_LEAKY, false, true>(e, g, st);
        } else if (e->p2) rc = AH == 4 ? agg4_p2(a, st) : AH == 5 ? agg5_p2(a, st) : AH == 6 ? agg6_p2(a, st) : agg8_p2(a, st);
        else if (e->b3_weights.count(e->agg_w.p)) rc = AH == 4 ? agg4_b3(a, st) : AH == 5 ? agg5_b3(a, st) : AH == 6 ? agg6_b3(a, st) : agg8_b3(a, st);
        else rc = AH == 4 ? agg4_k(a, st) : AH == 5 ? agg5_k(a, st) : AH == 6 ? agg6_k(a, st) : agg8_k(a, st);
        if (rc) return rc;
        if (c.embed_num > 0) {      // f * (1 + scale) + shift, the chosen style row for every line (pytorch_ocr_engine.py:64-66)
            if (e->embed_id < 0) return fail("this model has an embeddings layer: pocr_set_embed_id first");
            const size_t total = (size_t)rows * E;
            hipLaunchKernelGGL(style_embed_kernel, dim3((unsigned)std::min<size_t>(4096, (total / 4 + 255) / 256)), dim3(256), 0, st,
                               s.feat.as<float>(), e->embed_ss.as<float>(), total, E);
            HIP_TRY(hipGetLastError());
        }
    }
    if (prof) (void)hipEventRecord(s.ev_conv_end, st);
    mark(POCR_STAGE_LSTM);                // (on the conv stream: the sequence stage begins with the first projection, below)
    // The FIRST BiLSTM layer's input projection depends on the features only: it runs HERE, on the conv stream, behind the
    // aggregation conv and before the next launch's backbone may start - alone on the chip, 0.27 ms for a c2 chunk - instead
    // of on the sequence stream, where its 256 persistent workgroups (147 KB of LDS each: a CU must drain both of its conv
    // workgroups before one fits) took 0.55 ms next to the other slot's convolutions and held the recurrence back
    // (profiles/r04_bench_c2_kernel_stats.txt).
    bool proj0_done = false, xproj_moved = false;
    if (c.arch == POCR_ARCH_BLSTM && s.feat_is_p2) {
        const int Hh0 = c.lstm_hidden;
        const void *xp0 = s.xproj.p;
        if (s.xproj.reserve((size_t)rows * 8 * Hh0 * sizeof(float))) return 1;
        xproj_moved = xp0 != s.xproj.p;
        GemmP2Args g{};
        g.a = s.feat.p; g.w = e->proj_w[0].p; g.bias = e->proj_b[0].as<float>(); g.y = s.xproj.p;
        g.M = rows; g.nk = E / 32; g.N16 = e->proj_cout16; g.n_valid = 8 * Hh0; g.ldy = 8 * Hh0; g.lda = (int64_t)E * 4;
        gemm2_shape(e, g, rows, e->proj_cout16);
        if (launch_gemm2<ACT_NONE, false, false>(e, g, st)) return 1;
        proj0_done = true;
    }
    // The sequence stage runs on the slot's second stream, so that the next launch's backbone shares the chip with it.  (Keeping it on the
    // conv stream - no two launches ever share the chip - was measured for config 4, whose two halves are both matrix-pipe work: equal
    // within 0.5 %)
    HIP_TRY(hipEventRecord(s.conv_done, st));
    s.conv_done_valid = true;
    HIP_TRY(hipStreamWaitEvent(s.seq_stream, s.conv_done, 0));
    st = s.seq_stream;
    const float *layer_in = s.feat.as<float>();
    int din = E;
    // head2: the output layer on the persistent f16x2 GEMM, reading the last sequence layer's output in P2 (POCR_HEAD_FP32=1:
    // the fp32-MFMA GEMM on the fp32 output)
    const void *layer_in_p2 = nullptr;
    auto head2_ok = [&](int din_) {
        return e->gemm2 && !e->head_fp32 && e->head_w2.p && can_gemm2(e, din_, e->head2_cout16, din_ / 32, c.num_classes, false);
    };
    if (c.arch == POCR_ARCH_SA || c.arch == POCR_ARCH_S2S) {
    // ---- self-attention encoder (transformer.py:366-385)
    const int FF = c.sa_ff, heads = c.sa_heads, D = E / heads;
    const size_t xe = (size_t)rows * E * sizeof(float);
    if (s.sa_x.reserve(xe) || s.sa_x1.reserve(xe) || s.sa_att.reserve(xe) || s.sa_tmp.reserve(xe)) return 1;
    if (s.sa_qkv.reserve(3 * xe) || s.sa_ff.reserve((size_t)rows * FF * sizeof(float))) return 1;
    // the decoder adds pe[step] while OTHER slots' launches run (worker threads): the table gets its largest size - every step the
    // decoder can take, every frame a staged line can have - on the first launch and is not replaced under a running decode
    const int pe_need = std::max(T, c.arch == POCR_ARCH_S2S ? DEC_MAX_KEYS + 8 : 1024);
    if (pe_need > e->pe_rows) {      // sinusoidal table, float32 like PositionalEncoding (transformer.py:316-332)
        const int rows_pe = round_up(pe_need, 256);
        std::vector<float> pe((size_t)rows_pe * E);
        for (int k = 0; k < E; k += 2) {
            const float div = expf((float)k * (-logf(10000.0f) / (float)E));
            for (int t = 0; t < rows_pe; ++t) {
                pe[(size_t)t * E + k] = sinf((float)t * div);
                if (k + 1 < E) pe[(size_t)t * E + k + 1] = cosf((float)t * div);
            }
        }
        HIP_TRY(locked_device_sync());     // another slot may still be reading the old table
        if (upload(e->pe, pe, st)) return 1;
        e->pe_rows = rows_pe;
    }
    // The persistent GEMM (gemm_f16x2.hpp) reads P2 (pre-split) activations: LayerNorm writes a P2 copy next to the fp32
    // residual, attention and the first feed-forward linear write P2 only (their outputs feed nothing but the next GEMM).
    const bool g2 = e->gemm2 && E % 32 == 0 && FF % 32 == 0 &&
                    can_gemm2(e, E, round_up(3 * E, kProjNT) / 16, E / 32, 3 * E, false) && can_gemm2(e, E, round_up(E, kProjNT) / 16, E / 32, E, false) &&
                    can_gemm2(e, E, round_up(FF, kProjNT) / 16, E / 32, FF, true) && can_gemm2(e, FF, round_up(E, kProjNT) / 16, FF / 32, E, false) &&
                    e->b3_weights.count(e->sa[0].w_in.p) != 0;
    if (g2 && (s.sa_xp2.reserve(xe) || s.sa_x1p2.reserve(xe))) return 1;
    auto ln = [&](const float *a_, const float *b_, const DevBuf &gw, const DevBuf &gb, const float *pe_, float *y_, void *y2_) {
        hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a_, b_, gw.as<float>(), gb.as<float>(),
                           pe_, y_, rows, E, T, 1e-5f, s.g_row_t, (const int32_t *)nullptr, y2_, rset(kRangeOther));
    };
    // x_: fp32 activations (old kernels) or, with g2, their P2 form; p2out: the output in P2 (g2 only)
    auto gemm = [&](const void *x_, int cin_, const DevBuf &w_, const DevBuf &b_, int cout_, void *y_, bool relu, bool p2in, bool p2out) {
        if (p2in) {
            GemmP2Args g{};
            g.a = x_; g.w = w_.p; g.bias = b_.as<float>(); g.y = y_;
            g.M = rows; g.nk = cin_ / 32; g.N16 = round_up(cout_, kProjNT) / 16; g.n_valid = cout_; g.ldy = cout_; g.lda = (int64_t)cin_ * 4;
            gemm2_shape(e, g, rows, g.N16);
            g.range_flag = rset(kRangeOther);     // (every output: only the feed-forward one feeds another f16x2 GEMM, the others are O(that) anyway)
            if (p2out) return relu ? launch_gemm2<ACT_RELU, true, false>(e, g, st) : launch_gemm2<ACT_NONE, true, false>(e, g, st);
            return relu ? launch_gemm2<ACT_RELU, false, false>(e, g, st) : launch_gemm2<ACT_NONE, false, false>(e, g, st);
        }
        ConvArgs g{};
        g.x = static_cast<const float *>(x_); g.n = 1; g.H = 1; g.W = rows; g.Ho = 1; g.Wo = rows; g.cin = cin_;
        g.cout16 = round_up(cout_, kProjNT) / 16; g.cout_valid = cout_; g.out_stride = cout_;
        g.wfrag = w_.as<float>(); g.bias = b_.as<float>(); g.y = static_cast<float *>(y_);
        g.range_max = rset(kRangeOther);
        if (e->b3_weights.count(w_.p)) return relu ? gemm128_relu_b3(g, st) : gemm128_b3(g, st);
        return relu ? gemm128_relu_k(g, st) : gemm128_k(g, st);
    };
    ln(s.feat.as<float>(), nullptr, e->sa_nw, e->sa_nb, e->pe.as<float>(), s.sa_x.as<float>(), g2 ? s.sa_xp2.p : nullptr);
    const float *xin = s.sa_x.as<float>();             // fp32 input of the layer (residual); with g2 its P2 form is in sa_xp2
    for (int l = 0; l < c.sa_layers; ++l) {
        pocr_engine::SaLayer &L = e->sa[l];
        if (s.sa_y[l].reserve(xe)) return 1;
        // att2: q | k | v leave the projection in P2 and attention runs on f16x2 MFMAs (encoder.hpp); the other arithmetics keep
        // the fp32-MFMA kernel on fp32 q | k | v
        const bool att2 = g2 && conv_split() == 2 && (D == 32 || D == 64 || D == 128);
        if (gemm(g2 ? s.sa_xp2.p : (const void *)xin, E, L.w_in, L.b_in, 3 * E, s.sa_qkv.p, false, g2, att2)) return 1;
        const dim3 agrid((T + 15) / 16, heads, n);
        const float scale = 1.0f / sqrtf((float)D);
        if (att2) {
            const float scale2 = scale * 1.44269504088896340736f;
#define POCR_ATT2(DD, QQ) hipLaunchKernelGGL((attention_f16x2_kernel<DD, QQ>), dim3((T + 64 * QQ - 1) / (64 * QQ), heads, n), dim3(256), 0, st, s.sa_qkv.p, s.sa_att.p, T, E, scale2, s.g_line_T, s.g_row_off, rset(kRangeOther))
            if (D == 32) POCR_ATT2(32, 2); else if (D == 64) POCR_ATT2(64, 2); else POCR_ATT2(128, 1);
#undef POCR_ATT2
        } else {
#define POCR_ATT(DD)                                                                                                                         \
        do {                                                                                                                                 \
            if (g2) hipLaunchKernelGGL((attention_kernel<DD, true>), agrid, dim3(64), 0, st, s.sa_qkv.as<float>(), s.sa_att.as<float>(), T, E, scale, s.g_line_T, s.g_row_off, rset(kRangeOther)); \
            else hipLaunchKernelGGL((attention_kernel<DD, false>), agrid, dim3(64), 0, st, s.sa_qkv.as<float>(), s.sa_att.as<float>(), T, E, scale, s.g_line_T, s.g_row_off, rset(kRangeOther)); \
        } while (0)
        if (D == 32) POCR_ATT(32); else if (D == 64) POCR_ATT(64); else POCR_ATT(128);
#undef POCR_ATT
        }
        if (gemm(s.sa_att.p, E, L.w_out, L.b_out, E, s.sa_tmp.p, false, g2, false)) return 1;
        ln(xin, s.sa_tmp.as<float>(), L.n1w, L.n1b, nullptr, s.sa_x1.as<float>(), g2 ? s.sa_x1p2.p : nullptr);
        if (gemm(g2 ? s.sa_x1p2.p : s.sa_x1.p, E, L.w1, L.b1, FF, s.sa_ff.p, true, g2, g2)) return 1;
        if (gemm(s.sa_ff.p, FF, L.w2, L.b2, E, s.sa_tmp.p, false, g2, false)) return 1;
        ln(s.sa_x1.as<float>(), s.sa_tmp.as<float>(), L.n2w, L.n2b, nullptr, s.sa_y[l].as<float>(), g2 ? s.sa_xp2.p : nullptr);
        HIP_TRY(hipGetLastError());
        xin = s.sa_y[l].as<float>();                    // the next layer's residual input (per-layer outputs stay for the test taps)
    }
    layer_in = s.sa_y[c.sa_layers - 1].as<float>();
    din = E;
    if (g2 && head2_ok(E)) layer_in_p2 = s.sa_xp2.p;      // the last LayerNorm's P2 copy
    if (c.arch == POCR_ARCH_S2S) {
        // keys / values of the encoder output for every decoder layer, computed once per launch
        // (CustomMultiheadAttention.cached_forward, transformer.py:237-247): [rows][2E] = memory W[E:3E]^T + b[E:3E]
        for (int l = 0; l < c.dec_layers; ++l) {
            if (s.s2s_kv[l].reserve(2 * xe)) return 1;
            const bool kv2 = g2 && can_gemm2(e, E, round_up(2 * E, kProjNT) / 16, E / 32, 2 * E, false) && e->b3_weights.count(e->dec[l].wc_kv.p) != 0;
            if (gemm(kv2 ? s.sa_xp2.p : (const void *)layer_in, E, e->dec[l].wc_kv, e->dec[l].bc_kv, 2 * E, s.s2s_kv[l].p, false, kv2, false)) return 1;
        }
        mark(POCR_STAGE_HEAD); mark(POCR_STAGE_CTC); mark(POCR_NUM_STAGES);
        if (guard) HIP_TRY(hipMemcpyAsync(s.range_host, s.range.p, kRangeWords * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        return 0;
    }
    } else {
    // ---- BiLSTM stack
    const int Hh = c.lstm_hidden, npad = round_up(n, 16);
    {   // (re)allocation of any buffer whose address is baked into the cached graphs flushes them
        const void *before[4] = {s.xproj.p, s.hbuf.p, s.cbuf.p, nullptr};
        const size_t stride_before = s.h_stride;
        if (s.xproj.reserve((size_t)rows * 8 * Hh * sizeof(float))) return 1;
        if (2 * (size_t)2 * npad * Hh > 2 * s.h_stride || !s.hbuf.p) {
            const size_t cap_pad = (size_t)round_up(npad, 64);
            if (s.hbuf.reserve((size_t)2 * 2 * cap_pad * Hh * sizeof(float))) return 1;
            if (s.cbuf.reserve((size_t)2 * cap_pad * Hh * sizeof(float))) return 1;
            s.h_stride = (size_t)2 * cap_pad * Hh;
        }
        // (h_stride is baked into the captured launches too - as the ping-pong offset and the memset size - and can change
        // while the over-allocated buffer stays where it is)
        bool moved = xproj_moved || before[0] != s.xproj.p || before[1] != s.hbuf.p || before[2] != s.cbuf.p || stride_before != s.h_stride;
        for (int l = 0; l < c.lstm_layers; ++l) {
            const void *yb = s.lstm_y[l].p;
            if (s.lstm_y[l].reserve((size_t)rows * 2 * Hh * sizeof(float))) return 1;
            moved = moved || yb != s.lstm_y[l].p;
        }
        if (s.seqgeom.p != s.graph_geom) { moved = true; s.graph_geom = s.seqgeom.p; }    // per-line tables moved
        if (moved) {
            for (auto &kv : s.lstm_graphs) (void)hipGraphExecDestroy(kv.second);
            s.lstm_graphs.clear();
        }
        if (!s.lstm_dims.p) {
            if (s.lstm_dims.reserve(16)) return 1;
            HIP_TRY(locked_host_malloc(reinterpret_cast<void **>(&s.lstm_dims_host), 16, hipHostMallocDefault));
        }
    }
    auto launch_step = [&](int l, int step, int slices, const int32_t *dims) {
        LstmStepArgs la{};
        la.xproj = s.xproj.as<float>(); la.whh_frag = e->whh[l].as<float>();
        la.whh2 = e->whh2.empty() ? nullptr : e->whh2[l].p;
        la.h_in = s.hbuf.as<float>() + (size_t)(step & 1) * s.h_stride;
        la.h_out = s.hbuf.as<float>() + (size_t)((step + 1) & 1) * s.h_stride;
        la.c = s.cbuf.as<float>(); la.y = s.lstm_y[l].as<float>(); la.dims = dims;
        la.line_T = s.g_line_T; la.row_off = s.g_row_off; la.slice_T = s.g_slice_T;
        la.n = n; la.npad = npad; la.T = T; la.H = Hh; la.step = step;
        la.y_p2 = l < s.lstm_p2_layers;
        const dim3 grid(Hh / 16, slices, 2);
        switch (Hh) {
            case 64: hipLaunchKernelGGL(lstm_step_kernel<1>, grid, dim3(256), 0, st, la); break;
            case 128: hipLaunchKernelGGL(lstm_step_kernel<2>, grid, dim3(256), 0, st, la); break;
            case 256: hipLaunchKernelGGL(lstm_step_kernel<4>, grid, dim3(256), 0, st, la); break;
            case 512: hipLaunchKernelGGL(lstm_step_kernel<8>, grid, dim3(256), 0, st, la); break;
            default: hipLaunchKernelGGL(lstm_step_kernel<0>, grid, dim3(256), 0, st, la); break;
        }
    };
    int bucket = 1;                          // slices rounded up to a power of two: few distinct graphs
    while (bucket * 16 < npad) bucket *= 2;
    s.lstm_dims_host[0] = n; s.lstm_dims_host[1] = npad;
    HIP_TRY(hipMemcpyAsync(s.lstm_dims.p, s.lstm_dims_host, 2 * sizeof(int32_t), hipMemcpyHostToDevice, st));
    // resident recurrence: hidden sizes whose W_hh fragments fit a workgroup's registers, launches whose clusters the chip can
    // hold (1024 workgroups = 512 lines at H = 256); otherwise one launch per step as before
    const int n_clusters = 2 * (npad / 16);
    // slices per workgroup: 1 for launches of a few slices (pages of long lines: the chain's latency is what counts),
    // 2 / 4 for many slices (the chain hides behind the next launch's convolutions: fewer resident workgroups cost
    // those less).  Measured (profiles/r03_lstm_resident.txt): c5 SL 1, c3 SL 2, c2 SL 4.
    const int n_sl = npad / 16, ug_n = Hh / 16;
    int SLn = n_sl <= 4 ? 1 : (n_sl >= 16 && T <= 160) ? 4 : 2;
    // The clusters of a launch wait for one another's members inside ONE ordinary launch: every workgroup of the grid must be
    // able to be resident at the same time, or a cluster whose tail was not dispatched spins until its timeout.  The capacity
    // is what the runtime reports for this kernel (workgroups per CU x CUs, one taken off per CU as the margin the guide asks
    // for where the query is known to be optimistic); a launch that does not fit takes more slices per workgroup, then the
    // step kernels.
    auto resident_grid = [&](int sl) { return ((2 * ((n_sl + sl - 1) / sl) + 7) / 8) * ug_n * 8; };
    // capacity of the instantiation that would be launched (they differ in registers and LDS)
    auto capacity = [&](int sl) {
        const int ki = Hh == 64 ? 0 : Hh == 128 ? 1 : 2, si = sl == 1 ? 0 : sl == 2 ? 1 : 2;
        int &cap = e->lstm_capacity[ki][si];
        if (cap < 0) {
            int per_cu = 0;
            hipError_t qe = hipErrorUnknown;
#define POCR_OCC(KPW_, SL_) qe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_resident_kernel<KPW_, SL_>, 256, 0)
            switch (ki * 3 + si) {
                case 0: POCR_OCC(1, 1); break; case 1: POCR_OCC(1, 2); break; case 2: POCR_OCC(1, 4); break;
                case 3: POCR_OCC(2, 1); break; case 4: POCR_OCC(2, 2); break; case 5: POCR_OCC(2, 4); break;
                case 6: POCR_OCC(4, 1); break; case 7: POCR_OCC(4, 2); break; default: POCR_OCC(4, 4); break;
            }
#undef POCR_OCC
            if (qe != hipSuccess || per_cu < 1) per_cu = 1;
            cap = std::min(per_cu, 4) * e->n_cus;
            if (per_cu > 1) cap -= e->n_cus;
        }
        // (Resident launches of OTHER slots still running are not subtracted: their workgroups retire within milliseconds and a
        // later launch's clusters simply become resident then - every wait is bounded far above that.  Subtracting them was
        // measured: the third launch of a page of long lines fell back to the step kernels, one page at a time 13.7 -> 19.1 ms of OCR.)
        return cap;
    };
    while (SLn < 4 && resident_grid(SLn) > capacity(SLn)) SLn *= 2;
    const bool resident_shape = e->lstm_resident && (Hh == 64 || Hh == 128 || Hh == 256) && c.lstm_layers <= 8;
    bool paused = false;                                                               // back-off after a timeout (sync_and_guard)
    if (resident_shape && !s.lstm_force_step) {
        int left = e->lstm_skip.load();
        while (left > 0 && !e->lstm_skip.compare_exchange_weak(left, left - 1)) {}
        paused = left > 0;
    }
    const bool resident = resident_shape && !s.lstm_force_step && !paused && resident_grid(SLn) <= capacity(SLn);
    const size_t sync_words = (size_t)n_clusters * 32 + 32;     // + error / diagnostic words
    s.lstm_resident_used = resident;
    s.lstm_judged = false;
    if (resident) {
        if (s.lstm_sync.reserve(((sync_words + 3) / 4 * 4) * sizeof(uint32_t))) return 1;
        s.lstm_err_off = (size_t)n_clusters * 32;
        if (!s.lstm_err_host) {
            HIP_TRY(locked_host_malloc(reinterpret_cast<void **>(&s.lstm_err_host), 8 * 4 * sizeof(uint32_t), hipHostMallocDefault));
        }
        memset(s.lstm_err_host, 0, 8 * 4 * sizeof(uint32_t));
    }
    // Layer outputs that only feed the next layer's input projection are written in the P2 layout (by the recurrence kernels:
    // lstm_store_y) when that projection runs on the persistent GEMM; the last layer's output feeds the head (fp32 MFMA).
    const bool proj2 = e->gemm2 && (2 * Hh) % 32 == 0 && can_gemm2(e, 2 * Hh, e->proj_cout16, 2 * Hh / 32, 8 * Hh, false) &&
                       e->b3_weights.count(e->proj_w[0].p) != 0;
    // ... unless the head runs on the persistent GEMM too (head2): then every layer writes P2
    const bool head2_lstm = proj2 && head2_ok(2 * Hh);
    s.lstm_p2_layers = proj2 ? c.lstm_layers - (head2_lstm ? 0 : 1) : 0;
    bool in_p2 = s.feat_is_p2;
    for (int l = 0; l < c.lstm_layers; ++l) {
        const bool y_p2 = l < s.lstm_p2_layers;
        if (l == 0 && proj0_done) {
            // (already computed on the conv stream)
        } else if (in_p2) {
            GemmP2Args g{};
            g.a = layer_in; g.w = e->proj_w[l].p; g.bias = e->proj_b[l].as<float>(); g.y = s.xproj.p;
            g.M = rows; g.nk = din / 32; g.N16 = e->proj_cout16; g.n_valid = 8 * Hh; g.ldy = 8 * Hh; g.lda = (int64_t)din * 4;
            gemm2_shape(e, g, rows, e->proj_cout16);
            if (launch_gemm2<ACT_NONE, false, false>(e, g, st)) return 1;
        } else {
        ConvArgs a{};
        a.x = layer_in; a.n = 1; a.H = 1; a.W = rows; a.Ho = 1; a.Wo = rows; a.cin = din;
        a.cout16 = e->proj_cout16; a.cout_valid = 8 * Hh; a.out_stride = 8 * Hh;
        a.wfrag = e->proj_w[l].as<float>(); a.bias = e->proj_b[l].as<float>(); a.y = s.xproj.as<float>();
        if (e->b3_weights.count(e->proj_w[l].p) ? gemm128_b3(a, st) : gemm128_k(a, st)) return 1;
        }
        in_p2 = y_p2;
        if (resident) {
            // the serial part in ONE launch: clusters of H / 16 workgroups, one per (16-line slice, direction), hand the hidden
            // state from step to step through their XCD's L2 (lstm_resident.hpp)
            const size_t n4 = (sync_words + 3) / 4;
            hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)std::min<size_t>(256, (n4 + 255) / 256)), dim3(256), 0, st, s.lstm_sync.as<f32x4>(), n4);
            LstmResidentArgs ra{};
            ra.whh2 = e->whh2.empty() ? nullptr : e->whh2[l].p;
            ra.xproj = s.xproj.as<float>(); ra.whh_frag = e->whh[l].as<float>(); ra.hbuf = s.hbuf.as<float>(); ra.y = s.lstm_y[l].as<float>();
            ra.sync = s.lstm_sync.as<unsigned>(); ra.err = s.lstm_sync.as<unsigned>() + s.lstm_err_off;
            ra.line_T = s.g_line_T; ra.row_off = s.g_row_off; ra.slice_T = s.g_slice_T;
            ra.n = n; ra.npad = npad; ra.T = T; ra.spin_limit = e->lstm_spin_limit;
            ra.y_p2 = y_p2;
            static const int force_agent = getenv("POCR_LSTM_FORCE_AGENT") ? atoi(getenv("POCR_LSTM_FORCE_AGENT")) : 0;
            ra.force_agent = force_agent;
            const unsigned grid = (unsigned)resident_grid(SLn);
#define POCR_RES(KPW_)                                                                                                             \
            do {                                                                                                                   \
                if (SLn == 1) hipLaunchKernelGGL((lstm_resident_kernel<KPW_, 1>), dim3(grid), dim3(256), 0, st, ra);                 \
                else if (SLn == 2) hipLaunchKernelGGL((lstm_resident_kernel<KPW_, 2>), dim3(grid), dim3(256), 0, st, ra);            \
                else hipLaunchKernelGGL((lstm_resident_kernel<KPW_, 4>), dim3(grid), dim3(256), 0, st, ra);                          \
            } while (0)
            switch (Hh) {
                case 64: POCR_RES(1); break;
                case 128: POCR_RES(2); break;
                default: POCR_RES(4); break;
            }
#undef POCR_RES
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(s.lstm_err_host + 4 * l, s.lstm_sync.as<unsigned>() + s.lstm_err_off, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            layer_in = s.lstm_y[l].as<float>();
            din = 2 * Hh;
            continue;
        }
        // the serial part: 2 memsets + T dependent step launches, replayed from a captured graph
        const auto key = std::make_tuple(l, T, bucket);
        auto it = s.lstm_graphs.find(key);
        if (it == s.lstm_graphs.end() && e->use_graphs) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            UnsafeLock capture_lock;                     // (released at the end of this block, after instantiation)
            bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                // (kernel nodes, not memset nodes: the initial state must be zeroed by every replay on every HIP runtime)
                hipLaunchKernelGGL(zero_fill_kernel, dim3(256), dim3(256), 0, st, s.hbuf.as<f32x4>(), 2 * s.h_stride / 4);
                hipLaunchKernelGGL(zero_fill_kernel, dim3(256), dim3(256), 0, st, s.cbuf.as<f32x4>(), s.h_stride / 4);
                for (int step = 0; step < T; ++step) launch_step(l, step, bucket, s.lstm_dims.as<int32_t>());
                ok = hipStreamEndCapture(st, &graph) == hipSuccess && graph != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (ok) {
                if (s.lstm_graphs.size() >= 256) {       // bound the cache
                    for (auto &kv : s.lstm_graphs) (void)hipGraphExecDestroy(kv.second);
                    s.lstm_graphs.clear();
                }
                it = s.lstm_graphs.emplace(key, exec).first;
            } else {
                (void)hipGetLastError();
                e->use_graphs = false;                   // capture is unavailable: plain launches from now on
            }
        }
        if (it != s.lstm_graphs.end()) {
            HIP_TRY(hipGraphLaunch(it->second, st));
        } else {
            HIP_TRY(hipMemsetAsync(s.hbuf.p, 0, 2 * s.h_stride * sizeof(float), st));
            HIP_TRY(hipMemsetAsync(s.cbuf.p, 0, s.h_stride * sizeof(float), st));
            for (int step = 0; step < T; ++step) launch_step(l, step, npad / 16, nullptr);
        }
        HIP_TRY(hipGetLastError());
        layer_in = s.lstm_y[l].as<float>();
        din = 2 * Hh;
    }
    if (head2_lstm) layer_in_p2 = s.lstm_y[c.lstm_layers - 1].p;
    }
    // ---- head: [n*T][din] -> logits [n][T][C]
    const int C = c.num_classes;
    {
        if (s.logits.reserve((size_t)rows * C * sizeof(float))) return 1;
        ConvArgs a{};
        a.x = layer_in; a.n = 1; a.H = 1; a.W = rows; a.Ho = 1; a.Wo = rows; a.cin = din;
        a.cout16 = e->head_cout16; a.cout_valid = C; a.out_stride = C;
        a.wfrag = e->head_w.as<float>(); a.bias = e->head_b.as<float>(); a.y = s.logits.as<float>();
        mark(POCR_STAGE_HEAD);
        if (layer_in_p2) {
            GemmP2Args g{};
            g.a = layer_in_p2; g.w = e->head_w2.p; g.bias = e->head_b2.as<float>(); g.y = s.logits.p;
            g.M = rows; g.nk = din / 32; g.N16 = e->head2_cout16; g.n_valid = C; g.ldy = C; g.lda = (int64_t)din * 4;
            gemm2_shape(e, g, rows, e->head2_cout16);
            g.range_flag = rset(kRangeOther);
            if (launch_gemm2<ACT_NONE, false, false>(e, g, st)) return 1;
        } else if (gemm64_k(a, st)) return 1;
    }
    // ---- greedy CTC
    {
        if (s.best.reserve((size_t)rows * sizeof(int32_t))) return 1;
        if (s.labels.reserve((size_t)n * T * sizeof(int32_t))) return 1;
        if (s.lens.reserve((size_t)n * sizeof(int32_t))) return 1;
        mark(POCR_STAGE_CTC);
        const int frames = rows;
        if (!s.nf_flag.p) {
            if (s.nf_flag.reserve(16)) return 1;
            HIP_TRY(locked_host_malloc(reinterpret_cast<void **>(&s.nf_host), 16, hipHostMallocDefault));
        }
        HIP_TRY(hipMemsetAsync(s.nf_flag.p, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(frame_argmax_kernel, dim3((frames + 3) / 4), dim3(256), 0, st,
                           s.logits.as<float>(), s.best.as<int32_t>(), frames, C, s.nf_flag.as<int32_t>());
        HIP_TRY(hipMemcpyAsync(s.nf_host, s.nf_flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        hipLaunchKernelGGL(ctc_collapse_kernel, dim3(n), dim3(64), 0, st, s.best.as<int32_t>(),
                           s.labels.as<int32_t>(), s.lens.as<int32_t>(), T, C - 1, s.g_line_T, s.g_row_off, T);
        HIP_TRY(hipGetLastError());
        mark(POCR_NUM_STAGES);
    }
    if (guard) HIP_TRY(hipMemcpyAsync(s.range_host, s.range.p, kRangeWords * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    return 0;
}

// ---- f16x2 range guard, host side.  0: the launch stayed inside f16's range; 1: some operand reached 65504 (or was not finite);
// 2: a whole activation tensor lay below 2^-13, where the low plane of the split is subnormal (conv_igemm.hpp: range_note).
int range_verdict(const Slot &s, int *which = nullptr) {
    if (!s.range_host) return 0;
    for (int k = 0; k < kRangeSets; ++k) {
        unsigned m = 0;
        for (int j = 0; j < 8; ++j) m = std::max(m, s.range_host[8 * k + j]);
        if (which) *which = k;
        if (m >= 0x477fe000u) return 1;                   // 65504.0f and above, inf, NaN
        if (k <= 9 && m != 0 && m < 0x39000000u) return 2;      // 0 < max < 2^-13
    }
    return 0;
}

// async D2H of the chunk's results into the slot's pinned buffer (layout: labels | argmax | lens | logits)
int enqueue_outputs(pocr_engine *e, Slot &s) {
    // pinned layout: labels [n][T_max] | frame argmax [rows] (room for n*T_max) | lens [n] | logits [rows][C]
    const int n = s.n, T = s.t_max, C = e->cfg.num_classes, rows = s.rows;
    hipStream_t st = s.seq_stream;
    const size_t nt_bytes = (size_t)n * T * sizeof(int32_t);
    const size_t lg_bytes = s.want_logits ? (size_t)rows * C * sizeof(float) : 0;
    const size_t need = 2 * nt_bytes + (size_t)round_up(n, 4) * sizeof(int32_t) + lg_bytes;
    if (need > s.pinned_cap) {
        if (s.pinned) (void)locked_host_free(s.pinned);
        s.pinned = nullptr; s.pinned_cap = 0;
        HIP_TRY(locked_host_malloc(&s.pinned, need + need / 4, hipHostMallocDefault));
        s.pinned_cap = need + need / 4;
    }
    char *pin = static_cast<char *>(s.pinned);
    HIP_TRY(hipMemcpyAsync(pin, s.labels.p, nt_bytes, hipMemcpyDeviceToHost, st));
    if (s.want_argmax) HIP_TRY(hipMemcpyAsync(pin + nt_bytes, s.best.p, (size_t)rows * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(pin + 2 * nt_bytes, s.lens.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (lg_bytes) HIP_TRY(hipMemcpyAsync(pin + 2 * nt_bytes + (size_t)round_up(n, 4) * sizeof(int32_t), s.logits.p, lg_bytes, hipMemcpyDeviceToHost, st));
    if (s.want_sparse) {
        if (C > 256 * SP_COLS) return fail("sparse logits: C = %d exceeds %d classes", C, 256 * SP_COLS);
        const size_t cap = (size_t)rows * C;
        const int nblk = sp_blocks(T);                       // T: the launch's longest line
        if (s.sp_rowstat.reserve((size_t)rows * 2 * sizeof(float)) || s.sp_colcount.reserve((size_t)n * nblk * C * sizeof(int32_t)) ||
            s.sp_line_nnz.reserve((size_t)n * sizeof(int32_t)) || s.sp_line_off.reserve((size_t)(n + 1) * sizeof(int64_t)) ||
            s.sp_indptr.reserve((size_t)n * (C + 1) * sizeof(int32_t)) || s.sp_data.reserve(cap * sizeof(float)) ||
            s.sp_indices.reserve(cap * sizeof(int32_t)) || s.sp_conf.reserve((size_t)n * sizeof(float)) ||
            s.sp_bid.reserve((size_t)rows * sizeof(int32_t)) || s.sp_bp.reserve((size_t)rows * sizeof(float)))
            return 1;
        const int32_t *r0 = s.sp_has_rows ? s.sp_rows.as<int32_t>() : nullptr;
        const int32_t *r1 = s.sp_has_rows ? s.sp_rows.as<int32_t>() + n : nullptr;
        sparsify_launch(st, s.logits.as<float>(), r0, r1, s.sp_rowstat.as<float>(), s.sp_colcount.as<int32_t>(), s.sp_line_nnz.as<int32_t>(),
                        s.sp_line_off.as<int64_t>(), s.sp_indptr.as<int32_t>(), s.sp_data.as<float>(), s.sp_indices.as<int32_t>(),
                        s.sp_bid.as<int32_t>(), s.sp_bp.as<float>(), s.sp_conf.as<float>(), n, T, T, C, s.sp_thr, -80.0f, (int64_t)cap,
                        s.g_line_T, s.g_row_off);
        HIP_TRY(hipGetLastError());
        const size_t off_bytes = (size_t)(n + 1) * sizeof(int64_t), ip_bytes = (size_t)n * (C + 1) * sizeof(int32_t);
        // The number of kept entries is known only on the device, but a copy enqueued at collect time would
        // queue up behind the NEXT launch's conv kernels (measured: 31 ms per launch).  So the triplets are
        // copied back speculatively now, in stream order: 1.25x the previous launch's kept entries PER FRAME times this
        // launch's frames (launches of one call differ in size - the last one of a call, pages of different lengths - and a
        // copy sized by the previous launch's absolute count then falls short: 10-20 ms per launch measured,
        // profiles/r04_launch_timeline.txt); first launch: 1/3 density; collect tops up the rest in the rare case
        // that was not enough.
        size_t spec = cap / 3;
        if (e->sp_prev_total && e->sp_prev_rows > 0) {
            const double per_row = (double)e->sp_prev_total / (double)e->sp_prev_rows;
            spec = (size_t)(per_row * 1.25 * (double)std::max(s.rows, 1)) + 4096;
        }
        if (const char *env = getenv("POCR_SPARSE_SPEC")) spec = (size_t)std::max(1L, atol(env));     // tests: force the top-up path
        spec = std::min(spec, cap);
        const size_t conf_off = off_bytes + ip_bytes;                    // [line_off | indptr | confidence | data | indices]
        const size_t trip_base = (conf_off + (size_t)n * sizeof(float) + 15) / 16 * 16;
        const size_t need_sp = trip_base + spec * 8;
        if (need_sp > s.sp_pinned_cap) {
            if (s.sp_pinned) (void)locked_host_free(s.sp_pinned);
            s.sp_pinned = nullptr; s.sp_pinned_cap = 0;
            HIP_TRY(locked_host_malloc(&s.sp_pinned, need_sp + need_sp / 4, hipHostMallocDefault));
            s.sp_pinned_cap = need_sp + need_sp / 4;
        }
        char *sp = static_cast<char *>(s.sp_pinned);
        HIP_TRY(hipMemcpyAsync(sp, s.sp_line_off.p, off_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(sp + off_bytes, s.sp_indptr.p, ip_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(sp + conf_off, s.sp_conf.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(sp + trip_base, s.sp_data.p, spec * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(sp + trip_base + spec * sizeof(float), s.sp_indices.p, spec * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        s.sp_spec = spec;
    }
    return 0;
}

int collect_outputs(pocr_engine *e, Slot &s, float *logits_ntc, int32_t *frame_argmax_nt, int32_t *labels_nt, int32_t *label_len_n) {
    const int n = s.n, T = s.t_max, C = e->cfg.num_classes, rows = s.rows;
    HIP_TRY(hipStreamSynchronize(s.seq_stream));
    s.in_flight = false;
    if (s.lstm_resident_used && s.lstm_err_host) {
        for (int l = 0; l < e->cfg.lstm_layers && l < 8; ++l)
            if (s.lstm_err_host[4 * l])
                return fail("BiLSTM layer %d: a hand-off of the resident recurrence timed out (internal error: sync_and_guard repeats such launches)", l);
        for (int l = 0; l < e->cfg.lstm_layers && l < 8; ++l)
            if (s.lstm_err_host[4 * l + 1] && !e->warned_placement) {
                e->warned_placement = true;
                fprintf(stderr, "NOTE: %u cluster(s) of the resident recurrence were not placed on one XCD and used the (slower) agent-scope hand-off.\n", s.lstm_err_host[4 * l + 1]);
            }
    }
    if (s.nf_host && *s.nf_host && !e->warned_nonfinite) {
        e->warned_nonfinite = true;
        fprintf(stderr, "WARNING: non-finite logits (NaN / inf) in a launch of %d lines - decoded like torch.argmax would (NaN is maximal).%s\n", n,
                "");
    }
    const size_t nt_bytes = (size_t)n * T * sizeof(int32_t);
    const char *pin = static_cast<const char *>(s.pinned);
    if (logits_ntc && !s.want_logits) return fail("logits were not requested at launch");
    if (frame_argmax_nt && !s.want_argmax) return fail("frame argmax was not requested at launch");
    if (labels_nt) memcpy(labels_nt, pin, nt_bytes);
    if (frame_argmax_nt) memcpy(frame_argmax_nt, pin + nt_bytes, (size_t)rows * sizeof(int32_t));
    if (label_len_n) memcpy(label_len_n, pin + 2 * nt_bytes, (size_t)n * sizeof(int32_t));
    if (logits_ntc) parallel_memcpy(logits_ntc, pin + 2 * nt_bytes + (size_t)round_up(n, 4) * sizeof(int32_t), (size_t)rows * C * sizeof(float));
    if (e->profiling) {
        for (int i = 0; i < POCR_NUM_STAGES; ++i) s.stage_ms[i] = 0.f;
        const int order[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, POCR_STAGE_AGG, POCR_STAGE_LSTM, POCR_STAGE_HEAD, POCR_STAGE_CTC, POCR_NUM_STAGES};
        for (int k = 0; k + 1 < (int)(sizeof(order) / sizeof(int)); ++k)
            HIP_TRY(hipEventElapsedTime(&s.stage_ms[order[k]], s.ev[order[k]], order[k] == POCR_STAGE_AGG ? s.ev_conv_end : s.ev[order[k + 1]]));
        HIP_TRY(hipEventElapsedTime(&s.stage_ms[POCR_STAGE_TOTAL], s.ev[0], s.ev[POCR_NUM_STAGES]));
        s.have_ms = true;
    }
    return 0;
}

template <bool RELU>
int launch_skinny(SkinnyArgs a, hipStream_t st) {
    const int c16 = a.cout16;
    auto wgs = [&](int rm, int cn) { return ((a.M + 16 * rm - 1) / (16 * rm)) * (c16 / cn); };
    // the largest tile that still gives every CU a workgroup; small problems take the smallest tile
    if (a.K >= 1024 && c16 % 2 == 0 && wgs(1, 2) >= 128) {     // long K, few columns (the second feed-forward GEMM): eight waves split K
        hipLaunchKernelGGL((skinny_gemm_kernel<1, 2, RELU, 0, 8>), dim3(c16 / 2, (a.M + 15) / 16), dim3(512), 0, st, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (c16 % 4 == 0 && wgs(2, 4) >= 256) hipLaunchKernelGGL((skinny_gemm_kernel<2, 4, RELU>), dim3(c16 / 4, (a.M + 31) / 32), dim3(256), 0, st, a);
    else if (c16 % 2 == 0 && wgs(2, 2) >= 256) hipLaunchKernelGGL((skinny_gemm_kernel<2, 2, RELU>), dim3(c16 / 2, (a.M + 31) / 32), dim3(256), 0, st, a);
    else if (c16 % 2 == 0 && wgs(1, 2) >= 256) hipLaunchKernelGGL((skinny_gemm_kernel<1, 2, RELU>), dim3(c16 / 2, (a.M + 15) / 16), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((skinny_gemm_kernel<1, 1, RELU>), dim3(c16, (a.M + 15) / 16), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}


// skinny GEMM whose input is LayerNorm(ln_a + ln_b): E = 256 or 512 have fused instances
template <bool RELU, int LNE>
int launch_skinny_ln(SkinnyArgs a, hipStream_t st) {
    const int c16 = a.cout16;
    const int rows = (a.M + 15) / 16;
    if (c16 % 4 == 0 && rows * (c16 / 4) >= 256) hipLaunchKernelGGL((skinny_gemm_kernel<1, 4, RELU, LNE>), dim3(c16 / 4, rows), dim3(256), 0, st, a);
    else if (c16 % 2 == 0 && rows * (c16 / 2) >= 256) hipLaunchKernelGGL((skinny_gemm_kernel<1, 2, RELU, LNE>), dim3(c16 / 2, rows), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((skinny_gemm_kernel<1, 1, RELU, LNE>), dim3(c16, rows), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int check_slot(pocr_engine *e, int32_t slot) {
    if (!e) return fail("engine is NULL");
    if (slot < 0 || slot >= POCR_NUM_SLOTS) return fail("slot %d out of range (0..%d)", slot, POCR_NUM_SLOTS - 1);
    return 0;
}

}  // namespace

extern "C" {

const char *pocr_last_error(void) { return g_err.c_str(); }
int pocr_abi_version(void) { return POCR_ABI_VERSION; }
int pocr_conv_split(void) { return conv_split(); }

int pocr_set_embed_id(pocr_engine *e, int32_t embed_id) {
    if (!e) return fail("engine is NULL");
    const int E = e->cfg.conv_out, num = e->cfg.embed_num;
    if (num <= 0) return fail("this model has no embeddings layer (embed_num 0) but an embed_id was given");
    if (embed_id < 0 || embed_id > num) return fail("embed_id %d outside the embeddings table (0..%d)", embed_id, num);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(locked_device_sync());                       // launches in flight read the previous row
    std::vector<float> ss(2 * (size_t)E);
    const float *row = e->embed_table.data() + (size_t)embed_id * 2 * E;
    for (int k = 0; k < E; ++k) { ss[k] = 1.0f + row[k]; ss[E + k] = row[E + k]; }      // torch: (1.0 + emb[:, :E]) in float32
    if (upload(e->embed_ss, ss, e->stream)) return 1;
    e->embed_id = embed_id;
    return 0;
}

int pocr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t pocr_num_weight_floats(const pocr_config *c) {
    if (check_cfg(c)) return 0;
    size_t t = 0;
    for (const ConvLayer &L : kConvPlan) t += (size_t)L.cout * L.cin * 9 + L.cout;
    t += 4 * 512;
    t += (size_t)c->conv_out * 512 * (c->height / 8) + c->conv_out;
    if (c->arch == POCR_ARCH_SA || c->arch == POCR_ARCH_S2S) {
        const size_t E = c->conv_out, FF = c->sa_ff;
        t += 2 * E;
        t += (size_t)c->sa_layers * (3 * E * E + 3 * E + E * E + E + FF * E + FF + E * FF + E + 4 * E);
        if (c->arch == POCR_ARCH_S2S) {
            t += (size_t)c->dec_layers * (2 * (3 * E * E + 3 * E + E * E + E) + FF * E + FF + E * FF + E + 6 * E);
            t += (size_t)c->num_classes * E;                           // embedding
        }
        t += (size_t)c->num_classes * E + c->num_classes;             // CTC head / decoder output projection
        if (c->embed_num > 0) t += (size_t)(c->embed_num + 1) * 2 * E;
        return t;
    }
    const size_t Hh = c->lstm_hidden;
    for (int l = 0; l < c->lstm_layers; ++l) {
        const size_t din = l == 0 ? c->conv_out : 2 * Hh;
        t += 2 * (4 * Hh * din + 4 * Hh * Hh + 8 * Hh);
    }
    t += (size_t)c->num_classes * 2 * Hh + c->num_classes;
    if (c->embed_num > 0) t += (size_t)(c->embed_num + 1) * 2 * c->conv_out;
    return t;
}

static int stage_ragged_impl(pocr_engine *e, int32_t slot, const uint8_t *crops, const int64_t *crop_offsets,
                             const int32_t *widths, const int32_t *w_pads, int32_t n, int32_t pad_left,
                             const int32_t *pad_lefts, const uint8_t *dev_base = nullptr);

// The constant padding column of every conv layer (conv_igemm.hpp, FillSeg): one all-zero line of 256 columns goes
// through the network once; column W/2 of every layer's output is far enough from both row ends (26 input pixels
// of receptive field) to be the value every interior padding column takes.
static int compute_pad_constants(pocr_engine *e) {
    const uint8_t dummy = 0;
    const int64_t off = 0;
    const int32_t width = 0, w_pad = 256;
    e->pad_skip = false;                         // this one launch convolves every column
    const int rc_stage = stage_ragged_impl(e, POCR_NUM_SLOTS, &dummy, &off, &width, &w_pad, 1, 0, nullptr);
    e->pad_skip = true;
    if (rc_stage) return 1;
    Slot &s = e->slot[POCR_NUM_SLOTS];
    s.want_logits = s.want_argmax = s.want_sparse = false;
    s.s2s_batches = 0; s.s2s_cap = 0;
    const bool fuse12 = e->fuse12;
    e->fuse12 = false;                           // ... and writes conv1's activation (its constant column serves pocr_debug_read)
    const int rc_run = run_network(e, s);
    e->fuse12 = fuse12;
    if (rc_run) return 1;
    HIP_TRY(hipStreamSynchronize(s.stream));
    HIP_TRY(hipStreamSynchronize(s.seq_stream));
    int wl[3] = {w_pad, w_pad / 2, w_pad / 4};
    for (int l = 0; l < 9; ++l) {
        const int W = wl[kConvLvlOut[l]], C = kConvPlan[l].cout, Hl = s.act_h[l];
        if (e->cconst[l].reserve((size_t)Hl * C * sizeof(float))) return 1;
        // (on the slot's stream and waited for below: a device-to-device hipMemcpy2D is ordered in the NULL stream and may return
        //  before it has run - the slots' streams are non-blocking, so the first launch's pad_fill_kernel could read the constants
        //  before they were there when other engines of the process kept the GPU busy: a wrong line in the FIRST call of a fresh
        //  engine, once in ~100 engines under tools/stress_first_job.py)
        HIP_TRY(hipMemcpy2DAsync(e->cconst[l].p, (size_t)C * sizeof(float), s.act[l].as<float>() + (size_t)(W / 2) * C,
                                 (size_t)W * C * sizeof(float), (size_t)C * sizeof(float), (size_t)Hl, hipMemcpyDeviceToDevice, s.stream));
    }
    HIP_TRY(hipStreamSynchronize(s.stream));
    s.staged = false;
    s.conv_done_valid = false; s.conv_part_valid = false;
    e->cconst_ready = true;
    return 0;
}

static void register_builder(pocr_engine *e);

int pocr_create(const pocr_config *cfg, const float *weights, size_t n_floats, int device_id, pocr_engine **out) {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (check_cfg(cfg)) return 1;
    if (!weights) return fail("weights is NULL");
    if (n_floats != pocr_num_weight_floats(cfg))
        return fail("weight blob has %zu floats, config needs %zu", n_floats, pocr_num_weight_floats(cfg));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail("no HIP device available: this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail("device_id %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);

    pocr_engine *e = new pocr_engine();
    e->cfg = *cfg;
    if (const char *env = getenv("POCR_NO_GRAPHS")) e->use_graphs = atoi(env) == 0;
    e->bf16x3 = conv_split() != 0;
    if (const char *env = getenv("POCR_LSTM_RESIDENT")) e->lstm_resident = atoi(env) != 0;
    if (const char *env = getenv("POCR_LSTM_SPIN_LIMIT")) e->lstm_spin_limit = std::max(1, atoi(env));
    e->p2 = conv_split() == 2 && !(getenv("POCR_NO_P2") && atoi(getenv("POCR_NO_P2")) != 0);
    e->fuse12 = e->p2 && !(getenv("POCR_NO_FUSE12") && atoi(getenv("POCR_NO_FUSE12")) != 0);
    // The fused conv1+2 kernel as 8 x 16 pixel tiles with THREE workgroups per CU (52 KB of LDS each) is 14 % faster alone than the 10 x 16 /
    // two-per-CU tiles, but it leaves no LDS for a resident recurrence of the launch ahead on the same CU: networks WITHOUT a recurrence (the
    // self-attention encoder) take it (c4 18.15 -> 18.49 k lines/s, three alternating pairs), the BiLSTM network keeps the larger tile (c3 -2.2 %
    // with the small one; profiles/r04_backbone_overlap.txt).
    e->conv2_tile8 = e->fuse12 && cfg->arch == POCR_ARCH_SA;
    e->gemm2 = e->p2 && !(getenv("POCR_NO_GEMM2") && atoi(getenv("POCR_NO_GEMM2")) != 0);
    e->head_fp32 = getenv("POCR_HEAD_FP32") && atoi(getenv("POCR_HEAD_FP32")) != 0;
    e->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e->device = device_id;
    auto bail = [&](int rc) { pocr_destroy(e); return rc; };
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail("hipStreamCreate failed"));
    for (Slot &sl : e->slot) {
        // (A CU-masked partition - backbone on 256-k CUs, sequence kernels on k - was measured twice and is not faster: round 1, step
        //  kernels: 6.9k / 4.7k / 8.4k lines/s for k = 8 / 16 / 32 vs 8.4k unmasked; round 6, the resident recurrence and the persistent
        //  GEMMs on hipExtStreamCreateWithCUMask streams of 64 / 96 / 128 CUs: 8.83 / 8.83 / 8.77 ms per c2 step against 8.69-8.70
        //  unmasked on the same box - conv2's stage time falls (3.5 -> 2.8 ms with 64 CUs) and the recurrence's rises by as much:
        //  profiles/r06_seq_cu_mask.txt.)
        if (hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking) != hipSuccess) return bail(fail("hipStreamCreate failed"));
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // numerically lower = higher priority
        if (hipStreamCreateWithPriority(&sl.seq_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) return bail(fail("hipStreamCreate failed"));
        if (hipEventCreateWithFlags(&sl.conv_part, hipEventDisableTiming) != hipSuccess) return bail(fail("hipEventCreate failed"));
        for (auto &ev : sl.ev)
            if (hipEventCreate(&ev) != hipSuccess) return bail(fail("hipEventCreate failed"));
        if (hipEventCreateWithFlags(&sl.conv_done, hipEventDisableTiming) != hipSuccess) return bail(fail("hipEventCreate failed"));
        if (hipEventCreate(&sl.ev_conv_end) != hipSuccess) return bail(fail("hipEventCreate failed"));
        sl.lstm_y.resize(cfg->arch == POCR_ARCH_BLSTM ? cfg->lstm_layers : 0);
        sl.sa_y.resize(cfg->arch == POCR_ARCH_SA || cfg->arch == POCR_ARCH_S2S ? cfg->sa_layers : 0);
        if (cfg->arch == POCR_ARCH_S2S) {
            sl.s2s_kv.resize(cfg->dec_layers);
            sl.s2s_cache.resize(cfg->dec_layers);
            for (auto &ev : sl.s2s_ev)
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return bail(fail("hipEventCreate failed"));
        }
    }
    hipStream_t st = e->stream;

    WeightCursor cur{weights};
    // conv1..9
    for (int i = 0; i < 9; ++i) {
        const ConvLayer &L = kConvPlan[i];
        const float *w = cur.take((size_t)L.cout * L.cin * 9);
        const float *b = cur.take(L.cout);
        const bool b3 = e->bf16x3 && i > 0;
        const int cout16 = round_up(L.cout, b3 ? kConvNT3[i] : kConvNT[i]) / 16;
        e->conv_cout16[i] = cout16;
        if (b3) {
            auto wsp = build_wsplit(9, L.cin, cout16, [&](int co, int ci, int tap) { return w[((size_t)co * L.cin + ci) * 9 + tap]; }, L.cout);
            std::vector<float> bias(cout16 * 16, 0.f);
            for (int k = 0; k < L.cout; ++k) bias[k] = b[k];
            if (e->conv_w[i].reserve(wsp.size() * 2)) return bail(1);
            if (locked_memcpy(e->conv_w[i].p, wsp.data(), wsp.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return bail(fail("weight upload failed"));
            if (upload(e->conv_b[i], bias, st)) return bail(1);
            continue;
        }
        std::vector<float> frag;
        if (i == 0 && conv_split() == 2) {      // the same weights as f16x2 fragments: [cout/16][plane][lane] x 8 f16, k = 8 (lane >> 4) + j
            std::vector<uint16_t> w2((size_t)4 * 2 * 64 * 8, 0);
            for (int sgrp = 0; sgrp < 4; ++sgrp)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 8 * (lane >> 4) + j, co = 16 * sgrp + (lane & 15);
                        if (k > 27 || co >= L.cout) continue;
                        uint16_t pl[2];
                        // k = 27: the bias, multiplied by the constant 1 the kernels put into that slot (conv_bf16x3.hpp: conv1_koff);
                        // k < 27: w * 256 / 255 - the kernels feed pixel BYTE / 256 (exact in f16: no low plane of the input, two
                        // MFMAs instead of three, no table look-up), so what is missing from the reference's float32(x) / 255.0
                        // (pytorch_ocr_engine.py:61) moves into the weight: w * fl(k / 255) against fl(w / 255) * k, 2^-23 relative apart
                        split_weight(k == 27 ? b[co] : w[((size_t)co * 3 + k % 3) * 9 + k / 3] / 255.0f * 256.0f, 2, pl);
                        w2[(((size_t)sgrp * 2 + 0) * 64 + lane) * 8 + j] = pl[0];
                        w2[(((size_t)sgrp * 2 + 1) * 64 + lane) * 8 + j] = pl[1];
                    }
            if (e->conv1_w2.reserve(w2.size() * 2)) return bail(1);
            if (locked_memcpy(e->conv1_w2.p, w2.data(), w2.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return bail(fail("weight upload failed"));
        }
        if (i == 0) {   // im2col form: one tap, "cin" k = (ky*3+kx)*3 + c, padded 27 -> 32
            frag = build_wfrag(1, 32, cout16, [&](int co, int k, int) { const int tap = k / 3, c = k % 3; return w[((size_t)co * 3 + c) * 9 + tap]; }, 27, L.cout);
        } else {
            frag = build_wfrag(9, L.cin, cout16, [&](int co, int ci, int tap) { return w[((size_t)co * L.cin + ci) * 9 + tap]; }, L.cin, L.cout);
        }
        std::vector<float> bias(cout16 * 16, 0.f);
        for (int k = 0; k < L.cout; ++k) bias[k] = b[k];
        if (upload(e->conv_w[i], frag, st) || upload(e->conv_b[i], bias, st)) return bail(1);
    }
    {   // BatchNorm2d(512) eval: y = (x - mean) / sqrt(var + eps) * gamma + beta = x * scale + shift
        const float *gamma = cur.take(512), *beta = cur.take(512), *mean = cur.take(512), *var = cur.take(512);
        std::vector<float> sc(512), sh(512);
        for (int k = 0; k < 512; ++k) {
            const float inv = 1.0f / std::sqrt(var[k] + kBnEps);
            sc[k] = gamma[k] * inv;
            sh[k] = beta[k] - mean[k] * sc[k];
        }
        if (upload(e->bn_scale, sc, st) || upload(e->bn_shift, sh, st)) return bail(1);
    }
    {   // aggregation conv (AH x 1)
        const int AH = cfg->height / 8, E = cfg->conv_out;
        const float *w = cur.take((size_t)E * 512 * AH);
        const float *b = cur.take(E);
        e->agg_cout16 = round_up(E, kAggNT) / 16;
        std::vector<float> bias(e->agg_cout16 * 16, 0.f);
        for (int k = 0; k < E; ++k) bias[k] = b[k];
        if (e->bf16x3) {
            auto wsp = build_wsplit(AH, 512, e->agg_cout16, [&](int co, int ci, int tap) { return w[((size_t)co * 512 + ci) * AH + tap]; }, E);
            if (upload_u16(e->agg_w, wsp, st) || upload(e->agg_b, bias, st)) return bail(1);
            e->b3_weights.insert(e->agg_w.p);
        } else {
            auto frag = build_wfrag(AH, 512, e->agg_cout16, [&](int co, int ci, int tap) { return w[((size_t)co * 512 + ci) * AH + tap]; }, 512, E);
            if (upload(e->agg_w, frag, st) || upload(e->agg_b, bias, st)) return bail(1);
        }
    }
    int head_in = 2 * cfg->lstm_hidden;
    if (cfg->arch == POCR_ARCH_SA || cfg->arch == POCR_ARCH_S2S) {   // self-attention encoder weights
        const int E = cfg->conv_out, FF = cfg->sa_ff;
        head_in = E;
        auto vec = [&](size_t n_) { const float *p_ = cur.take(n_); return std::vector<float>(p_, p_ + n_); };
        // rows [r0, r0 + cout_) of a [*, cin_] matrix at w / b -> fragment order, cout padded to a multiple of nt
        auto lin_rows = [&](DevBuf &wbuf, DevBuf &bbuf, const float *w, const float *b, int r0, int cout_, int cin_, int nt) {
            const int c16 = round_up(cout_, nt) / 16;
            std::vector<float> bias(c16 * 16, 0.f);
            for (int k = 0; k < cout_; ++k) bias[k] = b[r0 + k];
            if (e->bf16x3 && nt == kProjNT && cin_ % 32 == 0) {          // runs on gemm128_b3
                auto wsp = build_wsplit(1, cin_, c16, [&](int co, int ci, int) { return w[(size_t)(r0 + co) * cin_ + ci]; }, cout_);
                if (upload_u16(wbuf, wsp, st) || upload(bbuf, bias, st)) return true;
                e->b3_weights.insert(wbuf.p);
                return false;
            }
            auto frag = build_wfrag(1, cin_, c16, [&](int co, int ci, int) { return w[(size_t)(r0 + co) * cin_ + ci]; }, cin_, cout_);
            return upload(wbuf, frag, st) || upload(bbuf, bias, st);
        };
        auto lin = [&](DevBuf &wbuf, DevBuf &bbuf, int cout_, int cin_) {
            const float *w = cur.take((size_t)cout_ * cin_);
            const float *b = cur.take(cout_);
            return lin_rows(wbuf, bbuf, w, b, 0, cout_, cin_, kProjNT);
        };
        if (upload(e->sa_nw, vec(E), st) || upload(e->sa_nb, vec(E), st)) return bail(1);
        e->sa.resize(cfg->sa_layers);
        for (int l = 0; l < cfg->sa_layers; ++l) {
            pocr_engine::SaLayer &L = e->sa[l];
            if (lin(L.w_in, L.b_in, 3 * E, E) || lin(L.w_out, L.b_out, E, E) || lin(L.w1, L.b1, FF, E) || lin(L.w2, L.b2, E, FF)) return bail(1);
            if (upload(L.n1w, vec(E), st) || upload(L.n1b, vec(E), st) || upload(L.n2w, vec(E), st) || upload(L.n2b, vec(E), st)) return bail(1);
        }
        if (cfg->arch == POCR_ARCH_S2S) {   // decoder: skinny-GEMM weights are padded to 64 columns (kSkinnyNT)
            e->dec.resize(cfg->dec_layers);
            for (int l = 0; l < cfg->dec_layers; ++l) {
                pocr_engine::DecLayer &L = e->dec[l];
                auto skinny = [&](DevBuf &wbuf, DevBuf &bbuf, int cout_, int cin_) {
                    const float *w = cur.take((size_t)cout_ * cin_);
                    const float *b = cur.take(cout_);
                    return lin_rows(wbuf, bbuf, w, b, 0, cout_, cin_, kSkinnyNT);
                };
                if (skinny(L.ws_in, L.bs_in, 3 * E, E) || skinny(L.ws_out, L.bs_out, E, E)) return bail(1);
                {
                    const float *w = cur.take((size_t)3 * E * E);
                    const float *b = cur.take(3 * E);
                    if (lin_rows(L.wc_q, L.bc_q, w, b, 0, E, E, kSkinnyNT)) return bail(1);
                    if (lin_rows(L.wc_kv, L.bc_kv, w, b, E, 2 * E, E, kProjNT)) return bail(1);     // runs on gemm128_k over all memory rows
                }
                if (skinny(L.wc_out, L.bc_out, E, E) || skinny(L.w1, L.b1, FF, E) || skinny(L.w2, L.b2, E, FF)) return bail(1);
                if (upload(L.n1w, vec(E), st) || upload(L.n1b, vec(E), st) || upload(L.n2w, vec(E), st) || upload(L.n2b, vec(E), st) ||
                    upload(L.n3w, vec(E), st) || upload(L.n3b, vec(E), st)) return bail(1);
            }
            if (upload(e->dec_embed, vec((size_t)cfg->num_classes * E), st)) return bail(1);
        }
    } else {   // BiLSTM layers
        const int Hh = cfg->lstm_hidden, KGT = Hh / 16;
        e->proj_cout16 = round_up(8 * Hh, kProjNT) / 16;
        e->proj_w.resize(cfg->lstm_layers); e->proj_b.resize(cfg->lstm_layers);
        e->whh.resize(cfg->lstm_layers);
        // the recurrent GEMM follows the conv arithmetic: f16x2 for the hidden sizes the unrolled kernels cover, fp32 MFMA otherwise
        const bool lstm_f16 = conv_split() == 2 && (cfg->lstm_hidden == 64 || cfg->lstm_hidden == 128 || cfg->lstm_hidden == 256 || cfg->lstm_hidden == 512);
        if (lstm_f16) e->whh2.resize(cfg->lstm_layers);
        for (int l = 0; l < cfg->lstm_layers; ++l) {
            const int din = l == 0 ? cfg->conv_out : 2 * Hh;
            const float *wih[2], *whh[2], *bih[2], *bhh[2];
            for (int d = 0; d < 2; ++d) {
                wih[d] = cur.take((size_t)4 * Hh * din); whh[d] = cur.take((size_t)4 * Hh * Hh);
                bih[d] = cur.take(4 * Hh); bhh[d] = cur.take(4 * Hh);
            }
            const bool proj_b3 = e->bf16x3 && din % 32 == 0;
            std::vector<float> frag;
            std::vector<uint16_t> proj_split;
            if (proj_b3) proj_split = build_wsplit(1, din, e->proj_cout16, [&](int co, int ci, int) { const int d = co / (4 * Hh), r = co % (4 * Hh); return wih[d][(size_t)r * din + ci]; }, 8 * Hh);
            else frag = build_wfrag(1, din, e->proj_cout16, [&](int co, int ci, int) { const int d = co / (4 * Hh), r = co % (4 * Hh); return wih[d][(size_t)r * din + ci]; }, din, 8 * Hh);
            std::vector<float> bias(e->proj_cout16 * 16, 0.f);
            for (int d = 0; d < 2; ++d)
                for (int r = 0; r < 4 * Hh; ++r) bias[d * 4 * Hh + r] = bih[d][r] + bhh[d][r];
            // whh_frag[dir][ug][kg][gate][lane][j] = W_hh[gate*H + 16*ug + (lane&15)][16*kg + 4*(lane>>4) + j]
            std::vector<float> wf((size_t)2 * KGT * KGT * 4 * 256);
            size_t o = 0;
            for (int d = 0; d < 2; ++d)
                for (int ug = 0; ug < KGT; ++ug)
                    for (int kg = 0; kg < KGT; ++kg)
                        for (int g = 0; g < 4; ++g)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 4; ++j)
                                    wf[o++] = whh[d][(size_t)(g * Hh + 16 * ug + (lane & 15)) * Hh + 16 * kg + 4 * (lane >> 4) + j];
            if ((proj_b3 ? upload_u16(e->proj_w[l], proj_split, st) : upload(e->proj_w[l], frag, st)) || upload(e->proj_b[l], bias, st) || upload(e->whh[l], wf, st)) return bail(1);
            if (lstm_f16) {
                // whh2[dir][ug][blk][gate][plane][lane][8 f16] = plane of W_hh[gate*H + 16 ug + (lane & 15)][32 blk + 8 (lane >> 4) + j]
                const int NB = Hh / 32;
                std::vector<uint16_t> w2((size_t)2 * KGT * NB * 4 * 2 * 64 * 8);
                size_t o2 = 0;
                for (int d = 0; d < 2; ++d)
                    for (int ug = 0; ug < KGT; ++ug)
                        for (int blk = 0; blk < NB; ++blk)
                            for (int g = 0; g < 4; ++g) {
                                uint16_t part[2][64][8];
                                for (int lane = 0; lane < 64; ++lane)
                                    for (int j = 0; j < 8; ++j) {
                                        uint16_t pl[3] = {0, 0, 0};
                                        split_weight(whh[d][(size_t)(g * Hh + 16 * ug + (lane & 15)) * Hh + 32 * blk + 8 * (lane >> 4) + j], 2, pl);
                                        part[0][lane][j] = pl[0]; part[1][lane][j] = pl[1];
                                    }
                                memcpy(&w2[o2], part, sizeof(part));
                                o2 += 2 * 64 * 8;
                            }
                if (upload_u16(e->whh2[l], w2, st)) return bail(1);
            }
            if (proj_b3) e->b3_weights.insert(e->proj_w[l].p);
        }
    }
    {   // head
        const int C = cfg->num_classes;
        const float *w = cur.take((size_t)C * head_in);
        const float *b = cur.take(C);
        e->head_cout16 = round_up(C, kHeadNT) / 16;
        auto frag = build_wfrag(1, head_in, e->head_cout16, [&](int co, int ci, int) { return w[(size_t)co * head_in + ci]; }, head_in, C);
        std::vector<float> bias(e->head_cout16 * 16, 0.f);
        for (int k = 0; k < C; ++k) bias[k] = b[k];
        if (upload(e->head_w, frag, st) || upload(e->head_b, bias, st)) return bail(1);
        if (e->bf16x3 && conv_split() == 2 && head_in % 32 == 0) {     // the same layer for gemm_f16x2_kernel (P2 input)
            e->head2_cout16 = round_up(C, kGemmBN) / 16;
            auto wsp = build_wsplit(1, head_in, e->head2_cout16, [&](int co, int ci, int) { return w[(size_t)co * head_in + ci]; }, C);
            std::vector<float> bias2(e->head2_cout16 * 16, 0.f);
            for (int k = 0; k < C; ++k) bias2[k] = b[k];
            if (upload_u16(e->head_w2, wsp, st) || upload(e->head_b2, bias2, st)) return bail(1);
        }
    }
    if (cfg->embed_num > 0) {   // style embeddings: kept on the host, one row goes to the device per pocr_set_embed_id
        const size_t nf = (size_t)(cfg->embed_num + 1) * 2 * cfg->conv_out;
        const float *w = cur.take(nf);
        e->embed_table.assign(w, w + nf);
    }
    {   // u8 -> f32 table, bit-exact with torch's .float() / 255.0 (true division, pytorch_ocr_engine.py:61)
        std::vector<float> lut(256);
        for (int i = 0; i < 256; ++i) lut[i] = (float)i / 255.0f;
        if (upload(e->lut, lut, st)) return bail(1);
    }
    {   // padding-column skipping (POCR_NO_PAD_SKIP=1: convolve every column, for A/B checks); the constants
        // themselves are computed by the first launch that has such columns (stage_ragged_impl)
        const char *env = getenv("POCR_NO_PAD_SKIP");
        e->pad_skip = !(env && atoi(env) != 0);
    }
    if (conv_split() == 2 && g_f16_weight_overflow.exchange(false))
        return bail(fail("a convolution / projection weight lies outside f16's range (|w| > 65504 or not finite): the default f16x2 "
                         "arithmetic cannot represent it - set POCR_CONV_SPLIT=3 (bf16x3, fp32's range)"));
    if (conv_split() == 2) {
        e->weights_host.assign(weights, weights + n_floats);      // for the fall-back engine of the range guard
        // The fall-back engine is built now, on a thread of its own, so that the first launch that needs it does not pay for it
        // (weights laid out a second time: ~0.2 s of host work, a second copy of the weights in HBM; its activation buffers are
        // still allocated by the first re-run of each slot).  POCR_FALLBACK_EAGER=0: build it when it is first needed.
        const char *env = getenv("POCR_FALLBACK_EAGER");
        if (!(env && atoi(env) == 0)) {
            e->shadow_state.store(1);
            register_builder(e);
            e->shadow_builder = std::thread([e] {
                std::lock_guard<std::mutex> lock(e->shadow_mu);
                if (e->shadow.load()) return;           // a launch needed it before this thread ran (ensure_shadow built it)
                // (an exception leaving a thread's function is std::terminate: a failed allocation here must only mean "no eager
                // fall-back engine" - it is then built on demand, where the error can be reported to the caller)
                try {
                    SplitScope scope(3);
                    pocr_engine *sh = nullptr;
                    if (hipSetDevice(e->device) == hipSuccess &&
                        pocr_create(&e->cfg, e->weights_host.data(), e->weights_host.size(), e->device, &sh) == 0) {
                        sh->is_shadow = true;
                        e->shadow.store(sh);
                        e->shadow_state.store(2);
                    } else {
                        e->shadow_state.store(-1);
                    }
                } catch (...) {
                    e->shadow_state.store(-1);
                }
            });
        }
    }
    *out = e;
    return 0;
}

static void comm_release(pocr_engine *e);

// Engines whose fall-back builder thread may still be inside HIP calls: a process that exits without pocr_destroy (an interpreter
// that never closed its engines) joins them before the runtime is torn down under them.
static std::mutex g_builders_mu;
static std::unordered_set<pocr_engine *> g_builders;
static void join_builders_at_exit() {
    std::vector<pocr_engine *> live;
    { std::lock_guard<std::mutex> lock(g_builders_mu); live.assign(g_builders.begin(), g_builders.end()); g_builders.clear(); }
    for (pocr_engine *e : live)
        if (e->shadow_builder.joinable()) e->shadow_builder.join();
}
static void register_builder(pocr_engine *e) {
    static std::once_flag once;
    std::call_once(once, [] { std::atexit(join_builders_at_exit); });
    std::lock_guard<std::mutex> lock(g_builders_mu);
    g_builders.insert(e);
}

void pocr_destroy(pocr_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    { std::lock_guard<std::mutex> lock(g_builders_mu); g_builders.erase(e); }
    if (e->shadow_builder.joinable()) e->shadow_builder.join();
    (void)locked_device_sync();
    if (pocr_engine *sh = e->shadow.exchange(nullptr)) pocr_destroy(sh);
    if (e->comm.active()) comm_release(e);
    for (auto &b : e->conv_w) b.release();
    for (auto &b : e->cconst) b.release();
    for (auto &b : e->conv_b) b.release();
    for (auto &v : {&e->proj_w, &e->proj_b, &e->whh, &e->whh2})
        for (auto &b : *v) b.release();
    for (auto &L : e->sa)
        for (DevBuf *b : {&L.w_in, &L.b_in, &L.w_out, &L.b_out, &L.w1, &L.b1, &L.w2, &L.b2, &L.n1w, &L.n1b, &L.n2w, &L.n2b}) b->release();
    for (auto &L : e->dec)
        for (DevBuf *b : {&L.ws_in, &L.bs_in, &L.ws_out, &L.bs_out, &L.wc_q, &L.bc_q, &L.wc_kv, &L.bc_kv, &L.wc_out, &L.bc_out,
                          &L.w1, &L.b1, &L.w2, &L.b2, &L.n1w, &L.n1b, &L.n2w, &L.n2b, &L.n3w, &L.n3b}) b->release();
    e->dec_embed.release();
    for (DevBuf *b : {&e->sa_nw, &e->sa_nb, &e->pe, &e->bn_scale, &e->bn_shift, &e->agg_w, &e->agg_b, &e->head_w, &e->head_b, &e->head_w2, &e->head_b2, &e->lut, &e->conv1_w2})
        b->release();
    for (Slot &s : e->slot) {
        for (auto &b : s.act) b.release();
        for (auto &v : {&s.lstm_y, &s.sa_y, &s.s2s_kv, &s.s2s_cache})
            for (auto &b : *v) b.release();
        for (DevBuf *b : {&s.s2s_x, &s.s2s_x1, &s.s2s_x2, &s.s2s_t, &s.s2s_ctx, &s.s2s_q, &s.s2s_ff, &s.s2s_logits, &s.s2s_tokens,
                          &s.s2s_state, &s.s2s_tables}) b->release();
        if (s.s2s_pinned) (void)locked_host_free(s.s2s_pinned);
        if (s.s2s_flags) (void)locked_host_free(s.s2s_flags);
        for (auto &ev : s.s2s_ev)
            if (ev) (void)hipEventDestroy(ev);
        for (DevBuf *b : {&s.crops, &s.lines, &s.feat, &s.xproj, &s.hbuf, &s.cbuf, &s.logits, &s.best, &s.labels, &s.lens,
                          &s.sa_x, &s.sa_x1, &s.sa_qkv, &s.sa_att, &s.sa_tmp, &s.sa_ff, &s.sa_xp2, &s.sa_x1p2, &s.feat_p2, &s.sp_rowstat, &s.sp_colcount,
                          &s.sp_line_nnz, &s.sp_line_off, &s.sp_indptr, &s.sp_data, &s.sp_indices, &s.sp_rows, &s.sp_conf, &s.sp_bid, &s.sp_bp, &s.geom, &s.seqgeom})
            b->release();
        if (s.pinned) (void)locked_host_free(s.pinned);
        if (s.sp_pinned) (void)locked_host_free(s.sp_pinned);
        for (auto &kv : s.lstm_graphs) (void)hipGraphExecDestroy(kv.second);
        s.lstm_graphs.clear();
        s.lstm_dims.release();
        if (s.lstm_dims_host) (void)locked_host_free(s.lstm_dims_host);
        s.nf_flag.release();
        if (s.nf_host) (void)locked_host_free(s.nf_host);
        s.range.release();
        if (s.range_host) (void)locked_host_free(s.range_host);
        s.lstm_sync.release();
        if (s.lstm_err_host) (void)locked_host_free(s.lstm_err_host);
        if (s.host_in) (void)locked_host_free(s.host_in);
        for (auto &ev : s.ev)
            if (ev) (void)hipEventDestroy(ev);
        if (s.conv_done) (void)hipEventDestroy(s.conv_done);
        if (s.conv_part) (void)hipEventDestroy(s.conv_part);
        if (s.ev_conv_end) (void)hipEventDestroy(s.ev_conv_end);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.seq_stream) (void)hipStreamDestroy(s.seq_stream);
    }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int pocr_num_slots(void) { return POCR_NUM_SLOTS; }

// Builds the per-line geometry tables of a staged set of lines (host) and uploads them in one blob.
static int build_geometry(pocr_engine *e, Slot &s, const int32_t *w_pads, int n, const int32_t *widths, int32_t pad_left,
                          const int32_t *pad_lefts) {
    const pocr_config &c = e->cfg;
    const int H = c.height, E = c.conv_out;
    struct Blob {
        std::vector<char> &buf;
        size_t add(const void *src, size_t bytes) {
            const size_t off = (buf.size() + 15) / 16 * 16;
            buf.resize(off + bytes);
            if (bytes) memcpy(buf.data() + off, src, bytes);
            return off;
        }
    } blob{s.geom_host};
    s.geom_host.clear();
    std::vector<int32_t> lvl[3];
    for (auto &v : lvl) v.resize(n);
    int t_max = 0, w_max = 0;
    for (int i = 0; i < n; ++i) {
        lvl[0][i] = w_pads[i]; lvl[1][i] = w_pads[i] / 2; lvl[2][i] = lvl[1][i] / 2;
        t_max = std::max(t_max, lvl[2][i]); w_max = std::max(w_max, w_pads[i]);
    }
    size_t off_lvl[3];
    for (int k = 0; k < 3; ++k) off_lvl[k] = blob.add(lvl[k].data(), (size_t)n * sizeof(int32_t));
    s.h_lvl2_off = off_lvl[2];
    // per conv output: element offsets (prefix sums), and pixel-tile tables
    size_t off_act[9], off_tiles[10];
    int hh = H;
    std::vector<int64_t> offs(n + 1);
    std::vector<PixelTile> tiles;
    std::vector<FillSeg> fills;
    std::vector<char> skipped;
    // constant padding columns of every line on both sides of its crop, in the coordinates of the current layer's
    // input: zlo/zhi[side][line].  Level 0: the zero columns themselves; every 3x3 conv shrinks an interval by one
    // column per side (conservative at the row ends), every width pool halves it inwards.
    std::vector<int32_t> zlo[2], zhi[2];
    for (int sd = 0; sd < 2; ++sd) { zlo[sd].assign(n, 0); zhi[sd].assign(n, 0); }
    const bool skip = e->pad_skip;
    if (skip)
        for (int i = 0; i < n; ++i) {
            const int pl = pad_lefts ? pad_lefts[i] : pad_left;
            const int c_lo = std::min(pl, w_pads[i]), c_hi = std::min(pl + widths[i], w_pads[i]);
            zlo[0][i] = 0; zhi[0][i] = c_lo;
            zlo[1][i] = std::max(c_hi, c_lo); zhi[1][i] = w_pads[i];
        }
    for (int k = 0; k < 10; ++k) {
        const int th = conv_tile_h(e->p2, e->bf16x3, k, e->conv2_tile8), tw = conv_tile_w(e->p2, e->bf16x3, k);
        const int h_in = k < 9 ? hh : hh;                       // aggregation conv: one output row
        const int rows_out = k < 9 ? h_in : 1;
        const int pw = k < 9 ? kConvPlan[k].pw : 1;
        tiles.clear();
        for (int i = 0; i < n; ++i) {
            const int w_in = lvl[kConvLvlIn[k]][i];
            const int nh = (rows_out + th - 1) / th, nw = (w_in + tw - 1) / tw;
            if (nh > 0x7fff || nw > 0xffff) return fail("line %d is too large for the tile table", i);
            // constant output columns of this layer (before pooling): the input intervals shrunk by the 3x3 kernel
            int plo[2] = {0, 0}, phi[2] = {0, 0};
            if (skip && k < 9)
                for (int sd = 0; sd < 2; ++sd) {
                    plo[sd] = std::max(zlo[sd][i] + 1, 0);
                    phi[sd] = std::min(zhi[sd][i] - 1, w_in);
                }
            int fb[2] = {-1, -1}, fe[2] = {-1, -1};             // first / last skipped w-tile per side
            skipped.assign(nw, 0);
            for (int b_ = 0; b_ < nw; ++b_) {
                const int x0 = b_ * tw, x1 = std::min((b_ + 1) * tw, w_in);
                for (int sd = 0; sd < 2; ++sd)
                    if (skip && k < 9 && x0 >= plo[sd] && x1 <= phi[sd]) {
                        if (fb[sd] < 0) fb[sd] = b_;
                        fe[sd] = b_;
                        skipped[b_] = 1;
                    }
            }
            for (int a_ = 0; a_ < nh; ++a_)
                for (int b_ = 0; b_ < nw; ++b_)
                    if (!skipped[b_]) tiles.push_back(PixelTile{i, (a_ << 16) | b_});
            if (k < 9) {
                const int w_out = lvl[kConvLvlOut[k]][i];
                for (int sd = 0; sd < 2; ++sd) {
                    if (fb[sd] >= 0) {
                        const int c0 = fb[sd] * tw / pw, c1 = std::min(std::min((fe[sd] + 1) * tw, w_in) / pw, w_out);
                        if (c1 > c0) fills.push_back(FillSeg{k, i, c0, c1});
                    }
                    // intervals of the next layer's input (this layer's pooled output)
                    zlo[sd][i] = std::min((plo[sd] + pw - 1) / pw, w_out);
                    zhi[sd][i] = std::max(std::min(phi[sd] / pw, w_out), 0);
                }
            }
        }
        s.g_ntiles[k] = (int)tiles.size();
        off_tiles[k] = blob.add(tiles.data(), tiles.size() * sizeof(PixelTile));
        if (k < 9) {
            const ConvLayer &L = kConvPlan[k];
            const int h_out = h_in / L.ph;
            int64_t acc = 0;
            for (int i = 0; i < n; ++i) { offs[i] = acc; acc += (int64_t)h_out * lvl[kConvLvlOut[k]][i] * L.cout; }
            offs[n] = acc;
            s.act_elems[k] = acc;
            off_act[k] = blob.add(offs.data(), (size_t)(n + 1) * sizeof(int64_t));
            hh = h_out;
        }
    }
    // sequence tensors: rows
    std::vector<int32_t> row_off(n + 1), slice_T((n + 15) / 16, 0), row_t, row_line;
    int rows = 0;
    for (int i = 0; i < n; ++i) {
        row_off[i] = rows;
        rows += lvl[2][i];
        slice_T[i / 16] = std::max(slice_T[i / 16], lvl[2][i]);
    }
    row_off[n] = rows;
    row_t.resize(rows > 0 ? rows : 1);
    row_line.resize(rows > 0 ? rows : 1);
    for (int i = 0; i < n; ++i)
        for (int t = 0; t < lvl[2][i]; ++t) { row_t[row_off[i] + t] = t; row_line[row_off[i] + t] = i; }
    for (int i = 0; i <= n; ++i) offs[i] = (int64_t)row_off[i] * E;
    const size_t off_feat = blob.add(offs.data(), (size_t)(n + 1) * sizeof(int64_t));
    const size_t off_row = blob.add(row_off.data(), (size_t)(n + 1) * sizeof(int32_t));
    const size_t off_slice = blob.add(slice_T.data(), slice_T.size() * sizeof(int32_t));
    const size_t off_rowt = blob.add(row_t.data(), row_t.size() * sizeof(int32_t));
    const size_t off_rowline = blob.add(row_line.data(), row_line.size() * sizeof(int32_t));
    {
        int cap = s.sg_cap > 0 ? s.sg_cap : 64;
        while (cap < n) cap *= 2;
        if (cap != s.sg_cap) {
            s.seqgeom.release();
            if (s.seqgeom.reserve(((size_t)2 * cap + 16 + cap / 16 + 16) * sizeof(int32_t))) return 1;
            s.sg_cap = cap;
        }
        s.sg_host.assign((size_t)2 * cap + 16 + cap / 16 + 16, 0);
        memcpy(s.sg_host.data(), lvl[2].data(), (size_t)n * sizeof(int32_t));
        memcpy(s.sg_host.data() + cap, row_off.data(), (size_t)(n + 1) * sizeof(int32_t));
        memcpy(s.sg_host.data() + 2 * cap + 16, slice_T.data(), slice_T.size() * sizeof(int32_t));
    }
    const size_t off_fill = blob.add(fills.data(), fills.size() * sizeof(FillSeg));
    if (s.geom.reserve(s.geom_host.size())) return 1;
    const char *d = static_cast<const char *>(s.geom.p);
    for (int k = 0; k < 3; ++k) s.g_lvl_w[k] = reinterpret_cast<const int32_t *>(d + off_lvl[k]);
    for (int k = 0; k < 9; ++k) s.g_act_off[k] = reinterpret_cast<const int64_t *>(d + off_act[k]);
    for (int k = 0; k < 10; ++k) s.g_tiles[k] = reinterpret_cast<const PixelTile *>(d + off_tiles[k]);
    s.g_fill = reinterpret_cast<const FillSeg *>(d + off_fill);
    s.g_nfill = (int)fills.size();
    s.g_feat_off = reinterpret_cast<const int64_t *>(d + off_feat);
    (void)off_row; (void)off_slice;
    s.g_line_T = s.seqgeom.as<int32_t>();
    s.g_row_off = s.seqgeom.as<int32_t>() + s.sg_cap;
    s.g_slice_T = s.seqgeom.as<int32_t>() + 2 * s.sg_cap + 16;
    s.g_row_t = reinterpret_cast<const int32_t *>(d + off_rowt);
    s.g_row_line = reinterpret_cast<const int32_t *>(d + off_rowline);
    s.rows = rows; s.t_max = t_max; s.w_pad = w_max;
    return 0;
}

// dev_base != nullptr: the crops are already in HBM (pocr_slot_stage_resident) - crop_offsets are byte offsets from
// dev_base (any sign: lines of several resident buffers), nothing is copied, conv1 reads them where they lie
static int stage_ragged_impl(pocr_engine *e, int32_t slot, const uint8_t *crops, const int64_t *crop_offsets,
                             const int32_t *widths, const int32_t *w_pads, int32_t n, int32_t pad_left,
                             const int32_t *pad_lefts, const uint8_t *dev_base) {
    if (slot != POCR_NUM_SLOTS && check_slot(e, slot)) return 1;         // POCR_NUM_SLOTS = the internal slot
    Slot &s = e->slot[slot];
    if (s.in_flight) return fail("slot %d has a launch in flight: collect it first", slot);
    s.staged = false;
    if (n <= 0) return fail("n must be positive (got %d)", n);
    if (pad_left < 0) return fail("pad_left must be >= 0");
    if ((!crops && !dev_base) || !crop_offsets || !widths || !w_pads) return fail("NULL input pointer");
    HIP_TRY(hipSetDevice(e->device));
    const int H = e->cfg.height;
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (widths[i] < 0) return fail("line %d has negative width", i);
        if (w_pads[i] < 4) return fail("line %d: w_pad must be >= 4 (got %d)", i, w_pads[i]);
        if ((pad_lefts ? pad_lefts[i] : pad_left) < 0) return fail("line %d: pad_left must be >= 0", i);
        if (dev_base) continue;
        if (crop_offsets[i] < 0) return fail("line %d has negative offset", i);
        const size_t end = (size_t)crop_offsets[i] + (size_t)H * widths[i] * 3;
        if (end > total) total = end;
    }
    if (build_geometry(e, s, w_pads, n, widths, pad_left, pad_lefts)) return 1;
    if (s.g_nfill > 0 && !e->cconst_ready && compute_pad_constants(e)) return 1;
    // pinned staging: [LineDesc table | crop pool]; the H2D copies then run asynchronously on the slot's
    // stream (they overlap the other slot's kernels) and the caller's buffers are free on return
    const size_t desc_bytes = (size_t)round_up(n, 4) * sizeof(LineDesc);
    const size_t need = desc_bytes + total;
    if (need > s.host_in_cap) {
        if (s.host_in) (void)locked_host_free(s.host_in);
        s.host_in = nullptr; s.host_in_cap = 0;
        HIP_TRY(locked_host_malloc(&s.host_in, need + need / 4, hipHostMallocDefault));
        s.host_in_cap = need + need / 4;
    }
    LineDesc *desc = static_cast<LineDesc *>(s.host_in);
    for (int i = 0; i < n; ++i) {
        desc[i].offset = crop_offsets[i]; desc[i].width = widths[i];
        desc[i].pad_left = pad_lefts ? pad_lefts[i] : pad_left;
    }
    if (total) memcpy(static_cast<char *>(s.host_in) + desc_bytes, crops, total);
    s.crops_ext = dev_base;
    if (!dev_base && s.crops.reserve(total ? total : 1)) return 1;
    if (s.lines.reserve((size_t)n * sizeof(LineDesc))) return 1;
    if (total) HIP_TRY(hipMemcpyAsync(s.crops.p, static_cast<char *>(s.host_in) + desc_bytes, total, hipMemcpyHostToDevice, s.stream));
    HIP_TRY(hipMemcpyAsync(s.lines.p, desc, (size_t)n * sizeof(LineDesc), hipMemcpyHostToDevice, s.stream));
    HIP_TRY(hipMemcpyAsync(s.geom.p, s.geom_host.data(), s.geom_host.size(), hipMemcpyHostToDevice, s.stream));
    HIP_TRY(hipMemcpyAsync(s.seqgeom.p, s.sg_host.data(), s.sg_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, s.stream));
    HIP_TRY(hipStreamSynchronize(s.stream));        // geom_host is pageable: finish the copy before it can change
    s.n = n; s.staged = true; s.have_ms = false;
    s.s2s_launched = s.s2s_decoded = false;
    s.redirect = false; s.guard_checked = true;
    if (conv_split() == 2 && !e->is_shadow && slot != POCR_NUM_SLOTS) {     // (the fall-back engine stages the same lines: run_fallback)
        s.st_off.assign(crop_offsets, crop_offsets + n); s.st_w.assign(widths, widths + n); s.st_wpad.assign(w_pads, w_pads + n);
        s.st_padl.resize(n);
        for (int i = 0; i < n; ++i) s.st_padl[i] = pad_lefts ? pad_lefts[i] : pad_left;
    }
    return 0;
}

int pocr_slot_stage_ragged(pocr_engine *e, int32_t slot, const uint8_t *crops, const int64_t *crop_offsets,
                           const int32_t *widths, const int32_t *w_pads, int32_t n, int32_t pad_left) {
    if (check_slot(e, slot)) return 1;
    return stage_ragged_impl(e, slot, crops, crop_offsets, widths, w_pads, n, pad_left, nullptr);
}

int pocr_slot_stage_lines(pocr_engine *e, int32_t slot, const uint8_t *crops, const int64_t *crop_offsets,
                          const int32_t *widths, int32_t n, int32_t w_pad, int32_t pad_left) {
    if (n <= 0) return fail("n must be positive (got %d)", n);
    if (w_pad < 4) return fail("w_pad must be >= 4 (got %d)", w_pad);
    std::vector<int32_t> wp(n, w_pad);
    return pocr_slot_stage_ragged(e, slot, crops, crop_offsets, widths, wp.data(), n, pad_left);
}

// ---- f16x2 range guard: a launch whose operands left f16's range (range_verdict) is run again on bf16x3 - the same lines, the
// same requests - on a second engine of this process (created on first use from the retained weight blob), and the slot's
// results are taken from there.  pero_ocr/ocr_engine/pytorch_ocr_engine.py:61-69 computes in plain fp32: no input may give
// worse than fp32's range here either.
static int sync_and_guard(pocr_engine *e, int32_t slot);
// the fall-back engine (bf16x3 kernels and weight layouts: call under SplitScope(3)), created on first use from the retained
// weight blob: a second set of weights (~90 MB for the recogniser) and, per slot it ever serves, its own activation buffers
// (17.6 MB per staged line at W_pad 576); an engine whose launches stay in range never pays for either
static int ensure_shadow(pocr_engine *e, int verdict, int which) {
    if (e->weights_host.empty()) return fail("internal error: range guard without a retained weight blob");
    if (!e->warned_range) {
        e->warned_range = true;
        fprintf(stderr, "NOTE: a launch left the range of the default f16x2 arithmetic (%s in activation set %d); it and any later such launch "
                        "are re-run on bf16x3 (fp32's range, ~0.6x the speed).  POCR_CONV_SPLIT=3 selects bf16x3 for everything.\n",
                verdict == 1 ? "|x| >= 65504 or not finite" : verdict == 2 ? "a whole tensor below 2^-13" : "POCR_FORCE_RANGE_FALLBACK", which);
    }
    if (e->shadow.load()) return 0;
    pocr_engine *sh = nullptr;
    if (pocr_create(&e->cfg, e->weights_host.data(), e->weights_host.size(), e->device, &sh)) return 1;
    sh->is_shadow = true;
    e->shadow.store(sh);
    e->shadow_state.store(2);
    return 0;
}
// 1: the fall-back engine exists; 0: it is being built (wait != 0: block until it is there or has failed); -1: this engine has
// none (not f16x2, or POCR_FALLBACK_EAGER=0 and no launch needed it yet, or the background build failed)
int pocr_fallback_ready(pocr_engine *e, int32_t wait) {
    if (!e) return -1;
    while (wait && e->shadow_state.load() == 1) {           // the builder holds shadow_mu from its first instruction to its last
        { std::lock_guard<std::mutex> lock(e->shadow_mu); }
        std::this_thread::yield();
    }
    const int st = e->shadow_state.load();
    return st == 2 ? 1 : st == 1 ? 0 : -1;
}
static int run_fallback(pocr_engine *e, int32_t slot, int verdict, int which) {
    Slot &s = e->slot[slot];
    SplitScope scope(3);
    std::lock_guard<std::mutex> lock(e->shadow_mu);
    if (ensure_shadow(e, verdict, which)) return 1;
    pocr_engine *sh = e->shadow.load();
    sh->lstm_resident = e->lstm_resident; sh->lstm_spin_limit = e->lstm_spin_limit;
    sh->lstm_skip.store(std::max(sh->lstm_skip.load(), e->lstm_skip.load()));
    if (e->cfg.embed_num > 0 && e->embed_id >= 0 && sh->embed_id != e->embed_id && pocr_set_embed_id(sh, e->embed_id)) return 1;
    Slot &t = sh->slot[slot];
    if (t.in_flight && pocr_slot_reset(sh, slot)) return 1;
    const uint8_t *base = s.crops_ext ? s.crops_ext : s.crops.as<uint8_t>();
    if (stage_ragged_impl(sh, slot, nullptr, s.st_off.data(), s.st_w.data(), s.st_wpad.data(), s.n, 0, s.st_padl.data(), base)) return 1;
    t.want_logits = s.want_logits; t.want_argmax = s.want_argmax; t.want_sparse = s.want_sparse;
    t.sp_thr = s.sp_thr; t.sp_has_rows = s.sp_has_rows;
    if (s.want_sparse && s.sp_has_rows) {
        if (t.sp_rows.reserve(s.sp_rows_host.size() * sizeof(int32_t))) return 1;
        HIP_TRY(locked_memcpy(t.sp_rows.p, s.sp_rows_host.data(), s.sp_rows_host.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (run_network(sh, t) || enqueue_outputs(sh, t)) return 1;
    t.in_flight = true;
    sh->last_slot = slot;
    if (sync_and_guard(sh, slot)) return 1;            // waits; a resident recurrence of the fall-back engine that timed out is repeated here too
    s.redirect = true;
    ++e->range_fallbacks;
    return 0;
}

// every read of a launch's results goes through this: wait for the launch, judge its range words once, fall back if needed
static int sync_and_guard(pocr_engine *e, int32_t slot) {
    Slot &s = e->slot[slot];
    // the fall-back engine's kernels and weight layouts are bf16x3 whichever thread reads its results
    SplitScope scope(e->is_shadow ? 3 : g_split_tls);
    HIP_TRY(hipStreamSynchronize(s.seq_stream));
    if (s.lstm_resident_used && s.lstm_err_host) {
        bool timed_out = false;
        for (int l = 0; l < e->cfg.lstm_layers && l < 8; ++l) timed_out = timed_out || s.lstm_err_host[4 * l] != 0;
        if (timed_out) {
            // A cluster's workgroups did not all become resident in time (other tenants on the chip, CU masking, a partitioned
            // device ...): THIS launch is run again on the step kernels - same slot, same requests - before anything of it is
            // read, and the next launches pause the resident path (4, then 8 ... 256 of them) before it is tried again.
            const int pause = std::min(256, std::max(4, 2 * e->lstm_skip_len.load()));
            e->lstm_skip_len.store(pause);
            e->lstm_skip.store(pause);
            e->lstm_ok_run.store(0);
            if (e->lstm_timeouts.fetch_add(1) == 0)
                fprintf(stderr, "NOTE: a hand-off of the resident BiLSTM recurrence timed out; the launch is repeated with one launch per step and "
                                "the resident path pauses for the next launches (pocr_lstm_timeouts counts; POCR_LSTM_RESIDENT=0 turns it off).\n");
            memset(s.lstm_err_host, 0, 8 * 4 * sizeof(uint32_t));
            const bool checked = s.guard_checked;
            s.lstm_force_step = true;
            const int rc = run_network(e, s) || enqueue_outputs(e, s);
            s.lstm_force_step = false;
            if (rc) return 1;
            HIP_TRY(hipStreamSynchronize(s.seq_stream));
            s.guard_checked = checked;
        } else if (!s.lstm_judged && e->lstm_skip_len.load() > 0 && e->lstm_ok_run.fetch_add(1) + 1 >= kLstmDecayAfter) {
            // kLstmDecayAfter resident launches in a row handed over in time: the pause a later timeout would start from is halved
            e->lstm_ok_run.store(0);
            const int len = e->lstm_skip_len.load();
            e->lstm_skip_len.store(len / 2 < 4 ? 0 : len / 2);
        }
        s.lstm_judged = true;
    }
    if (s.guard_checked) return 0;
    s.guard_checked = true;
    int which = 0;
    const int verdict = e->is_shadow ? 0 : range_verdict(s, &which);
    static const bool force = getenv("POCR_FORCE_RANGE_FALLBACK") && atoi(getenv("POCR_FORCE_RANGE_FALLBACK")) != 0;     // tests: every launch takes the fall-back
    if (verdict == 0 && !(force && !e->is_shadow && conv_split() == 2)) return 0;
    return run_fallback(e, slot, verdict, which);
}
// the slot (and engine) whose buffers hold the results of the launch on `slot`
static inline pocr_engine *result_engine(pocr_engine *e, int32_t slot) { return e->slot[slot].redirect ? e->shadow.load() : e; }

int64_t pocr_range_fallbacks(pocr_engine *e) { return e ? e->range_fallbacks : 0; }
int64_t pocr_lstm_timeouts(pocr_engine *e) {
    if (!e) return 0;
    const pocr_engine *sh = e->shadow.load();
    return e->lstm_timeouts.load() + (sh ? sh->lstm_timeouts.load() : 0);
}

int pocr_slot_launch(pocr_engine *e, int32_t slot, int32_t want_logits, int32_t want_argmax) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    if (!s.staged) return fail("slot %d: no chunk staged", slot);
    if (s.in_flight) return fail("slot %d already has a launch in flight", slot);
    if (e->cfg.arch == POCR_ARCH_S2S) return fail("sequence-to-sequence engine: use pocr_s2s_launch / pocr_s2s_decode");
    HIP_TRY(hipSetDevice(e->device));
    s.want_logits = want_logits != 0;
    s.want_argmax = want_argmax != 0;
    s.want_sparse = false;
    if (run_network(e, s)) return 1;
    if (enqueue_outputs(e, s)) return 1;
    s.in_flight = true;
    e->last_slot = slot;
    return 0;
}

int pocr_slot_collect(pocr_engine *e, int32_t slot, float *logits_ntc, int32_t *frame_argmax_nt,
                      int32_t *labels_nt, int32_t *label_len_n) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    if (!s.in_flight) return fail("slot %d has nothing in flight", slot);
    HIP_TRY(hipSetDevice(e->device));
    if (sync_and_guard(e, slot)) { s.in_flight = false; return 1; }
    pocr_engine *r = result_engine(e, slot);
    if (r != e) s.in_flight = false;
    return collect_outputs(r, r->slot[slot], logits_ntc, frame_argmax_nt, labels_nt, label_len_n);
}

int pocr_slot_launch_sparse(pocr_engine *e, int32_t slot, const int32_t *row_begin, const int32_t *row_end,
                            float threshold, int32_t want_argmax) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    if (!s.staged) return fail("slot %d: no chunk staged", slot);
    if (s.in_flight) return fail("slot %d already has a launch in flight", slot);
    if ((row_begin == nullptr) != (row_end == nullptr)) return fail("row_begin and row_end must both be given or both be NULL");
    if (e->cfg.arch == POCR_ARCH_S2S) return fail("sequence-to-sequence engine: use pocr_s2s_launch / pocr_s2s_decode");
    HIP_TRY(hipSetDevice(e->device));
    s.sp_has_rows = row_begin != nullptr;
    if (s.sp_has_rows) {
        std::vector<int32_t> rows(2 * (size_t)s.n);
        for (int i = 0; i < s.n; ++i) {
            const int Ti = (int)(reinterpret_cast<const int32_t *>(s.geom_host.data() + s.h_lvl2_off)[i]);
            if (row_begin[i] < 0 || row_end[i] > Ti || row_begin[i] > row_end[i])
                return fail("line %d: row range [%d, %d) outside [0, %d]", i, row_begin[i], row_end[i], Ti);
            rows[i] = row_begin[i]; rows[s.n + i] = row_end[i];
        }
        s.sp_rows_host = rows;
        if (s.sp_rows.reserve(rows.size() * sizeof(int32_t))) return 1;
        HIP_TRY(hipMemcpyAsync(s.sp_rows.p, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipStreamSynchronize(s.stream));       // `rows` is a stack-lifetime pageable buffer
    }
    s.want_logits = false;
    s.want_argmax = want_argmax != 0;
    s.want_sparse = true;
    s.sp_thr = threshold;
    if (run_network(e, s)) return 1;
    if (enqueue_outputs(e, s)) return 1;
    s.in_flight = true;
    e->last_slot = slot;
    return 0;
}

int pocr_slot_sparse_nnz(pocr_engine *e, int32_t slot, int64_t *total_nnz) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    if (!s.in_flight || !s.want_sparse) return fail("slot %d has no sparse launch in flight", slot);
    if (!total_nnz) return fail("total_nnz is NULL");
    HIP_TRY(hipSetDevice(e->device));
    if (sync_and_guard(e, slot)) return 1;
    const Slot &r = result_engine(e, slot)->slot[slot];
    *total_nnz = static_cast<const int64_t *>(r.sp_pinned)[r.n];
    return 0;
}

int pocr_slot_collect_sparse(pocr_engine *e, int32_t slot, float *data, int32_t *indices, int32_t *indptr,
                             int64_t *line_off, int32_t *frame_argmax_nt, int32_t *labels_nt, int32_t *label_len_n) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    if (!s.in_flight || !s.want_sparse) return fail("slot %d has no sparse launch in flight", slot);
    if (!data || !indices || !indptr || !line_off) return fail("NULL output pointer");
    HIP_TRY(hipSetDevice(e->device));
    if (sync_and_guard(e, slot)) { s.in_flight = false; return 1; }
    if (s.redirect) {                   // the launch was re-run on the fall-back engine: its slot holds the results
        s.in_flight = false;
        return pocr_slot_collect_sparse(e->shadow, slot, data, indices, indptr, line_off, frame_argmax_nt, labels_nt, label_len_n);
    }
    const int n = s.n, C = e->cfg.num_classes;
    const size_t off_bytes = (size_t)(n + 1) * sizeof(int64_t), ip_bytes = (size_t)n * (C + 1) * sizeof(int32_t);
    const char *sp = static_cast<const char *>(s.sp_pinned);
    const int64_t total = reinterpret_cast<const int64_t *>(sp)[n];
    memcpy(line_off, sp, off_bytes);
    memcpy(indptr, sp + off_bytes, ip_bytes);
    if (total > 0) {
        const size_t trip_base = (off_bytes + ip_bytes + (size_t)n * sizeof(float) + 15) / 16 * 16, have = std::min<size_t>(s.sp_spec, (size_t)total);
        const char *pin = static_cast<const char *>(s.sp_pinned) + trip_base;
        parallel_memcpy(data, pin, have * sizeof(float));                  // (flat posteriors: 30 MB each per 256 lines, into fresh pages of the caller's arrays)
        parallel_memcpy(indices, pin + s.sp_spec * sizeof(float), have * sizeof(int32_t));
        if ((size_t)total > have) {      // the speculative copy was too short: fetch the tail now
            HIP_TRY(hipMemcpyAsync(data + have, s.sp_data.as<float>() + have, ((size_t)total - have) * sizeof(float), hipMemcpyDeviceToHost, s.seq_stream));
            HIP_TRY(hipMemcpyAsync(indices + have, s.sp_indices.as<int32_t>() + have, ((size_t)total - have) * sizeof(int32_t), hipMemcpyDeviceToHost, s.seq_stream));
        }
    }
    e->sp_prev_total = (size_t)total;
    e->sp_prev_rows = s.rows;
    return collect_outputs(e, s, nullptr, frame_argmax_nt, labels_nt, label_len_n);
}


int pocr_slot_confidence(pocr_engine *e, int32_t slot, float *confidence_n) {
    if (check_slot(e, slot)) return 1;
    if (e->slot[slot].redirect && e->shadow) return pocr_slot_confidence(e->shadow, slot, confidence_n);
    Slot &s = e->slot[slot];
    if (!s.want_sparse || !s.sp_pinned || !s.staged) return fail("slot %d: no collected sparse launch", slot);
    if (s.in_flight) return fail("slot %d: collect the launch first", slot);
    if (!confidence_n) return fail("confidence_n is NULL");
    const int n = s.n, C = e->cfg.num_classes;
    const size_t conf_off = (size_t)(n + 1) * sizeof(int64_t) + (size_t)n * (C + 1) * sizeof(int32_t);
    memcpy(confidence_n, static_cast<const char *>(s.sp_pinned) + conf_off, (size_t)n * sizeof(float));
    return 0;
}

int pocr_slot_reset(pocr_engine *e, int32_t slot) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    HIP_TRY(hipSetDevice(e->device));
    // drain whatever the slot still has queued (both of its streams), then forget the launch
    hipError_t e1 = hipStreamSynchronize(s.stream), e2 = hipStreamSynchronize(s.seq_stream);
    s.in_flight = false;
    s.s2s_launched = s.s2s_decoded = false;
    s.staged = false;
    (void)hipGetLastError();
    if (e1 != hipSuccess || e2 != hipSuccess)
        return fail("slot %d: device error while draining: %s", slot, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return 0;
}

// ------------------------------------------------------------------ multi-GPU exchange (comm.hpp)

#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t _r = (expr);                                                               \
        if (_r != ncclSuccess) return fail("%s failed: %s", #expr, rccl().GetErrorString(_r));  \
    } while (0)

int pocr_comm_unique_id(uint8_t *id128) {
    if (!id128) return fail("id128 is NULL");
    if (rccl().load()) return fail("%s", rccl().err.c_str());
    ncclUniqueId id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    static_assert(sizeof(id) == POCR_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int pocr_comm_init(pocr_engine *e, const uint8_t *id128, int32_t rank, int32_t world) {
    if (!e) return fail("engine is NULL");
    if (!id128) return fail("id128 is NULL");
    if (world < 1 || rank < 0 || rank >= world) return fail("rank %d / world %d invalid", rank, world);
    if (e->comm.active()) return fail("engine already has a communicator");
    if (rccl().load()) return fail("%s", rccl().err.c_str());
    HIP_TRY(hipSetDevice(e->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    HIP_TRY(hipStreamCreateWithFlags(&e->comm.stream, hipStreamNonBlocking));
    ncclResult_t r = rccl().CommInitRank(&e->comm.comm, world, id, rank);
    if (r != ncclSuccess) {
        (void)hipStreamDestroy(e->comm.stream);
        e->comm = Comm{};
        return fail("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl().GetErrorString(r));
    }
    e->comm.rank = rank; e->comm.world = world;
    return 0;
}

// What the communicator itself says (ncclCommCount / ncclCommUserRank), not what the caller passed to pocr_comm_init: a
// bench line that claims N RCCL ranks carries these numbers.  No communicator: *count = *rank = 0, returns 0.
int pocr_comm_info(pocr_engine *e, int32_t *count, int32_t *rank) {
    if (!e) return fail("engine is NULL");
    if (!count || !rank) return fail("count / rank is NULL");
    *count = 0; *rank = 0;
    if (!e->comm.active()) return 0;
    int c = 0, r = 0;
    NCCL_TRY(rccl().CommCount(e->comm.comm, &c));
    NCCL_TRY(rccl().CommUserRank(e->comm.comm, &r));
    *count = c; *rank = r;
    return 0;
}

static void comm_release(pocr_engine *e) {
    Comm &c = e->comm;
    if (c.comm) (void)rccl().CommDestroy(c.comm);
    { UnsafeLock l; if (c.d_send) (void)hipFree(c.d_send); if (c.d_recv) (void)hipFree(c.d_recv); }
    if (c.h_send) (void)locked_host_free(c.h_send);
    if (c.h_recv) (void)locked_host_free(c.h_recv);
    if (c.stream) (void)hipStreamDestroy(c.stream);
    c = Comm{};
}

int pocr_comm_destroy(pocr_engine *e) {
    if (!e) return fail("engine is NULL");
    if (!e->comm.active()) return 0;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->comm.stream);
    comm_release(e);
    return 0;
}

static int comm_reserve(pocr_engine *e, size_t send_bytes) {
    Comm &c = e->comm;
    const size_t recv_bytes = send_bytes * (size_t)c.world;
    if (send_bytes > c.send_cap) {
        if (c.d_send) { UnsafeLock l; (void)hipFree(c.d_send); }
        if (c.h_send) (void)locked_host_free(c.h_send);
        c.d_send = c.h_send = nullptr; c.send_cap = 0;
        const size_t want = send_bytes + send_bytes / 4 + 256;
        { UnsafeLock l; HIP_TRY(hipMalloc(&c.d_send, want)); }
        HIP_TRY(locked_host_malloc(&c.h_send, want, hipHostMallocDefault));
        c.send_cap = want;
    }
    if (recv_bytes > c.recv_cap) {
        if (c.d_recv) { UnsafeLock l; (void)hipFree(c.d_recv); }
        if (c.h_recv) (void)locked_host_free(c.h_recv);
        c.d_recv = c.h_recv = nullptr; c.recv_cap = 0;
        const size_t want = recv_bytes + recv_bytes / 4 + 256;
        { UnsafeLock l; HIP_TRY(hipMalloc(&c.d_recv, want)); }
        HIP_TRY(locked_host_malloc(&c.h_recv, want, hipHostMallocDefault));
        c.recv_cap = want;
    }
    return 0;
}

int pocr_allgather_labels(pocr_engine *e, const int32_t *send, int64_t count, int32_t *recv) {
    if (!e) return fail("engine is NULL");
    if (!e->comm.active()) return fail("no communicator: call pocr_comm_init first");
    if (count < 0 || (count > 0 && (!send || !recv))) return fail("invalid buffers / count");
    if (count == 0) return 0;
    HIP_TRY(hipSetDevice(e->device));
    Comm &c = e->comm;
    const size_t bytes = (size_t)count * sizeof(int32_t);
    if (comm_reserve(e, bytes)) return 1;
    memcpy(c.h_send, send, bytes);
    HIP_TRY(hipMemcpyAsync(c.d_send, c.h_send, bytes, hipMemcpyHostToDevice, c.stream));
    NCCL_TRY(rccl().AllGather(c.d_send, c.d_recv, (size_t)count, ncclInt32, c.comm, c.stream));
    HIP_TRY(hipMemcpyAsync(c.h_recv, c.d_recv, bytes * c.world, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    memcpy(recv, c.h_recv, bytes * c.world);
    return 0;
}

int pocr_comm_allreduce_max(pocr_engine *e, double *value) {
    if (!e) return fail("engine is NULL");
    if (!e->comm.active()) return fail("no communicator: call pocr_comm_init first");
    if (!value) return fail("value is NULL");
    HIP_TRY(hipSetDevice(e->device));
    Comm &c = e->comm;
    if (comm_reserve(e, sizeof(double))) return 1;
    memcpy(c.h_send, value, sizeof(double));
    HIP_TRY(hipMemcpyAsync(c.d_send, c.h_send, sizeof(double), hipMemcpyHostToDevice, c.stream));
    NCCL_TRY(rccl().AllReduce(c.d_send, c.d_recv, 1, ncclDouble, ncclMax, c.comm, c.stream));
    HIP_TRY(hipMemcpyAsync(c.h_recv, c.d_recv, sizeof(double), hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    memcpy(value, c.h_recv, sizeof(double));
    return 0;
}

int pocr_device_synchronize(pocr_engine *e) {
    if (!e) return fail("engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(locked_device_sync());
    return 0;
}

// ------------------------------------------------------------------ sequence-to-sequence (decoder.hpp)

int pocr_s2s_stage(pocr_engine *e, int32_t slot, const uint8_t *crops, const int64_t *crop_offsets,
                   const int32_t *widths, const int32_t *w_pads, const int32_t *pad_lefts, int32_t n) {
    if (check_slot(e, slot)) return 1;
    if (e->cfg.arch != POCR_ARCH_S2S) return fail("pocr_s2s_stage needs a POCR_ARCH_S2S engine");
    if (!pad_lefts) return fail("NULL input pointer");
    for (int i = 0; i < n; ++i)
        if (pad_lefts[i] < 0) return fail("line %d: pad_left must be >= 0", i);
    if (stage_ragged_impl(e, slot, crops, crop_offsets, widths, w_pads, n, 0, pad_lefts)) return 1;
    e->slot[slot].s2s_wpads.assign(w_pads, w_pads + n);
    return 0;
}

int pocr_s2s_launch(pocr_engine *e, int32_t slot, const int32_t *batch_first, int32_t n_batches) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    if (e->cfg.arch != POCR_ARCH_S2S) return fail("pocr_s2s_launch needs a POCR_ARCH_S2S engine");
    if (!s.staged || (int)s.s2s_wpads.size() != s.n) return fail("slot %d: nothing staged with pocr_s2s_stage", slot);
    if (s.in_flight) return fail("slot %d already has a launch in flight", slot);
    if (!batch_first || n_batches < 1) return fail("batch_first / n_batches invalid");
    if (batch_first[0] != 0 || batch_first[n_batches] != s.n) return fail("batch_first must start at 0 and end at n = %d", s.n);
    HIP_TRY(hipSetDevice(e->device));
    s.s2s_batch_first.assign(batch_first, batch_first + n_batches + 1);
    s.s2s_limit.resize(n_batches);
    int cap = 0;
    for (int b = 0; b < n_batches; ++b) {
        if (batch_first[b + 1] <= batch_first[b]) return fail("batch %d is empty", b);
        const int wp = s.s2s_wpads[batch_first[b]];
        for (int i = batch_first[b]; i < batch_first[b + 1]; ++i)
            if (s.s2s_wpads[i] != wp) return fail("batch %d: all lines of a batch must have the same w_pad", b);
        s.s2s_limit[b] = wp / 4;                       // inputs.shape[-1] // 4 (transformer_ocr_engine.py:77)
        cap = std::max(cap, wp / 4 + 1);
    }
    if (cap > DEC_MAX_KEYS || s.t_max > DEC_MAX_KEYS) return fail("line too long for the decoder (%d steps / %d frames, limit %d)", cap, s.t_max, DEC_MAX_KEYS);
    s.s2s_batches = n_batches;
    s.s2s_cap = cap;
    if (run_network(e, s)) return 1;                   // encoder + key/value projections, asynchronous
    s.in_flight = true;
    s.s2s_launched = true;
    s.s2s_decoded = false;
    e->last_slot = slot;
    return 0;
}


int pocr_s2s_decode(pocr_engine *e, int32_t slot, int32_t want_logits, int32_t *steps, int32_t *s_max) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    if (!s.s2s_launched || !s.in_flight) return fail("slot %d: pocr_s2s_launch first", slot);
    if (!steps || !s_max) return fail("NULL output pointer");
    HIP_TRY(hipSetDevice(e->device));
    const pocr_config &c = e->cfg;
    const int n = s.n, E = c.conv_out, FF = c.sa_ff, C = c.num_classes, heads = c.sa_heads, D = E / heads;
    const int L = c.dec_layers, S_cap = s.s2s_cap, nb = s.s2s_batches;
    hipStream_t st = s.seq_stream;
    const size_t ne = (size_t)n * E * sizeof(float);
    for (DevBuf *b : {&s.s2s_x, &s.s2s_x1, &s.s2s_x2, &s.s2s_t, &s.s2s_ctx, &s.s2s_q})
        if (b->reserve(ne)) return 1;
    if (s.s2s_ff.reserve((size_t)n * FF * sizeof(float))) return 1;
    if (s.s2s_logits.reserve((size_t)n * S_cap * C * sizeof(float))) return 1;
    if (s.s2s_tokens.reserve((size_t)n * S_cap * sizeof(int32_t))) return 1;
    for (int l = 0; l < L; ++l)
        if (s.s2s_cache[l].reserve((size_t)S_cap * n * 3 * E * sizeof(float))) return 1;
    // tables: batch_first [nb+1] | limit [nb] | line_batch [n];  state: alive [n] | batch_done [nb] | steps [nb] | remaining
    std::vector<int32_t> tab((size_t)nb + 1 + nb + n);
    memcpy(tab.data(), s.s2s_batch_first.data(), (size_t)(nb + 1) * sizeof(int32_t));
    memcpy(tab.data() + nb + 1, s.s2s_limit.data(), (size_t)nb * sizeof(int32_t));
    for (int b = 0; b < nb; ++b)
        for (int i = s.s2s_batch_first[b]; i < s.s2s_batch_first[b + 1]; ++i) tab[2 * nb + 1 + i] = b;
    if (s.s2s_tables.reserve(tab.size() * sizeof(int32_t))) return 1;
    if (s.s2s_state.reserve(((size_t)2 * n + 2 * nb + 1) * sizeof(int32_t))) return 1;
    HIP_TRY(hipMemcpyAsync(s.s2s_tables.p, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));                 // `tab` is pageable and dies with this scope; also: the encoder is done
    if (!s.guard_checked) {
        s.guard_checked = true;
        int which = 0;
        if (const int verdict = range_verdict(s, &which)) {
            // the encoder left f16's range: the same lines and batches again on the bf16x3 engine (plain fp32 in the reference:
            // pero_ocr/ocr_engine/transformer_ocr_engine.py:32-47); this slot's reads are redirected to its slot there
            s.in_flight = false; s.s2s_launched = false;
            SplitScope scope(3);
            std::lock_guard<std::mutex> lock(e->shadow_mu);
            if (ensure_shadow(e, verdict, which)) return 1;
            pocr_engine *sh = e->shadow.load();
            Slot &t = sh->slot[slot];
            if (t.in_flight && pocr_slot_reset(sh, slot)) return 1;
            const uint8_t *base = s.crops_ext ? s.crops_ext : s.crops.as<uint8_t>();
            if (stage_ragged_impl(sh, slot, nullptr, s.st_off.data(), s.st_w.data(), s.st_wpad.data(), s.n, 0, s.st_padl.data(), base)) return 1;
            t.s2s_wpads = s.s2s_wpads;
            if (pocr_s2s_launch(sh, slot, s.s2s_batch_first.data(), s.s2s_batches)) return 1;
            if (pocr_s2s_decode(sh, slot, want_logits, steps, s_max)) return 1;
            s.redirect = true; s.s2s_decoded = true;
            ++e->range_fallbacks;
            return 0;
        }
    }
    const int32_t *d_batch_first = s.s2s_tables.as<int32_t>(), *d_limit = d_batch_first + nb + 1;
    int32_t *d_alive = s.s2s_state.as<int32_t>(), *d_done = d_alive + n, *d_steps = d_done + nb, *d_remaining = d_steps + nb;
    int32_t *d_line_done = d_remaining + 1;
    constexpr int BLK = 8, MAXBLK = (DEC_MAX_KEYS + BLK) / BLK + 2;
    const size_t flags_need = ((size_t)MAXBLK + nb) * sizeof(int32_t);
    if (flags_need > s.s2s_flags_cap) {
        if (s.s2s_flags) (void)locked_host_free(s.s2s_flags);
        s.s2s_flags = nullptr; s.s2s_flags_cap = 0;
        HIP_TRY(locked_host_malloc(reinterpret_cast<void **>(&s.s2s_flags), 2 * flags_need, hipHostMallocDefault));
        s.s2s_flags_cap = 2 * flags_need;
    }
    HIP_TRY(hipMemsetAsync(s.s2s_tokens.p, 0xFF, (size_t)n * S_cap * sizeof(int32_t), st));
    if (want_logits) HIP_TRY(hipMemsetAsync(s.s2s_logits.p, 0, (size_t)n * S_cap * C * sizeof(float), st));

    S2sState stt{};
    stt.tokens = s.s2s_tokens.as<int32_t>(); stt.alive = d_alive; stt.batch_done = d_done; stt.steps = d_steps;
    stt.line_done = d_line_done;
    stt.remaining = d_remaining; stt.batch_first = d_batch_first; stt.limit = d_limit;
    stt.embed = e->dec_embed.as<float>(); stt.pe = e->pe.as<float>(); stt.x = s.s2s_x.as<float>();
    stt.n = n; stt.n_batches = nb; stt.S_cap = S_cap; stt.C = C; stt.E = E; stt.boundary = C - 2;
    hipLaunchKernelGGL(s2s_init_kernel, dim3(std::max(1, std::min(1024, (n * E + 255) / 256))), dim3(256), 0, st, stt);
    HIP_TRY(hipGetLastError());

    auto gemm = [&](const float *x_, int64_t ldx, const DevBuf &w_, const DevBuf &b_, int cout_, int K_, float *y_, int64_t ldy, bool relu) {
        SkinnyArgs a{};
        a.x = x_; a.wfrag = w_.as<float>(); a.bias = b_.as<float>(); a.y = y_; a.ldx = ldx; a.ldy = ldy;
        a.M = n; a.K = K_; a.cout16 = round_up(cout_, kSkinnyNT) / 16; a.cout_valid = cout_; a.stop = d_remaining;
        return relu ? launch_skinny<true>(a, st) : launch_skinny<false>(a, st);
    };
    auto ln = [&](const float *a_, const float *b_, const DevBuf &gw, const DevBuf &gb, float *y_) {
        hipLaunchKernelGGL(layernorm_kernel, dim3((n + 3) / 4), dim3(256), 0, st, a_, b_, gw.as<float>(), gb.as<float>(),
                           (const float *)nullptr, y_, n, E, 1, 1e-5f, (const int32_t *)nullptr, (const int32_t *)d_remaining, (void *)nullptr, (unsigned *)nullptr);
    };
    auto attend = [&](DecAttnArgs a) {
        a.out = s.s2s_ctx.as<float>(); a.line_done = d_line_done; a.stop = d_remaining;
        a.E = E; a.scale = 1.0f / sqrtf((float)D);
        const dim3 grid(heads, n);
        if (a.row_off) {           // memory attention (encoder rows of the line)
            if (D == 32) hipLaunchKernelGGL((dec_attention_kernel<32, true>), grid, dim3(256), 0, st, a);
            else if (D == 64) hipLaunchKernelGGL((dec_attention_kernel<64, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((dec_attention_kernel<128, true>), grid, dim3(256), 0, st, a);
        } else if (D == 32) hipLaunchKernelGGL((dec_attention_kernel<32, false>), grid, dim3(256), 0, st, a);
        else if (D == 64) hipLaunchKernelGGL((dec_attention_kernel<64, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((dec_attention_kernel<128, false>), grid, dim3(256), 0, st, a);
    };
    const bool fuse_ln = E == 512 || E == 256;
    int blocks = 0;
    bool finished = false;
    for (int step = 0; step < S_cap && !finished; ++step) {
        if (step % BLK == 0 && step >= 2 * BLK) {          // look at the block before the previous one: two blocks stay in flight
            const int k = step / BLK - 2;
            HIP_TRY(hipEventSynchronize(s.s2s_ev[k & 1]));
            if (s.s2s_flags[k] == 0) { finished = true; break; }
        }
        float *x = s.s2s_x.as<float>(), *x1 = s.s2s_x1.as<float>(), *x2 = s.s2s_x2.as<float>(), *t = s.s2s_t.as<float>();
        float *ctx = s.s2s_ctx.as<float>(), *q = s.s2s_q.as<float>(), *ff = s.s2s_ff.as<float>();
        // LN(x_in + t) feeding a GEMM: fused into the GEMM's prologue when an instance exists, else LayerNorm kernel + GEMM
        auto ln_gemm = [&](const float *res, const DevBuf &gw, const DevBuf &gb, float *normed, const DevBuf &w_, const DevBuf &b_,
                           int cout_, float *y_, int64_t ldy, bool relu) -> int {
            if (fuse_ln) {
                SkinnyArgs a{};
                a.x = normed; a.wfrag = w_.as<float>(); a.bias = b_.as<float>(); a.y = y_; a.ldx = E; a.ldy = ldy;
                a.M = n; a.K = E; a.cout16 = round_up(cout_, kSkinnyNT) / 16; a.cout_valid = cout_; a.stop = d_remaining;
                a.ln_a = res; a.ln_b = t; a.gamma = gw.as<float>(); a.beta = gb.as<float>(); a.ln_out = normed; a.eps = 1e-5f;
                if (E == 512) return relu ? launch_skinny_ln<true, 512>(a, st) : launch_skinny_ln<false, 512>(a, st);
                return relu ? launch_skinny_ln<true, 256>(a, st) : launch_skinny_ln<false, 256>(a, st);
            }
            ln(res, t, gw, gb, normed);
            return gemm(normed, E, w_, b_, cout_, E, y_, ldy, relu);
        };
        for (int l = 0; l < L; ++l) {
            pocr_engine::DecLayer &W = e->dec[l];
            float *cache = s.s2s_cache[l].as<float>();
            float *row = cache + (size_t)step * n * 3 * E;          // linear_cache[seq_len - 1] (transformer.py:250)
            if (l == 0) {
                if (gemm(x, E, W.ws_in, W.bs_in, 3 * E, E, row, 3 * E, false)) return 1;
            } else {                                                 // x = norm3 of the layer below (x2 + t)
                if (ln_gemm(x2, e->dec[l - 1].n3w, e->dec[l - 1].n3b, x, W.ws_in, W.bs_in, 3 * E, row, 3 * E, false)) return 1;
            }
            {
                DecAttnArgs a{};
                a.q = row; a.ldq = 3 * E; a.k = cache + E; a.v = cache + 2 * E;
                a.pos_stride = (int64_t)n * 3 * E; a.line_stride = 3 * E; a.len = step + 1;
                attend(a);
            }
            if (gemm(ctx, E, W.ws_out, W.bs_out, E, E, t, E, false)) return 1;
            if (ln_gemm(x, W.n1w, W.n1b, x1, W.wc_q, W.bc_q, E, q, E, false)) return 1;          // x1 = norm1(x + t); q = x1 Wq
            {
                DecAttnArgs a{};
                a.q = q; a.ldq = E; a.k = s.s2s_kv[l].as<float>(); a.v = s.s2s_kv[l].as<float>() + E;
                a.pos_stride = 2 * E; a.row_off = s.g_row_off; a.line_len = s.g_line_T;
                attend(a);
            }
            if (gemm(ctx, E, W.wc_out, W.bc_out, E, E, t, E, false)) return 1;
            if (ln_gemm(x1, W.n2w, W.n2b, x2, W.w1, W.b1, FF, ff, FF, true)) return 1;          // x2 = norm2(x1 + t); ff = relu(x2 W1)
            if (gemm(ff, FF, W.w2, W.b2, E, FF, t, E, false)) return 1;
        }
        float *lg = s.s2s_logits.as<float>() + (size_t)step * C;
        // x = norm3(x2 + t) of the last layer; logits = x W_out
        if (ln_gemm(x2, e->dec[L - 1].n3w, e->dec[L - 1].n3b, x, e->head_w, e->head_b, C, lg, (int64_t)S_cap * C, false)) return 1;
        hipLaunchKernelGGL(s2s_sample_kernel, dim3(nb), dim3(256), 0, st, stt, (const float *)lg, (int64_t)S_cap * C, step);
        HIP_TRY(hipGetLastError());
        if (step % BLK == BLK - 1) {
            const int k = step / BLK;
            HIP_TRY(hipMemcpyAsync(&s.s2s_flags[k], d_remaining, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(s.s2s_ev[k & 1], st));
            blocks = k + 1;
        }
    }
    (void)blocks;
    int32_t *h_steps = s.s2s_flags + MAXBLK;
    HIP_TRY(hipMemcpyAsync(h_steps, d_steps, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_steps + nb, d_remaining, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_steps[nb] != 0) return fail("internal error: %d batches still decoding after %d steps", h_steps[nb], S_cap);
    int smax = 0;
    s.s2s_steps.assign(h_steps, h_steps + nb);
    for (int b = 0; b < nb; ++b) { steps[b] = h_steps[b]; smax = std::max(smax, h_steps[b]); }
    *s_max = smax;
    s.s2s_smax = smax;
    s.s2s_want_logits = want_logits != 0;
    // results -> pinned: tokens [n][S_cap] | logits [n][smax][C]
    const size_t tok_bytes = (size_t)n * S_cap * sizeof(int32_t);
    const size_t lg_bytes = want_logits ? (size_t)n * smax * C * sizeof(float) : 0;
    if (tok_bytes + lg_bytes > s.s2s_pinned_cap) {
        if (s.s2s_pinned) (void)locked_host_free(s.s2s_pinned);
        s.s2s_pinned = nullptr; s.s2s_pinned_cap = 0;
        const size_t want = (tok_bytes + lg_bytes) * 5 / 4;
        HIP_TRY(locked_host_malloc(&s.s2s_pinned, want, hipHostMallocDefault));
        s.s2s_pinned_cap = want;
    }
    char *pin = static_cast<char *>(s.s2s_pinned);
    HIP_TRY(hipMemcpyAsync(pin, s.s2s_tokens.p, tok_bytes, hipMemcpyDeviceToHost, st));
    if (lg_bytes)
        HIP_TRY(hipMemcpy2DAsync(pin + tok_bytes, (size_t)smax * C * sizeof(float), s.s2s_logits.p, (size_t)S_cap * C * sizeof(float),
                                 (size_t)smax * C * sizeof(float), (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    s.s2s_decoded = true;
    s.in_flight = false;
    return 0;
}

int pocr_s2s_collect(pocr_engine *e, int32_t slot, int32_t *tokens, float *logits) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    if (!s.s2s_decoded) return fail("slot %d: pocr_s2s_decode first", slot);
    if (s.redirect && e->shadow) return pocr_s2s_collect(e->shadow, slot, tokens, logits);     // decoded on the fall-back engine (range guard)
    if (logits && !s.s2s_want_logits) return fail("logits were not requested at pocr_s2s_decode");
    const int n = s.n, S_cap = s.s2s_cap, smax = s.s2s_smax, C = e->cfg.num_classes;
    const char *pin = static_cast<const char *>(s.s2s_pinned);
    if (tokens)
        for (int i = 0; i < n; ++i)
            memcpy(tokens + (size_t)i * smax, pin + (size_t)i * S_cap * sizeof(int32_t), (size_t)smax * sizeof(int32_t));
    if (logits) parallel_memcpy(logits, pin + (size_t)n * S_cap * sizeof(int32_t), (size_t)n * smax * C * sizeof(float));
    return 0;
}


int pocr_s2s_sparse(pocr_engine *e, int32_t slot, const int32_t *row_end, float threshold, int64_t *total_nnz) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    if (!s.s2s_decoded) return fail("slot %d: pocr_s2s_decode first", slot);
    if (s.redirect && e->shadow) { SplitScope scope(3); return pocr_s2s_sparse(e->shadow, slot, row_end, threshold, total_nnz); }
    if (!row_end || !total_nnz) return fail("NULL pointer");
    HIP_TRY(hipSetDevice(e->device));
    const int n = s.n, S_cap = s.s2s_cap, C = e->cfg.num_classes;
    if (C > 256 * SP_COLS) return fail("sparse logits: C = %d exceeds %d classes", C, 256 * SP_COLS);
    size_t cap = 0;
    for (int i = 0; i < n; ++i) {
        if (row_end[i] < 0 || row_end[i] > S_cap) return fail("line %d: row_end %d outside [0, %d]", i, row_end[i], S_cap);
        cap += (size_t)row_end[i] * C;
    }
    hipStream_t st = s.seq_stream;
    if (s.sp_rowstat.reserve((size_t)n * S_cap * 2 * sizeof(float)) || s.sp_colcount.reserve((size_t)n * sp_blocks(S_cap) * C * sizeof(int32_t)) ||
        s.sp_line_nnz.reserve((size_t)n * sizeof(int32_t)) || s.sp_line_off.reserve((size_t)(n + 1) * sizeof(int64_t)) ||
        s.sp_indptr.reserve((size_t)n * (C + 1) * sizeof(int32_t)) || s.sp_data.reserve(std::max<size_t>(cap, 1) * sizeof(float)) ||
        s.sp_indices.reserve(std::max<size_t>(cap, 1) * sizeof(int32_t)) || s.sp_rows.reserve((size_t)n * sizeof(int32_t)))
        return 1;
    const size_t off_bytes = (size_t)(n + 1) * sizeof(int64_t), ip_bytes = (size_t)n * (C + 1) * sizeof(int32_t);
    const size_t need_sp = off_bytes + ip_bytes + (size_t)n * sizeof(int32_t) + cap * 8;
    if (need_sp > s.sp_pinned_cap) {
        if (s.sp_pinned) (void)locked_host_free(s.sp_pinned);
        s.sp_pinned = nullptr; s.sp_pinned_cap = 0;
        HIP_TRY(locked_host_malloc(&s.sp_pinned, need_sp + need_sp / 4, hipHostMallocDefault));
        s.sp_pinned_cap = need_sp + need_sp / 4;
    }
    char *sp = static_cast<char *>(s.sp_pinned);
    int32_t *h_rows = reinterpret_cast<int32_t *>(sp + off_bytes + ip_bytes);
    memcpy(h_rows, row_end, (size_t)n * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(s.sp_rows.p, h_rows, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    const int32_t *r1 = s.sp_rows.as<int32_t>();
    // decoder logits: line i owns rows [i * S_cap, (i + 1) * S_cap); rows [0, row_end[i]) are compacted
    sparsify_launch(st, s.s2s_logits.as<float>(), nullptr, r1, s.sp_rowstat.as<float>(), s.sp_colcount.as<int32_t>(), s.sp_line_nnz.as<int32_t>(),
                    s.sp_line_off.as<int64_t>(), s.sp_indptr.as<int32_t>(), s.sp_data.as<float>(), s.sp_indices.as<int32_t>(), nullptr, nullptr, nullptr,
                    n, S_cap, S_cap, C, threshold, -80.0f, (int64_t)std::max<size_t>(cap, 1), nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(sp, s.sp_line_off.p, off_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(sp + off_bytes, s.sp_indptr.p, ip_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t total = reinterpret_cast<const int64_t *>(sp)[n];
    char *trip = sp + off_bytes + ip_bytes + (size_t)n * sizeof(int32_t);
    if (total > 0) {
        HIP_TRY(hipMemcpyAsync(trip, s.sp_data.p, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(trip + (size_t)total * sizeof(float), s.sp_indices.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    *total_nnz = total;
    return 0;
}

int pocr_s2s_collect_sparse(pocr_engine *e, int32_t slot, float *data, int32_t *indices, int32_t *indptr, int64_t *line_off) {
    if (check_slot(e, slot)) return 1;
    Slot &s = e->slot[slot];
    if (s.redirect && e->shadow) return pocr_s2s_collect_sparse(e->shadow, slot, data, indices, indptr, line_off);
    if (!s.s2s_decoded || !s.sp_pinned) return fail("slot %d: pocr_s2s_sparse first", slot);
    if (!data || !indices || !indptr || !line_off) return fail("NULL output pointer");
    const int n = s.n, C = e->cfg.num_classes;
    const size_t off_bytes = (size_t)(n + 1) * sizeof(int64_t), ip_bytes = (size_t)n * (C + 1) * sizeof(int32_t);
    const char *sp = static_cast<const char *>(s.sp_pinned);
    const int64_t total = reinterpret_cast<const int64_t *>(sp)[n];
    memcpy(line_off, sp, off_bytes);
    memcpy(indptr, sp + off_bytes, ip_bytes);
    const char *trip = sp + off_bytes + ip_bytes + (size_t)n * sizeof(int32_t);
    if (total > 0) {
        memcpy(data, trip, (size_t)total * sizeof(float));
        memcpy(indices, trip + (size_t)total * sizeof(float), (size_t)total * sizeof(int32_t));
    }
    return 0;
}

int pocr_stage_lines(pocr_engine *e, const uint8_t *crops, const int64_t *crop_offsets, const int32_t *widths,
                     int32_t n, int32_t w_pad, int32_t pad_left) {
    if (pocr_slot_stage_lines(e, 0, crops, crop_offsets, widths, n, w_pad, pad_left)) return 1;
    HIP_TRY(hipStreamSynchronize(e->slot[0].stream));       // resident in HBM on return
    return 0;
}

int pocr_run_staged(pocr_engine *e, float *logits_ntc, int32_t *frame_argmax_nt, int32_t *labels_nt, int32_t *label_len_n) {
    if (pocr_slot_launch(e, 0, logits_ntc != nullptr, frame_argmax_nt != nullptr)) return 1;
    return pocr_slot_collect(e, 0, logits_ntc, frame_argmax_nt, labels_nt, label_len_n);
}

int pocr_run_batch(pocr_engine *e, const uint8_t *batch_nhwc, int32_t n, int32_t w_pad, float *logits_ntc,
                   int32_t *frame_argmax_nt, int32_t *labels_nt, int32_t *label_len_n) {
    if (!e) return fail("engine is NULL");
    if (n <= 0) return fail("n must be positive (got %d)", n);
    if (w_pad < 4) return fail("w_pad must be >= 4 (got %d)", w_pad);
    std::vector<int64_t> off(n);
    std::vector<int32_t> wd(n, w_pad);
    const int64_t per = (int64_t)e->cfg.height * w_pad * 3;
    for (int i = 0; i < n; ++i) off[i] = per * i;
    if (pocr_stage_lines(e, batch_nhwc, off.data(), wd.data(), n, w_pad, 0)) return 1;
    return pocr_run_staged(e, logits_ntc, frame_argmax_nt, labels_nt, label_len_n);
}

int pocr_ctc_greedy(int device_id, const float *logits_ntc, int32_t n, int32_t T, int32_t C,
                    int32_t *frame_argmax_nt, int32_t *labels_nt, int32_t *label_len_n) {
    if (!logits_ntc || !labels_nt || !label_len_n) return fail("NULL pointer");
    if (n <= 0 || T <= 0 || C < 2) return fail("need n > 0, T > 0, C >= 2 (got %d, %d, %d)", n, T, C);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail("device_id %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    DevBuf lg, best, lab, len;
    const size_t nt = (size_t)n * T;
    int rc = lg.reserve(nt * C * sizeof(float)) || best.reserve(nt * sizeof(int32_t)) || lab.reserve(nt * sizeof(int32_t)) ||
             len.reserve((size_t)n * sizeof(int32_t));
    auto done = [&](int r) { lg.release(); best.release(); lab.release(); len.release(); return r; };
    if (rc) return done(1);
    if (locked_memcpy(lg.p, logits_ntc, nt * C * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return done(fail("H2D copy failed"));
    hipLaunchKernelGGL(frame_argmax_kernel, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, 0, lg.as<float>(), best.as<int32_t>(), (int)nt, C);
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(n), dim3(64), 0, 0, best.as<int32_t>(), lab.as<int32_t>(), len.as<int32_t>(), T, C - 1, nullptr, nullptr, T);
    if (hipGetLastError() != hipSuccess || locked_device_sync() != hipSuccess) return done(fail("CTC kernels failed"));
    if (frame_argmax_nt && locked_memcpy(frame_argmax_nt, best.p, nt * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return done(fail("D2H copy failed"));
    if (locked_memcpy(labels_nt, lab.p, nt * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return done(fail("D2H copy failed"));
    if (locked_memcpy(label_len_n, len.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return done(fail("D2H copy failed"));
    return done(0);
}

int pocr_sparsify(int device_id, const float *logits_ntc, int32_t n, int32_t T, int32_t C, float threshold,
                  float *data, int32_t *indices, int64_t capacity, int32_t *indptr, int64_t *line_off) {
    if (!logits_ntc || !data || !indices || !indptr || !line_off) return fail("NULL pointer");
    if (n <= 0 || T <= 0 || C < 1) return fail("need n > 0, T > 0, C >= 1 (got %d, %d, %d)", n, T, C);
    if (C > 256 * SP_COLS) return fail("sparse logits: C = %d exceeds %d classes", C, 256 * SP_COLS);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail("device_id %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    DevBuf lg, rowstat, colcount, nnz, off, ip, dd, di;
    const size_t cap = (size_t)n * T * C;
    int rc = lg.reserve(cap * sizeof(float)) || rowstat.reserve((size_t)n * T * 2 * sizeof(float)) || colcount.reserve((size_t)n * sp_blocks(T) * C * sizeof(int32_t)) ||
             nnz.reserve((size_t)n * sizeof(int32_t)) || off.reserve((size_t)(n + 1) * sizeof(int64_t)) ||
             ip.reserve((size_t)n * (C + 1) * sizeof(int32_t)) || dd.reserve(cap * sizeof(float)) || di.reserve(cap * sizeof(int32_t));
    auto done = [&](int r) { for (DevBuf *b : {&lg, &rowstat, &colcount, &nnz, &off, &ip, &dd, &di}) b->release(); return r; };
    if (rc) return done(1);
    if (locked_memcpy(lg.p, logits_ntc, cap * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return done(fail("H2D copy failed"));
    sparsify_launch(0, lg.as<float>(), nullptr, nullptr, rowstat.as<float>(), colcount.as<int32_t>(), nnz.as<int32_t>(), off.as<int64_t>(), ip.as<int32_t>(),
                    dd.as<float>(), di.as<int32_t>(), nullptr, nullptr, nullptr, n, T, T, C, threshold, -80.0f, (int64_t)cap, nullptr, nullptr);
    if (hipGetLastError() != hipSuccess || locked_device_sync() != hipSuccess) return done(fail("sparsify kernels failed"));
    if (locked_memcpy(line_off, off.p, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) return done(fail("D2H copy failed"));
    if (locked_memcpy(indptr, ip.p, (size_t)n * (C + 1) * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return done(fail("D2H copy failed"));
    const int64_t total = line_off[n];
    if (total > capacity) return done(fail("%lld entries kept, caller's buffers hold %lld", (long long)total, (long long)capacity));
    if (total > 0) {
        if (locked_memcpy(data, dd.p, (size_t)total * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return done(fail("D2H copy failed"));
        if (locked_memcpy(indices, di.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return done(fail("D2H copy failed"));
    }
    return done(0);
}

// ---- host-side helper of the transformer branch: find_best_overlap (line_ocr_engine.py:196-211) with the unit-cost
// edit distance of pero_ocr/sequence_alignment.py:4-13.  For i = 1 .. min(na, nb): cer_i = lev(a[na-i:], b[:i]) / i;
// returns the first i with the smallest cer below 1, or 0.  The ratios are compared exactly (cross-multiplied
// integers), which orders them like the reference's float divisions.  O(n^3) scalar work: ~10 ms for two
// 270-symbol parts (the reference's numpy/Python version needs seconds), no GPU involved.
int32_t pocr_best_overlap(const int32_t *a, int32_t na, const int32_t *b, int32_t nb) {
    if (!a || !b || na <= 0 || nb <= 0) return 0;
    const int n = std::min(na, nb);
    std::vector<int32_t> prev(n + 1), cur(n + 1);
    int64_t best_num = 1, best_den = 1;      // best cer so far = best_num / best_den (starts at 1)
    int32_t best = 0;
    for (int i = 1; i <= n; ++i) {
        const int32_t *x = a + (na - i);     // suffix of a, length i
        for (int q = 0; q <= i; ++q) prev[q] = q;
        for (int p = 1; p <= i; ++p) {
            cur[0] = p;
            const int32_t xp = x[p - 1];
            for (int q = 1; q <= i; ++q) {
                const int32_t sub = prev[q - 1] + (xp != b[q - 1]);
                const int32_t del = prev[q] + 1, ins = cur[q - 1] + 1;
                cur[q] = std::min(sub, std::min(del, ins));
            }
            std::swap(prev, cur);
        }
        const int64_t d = prev[i];
        if (d * best_den < best_num * i) { best_num = d; best_den = i; best = i; }     // d / i < best_num / best_den
    }
    return best;
}

int pocr_crop_lines(int device_id, const uint8_t *page_hwc, int32_t H, int32_t W, int32_t C, const float *coords,
                    const int64_t *coord_off, const int32_t *widths, int32_t n, int32_t line_height, uint8_t *crops,
                    const int64_t *crop_off) {
    if (!page_hwc || !coords || !coord_off || !widths || !crops || !crop_off) return fail("NULL pointer");
    if (H <= 0 || W <= 0 || C < 1 || C > 4 || line_height <= 0 || n <= 0) return fail("bad geometry (H %d, W %d, C %d, line height %d, n %d)", H, W, C, line_height, n);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail("device_id %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    std::vector<CropLine> tab(n);
    int64_t n_coord = 0, n_out = 0;
    int w_max = 0;
    for (int i = 0; i < n; ++i) {
        if (widths[i] < 0 || coord_off[i] < 0 || crop_off[i] < 0) return fail("line %d: negative width / offset", i);
        tab[i] = CropLine{coord_off[i], crop_off[i], widths[i], 0};
        n_coord = std::max<int64_t>(n_coord, coord_off[i] + (int64_t)line_height * widths[i] * 2);
        n_out = std::max<int64_t>(n_out, crop_off[i] + (int64_t)line_height * widths[i] * C);
        w_max = std::max(w_max, widths[i]);
    }
    if (w_max == 0) return 0;
    DevBuf dpage, dcoord, dtab, dout;
    const size_t page_bytes = (size_t)H * W * C;
    int rc = dpage.reserve(page_bytes) || dcoord.reserve((size_t)n_coord * sizeof(float)) || dtab.reserve((size_t)n * sizeof(CropLine)) ||
             dout.reserve((size_t)n_out);
    auto done = [&](int r) { dpage.release(); dcoord.release(); dtab.release(); dout.release(); return r; };
    if (rc) return done(1);
    if (locked_memcpy(dpage.p, page_hwc, page_bytes, hipMemcpyHostToDevice) != hipSuccess ||
        locked_memcpy(dcoord.p, coords, (size_t)n_coord * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        locked_memcpy(dtab.p, tab.data(), (size_t)n * sizeof(CropLine), hipMemcpyHostToDevice) != hipSuccess) return done(fail("H2D copy failed"));
    hipLaunchKernelGGL(remap_u8_kernel, dim3((line_height * w_max + 255) / 256, n), dim3(256), 0, 0, dpage.as<uint8_t>(), H, W, C,
                       dcoord.as<float>(), dtab.as<CropLine>(), line_height, dout.as<uint8_t>());
    if (hipGetLastError() != hipSuccess || locked_device_sync() != hipSuccess) return done(fail("remap kernel failed"));
    if (locked_memcpy(crops, dout.p, (size_t)n_out, hipMemcpyDeviceToHost) != hipSuccess) return done(fail("D2H copy failed"));
    return done(0);
}

int pocr_crop_curves(int device_id, const uint8_t *page_hwc, int32_t H, int32_t W, int32_t C, const double *curves,
                     const double *rows, const double *rot, const int32_t *widths, int32_t n, int32_t line_height,
                     uint8_t *crops, const int64_t *crop_off, float *grid_out) {
    if (!page_hwc || !curves || !rows || !rot || !widths || !crops || !crop_off) return fail("NULL pointer");
    if (H <= 0 || W <= 0 || C < 1 || C > 4 || line_height <= 0 || n <= 0) return fail("bad geometry (H %d, W %d, C %d, line height %d, n %d)", H, W, C, line_height, n);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail("device_id %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    std::vector<CurveLine> tab(n);
    int64_t n_curve = 0, n_out = 0, n_grid = 0;
    int w_max = 0;
    for (int i = 0; i < n; ++i) {
        if (widths[i] < 0 || crop_off[i] < 0) return fail("line %d: negative width / offset", i);
        tab[i] = CurveLine{n_curve, (int64_t)i * line_height, (int64_t)i * 4, crop_off[i], n_grid, widths[i], 0};
        n_curve += (int64_t)4 * widths[i];                       // curves are packed back to back in line order
        n_grid += (int64_t)2 * line_height * widths[i];
        n_out = std::max<int64_t>(n_out, crop_off[i] + (int64_t)line_height * widths[i] * C);
        w_max = std::max(w_max, widths[i]);
    }
    if (w_max == 0) return 0;
    DevBuf dpage, dcurve, drows, drot, dtab, dout, dgrid;
    const size_t page_bytes = (size_t)H * W * C;
    int rc = dpage.reserve(page_bytes) || dcurve.reserve((size_t)n_curve * sizeof(double)) || drows.reserve((size_t)n * line_height * sizeof(double)) ||
             drot.reserve((size_t)n * 4 * sizeof(double)) || dtab.reserve((size_t)n * sizeof(CurveLine)) || dout.reserve((size_t)n_out) ||
             (grid_out ? dgrid.reserve((size_t)n_grid * sizeof(float)) : 0);
    auto done = [&](int r) { for (DevBuf *b : {&dpage, &dcurve, &drows, &drot, &dtab, &dout, &dgrid}) b->release(); return r; };
    if (rc) return done(1);
    if (locked_memcpy(dpage.p, page_hwc, page_bytes, hipMemcpyHostToDevice) != hipSuccess ||
        locked_memcpy(dcurve.p, curves, (size_t)n_curve * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        locked_memcpy(drows.p, rows, (size_t)n * line_height * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        locked_memcpy(drot.p, rot, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        locked_memcpy(dtab.p, tab.data(), (size_t)n * sizeof(CurveLine), hipMemcpyHostToDevice) != hipSuccess) return done(fail("H2D copy failed"));
    hipLaunchKernelGGL(remap_curves_u8_kernel, dim3((line_height * w_max + 255) / 256, n), dim3(256), 0, 0, dpage.as<uint8_t>(), H, W, C,
                       dcurve.as<double>(), drows.as<double>(), drot.as<double>(), dtab.as<CurveLine>(), line_height, dout.as<uint8_t>(),
                       grid_out ? dgrid.as<float>() : (float *)nullptr);
    if (hipGetLastError() != hipSuccess || locked_device_sync() != hipSuccess) return done(fail("remap kernel failed"));
    if (locked_memcpy(crops, dout.p, (size_t)n_out, hipMemcpyDeviceToHost) != hipSuccess) return done(fail("D2H copy failed"));
    if (grid_out && locked_memcpy(grid_out, dgrid.p, (size_t)n_grid * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return done(fail("D2H copy failed"));
    return done(0);
}

int pocr_set_profiling(pocr_engine *e, int32_t enabled) {
    if (!e) return fail("engine is NULL");
    e->profiling = enabled != 0;
    for (Slot &s : e->slot) s.have_ms = false;
    return 0;
}

int pocr_last_stage_ms(pocr_engine *e, float *ms, int32_t cap) {
    if (!e || !ms) return 0;
    const Slot &s = e->slot[e->last_slot];
    if (!s.have_ms) return 0;
    int k = cap < POCR_NUM_STAGES ? cap : POCR_NUM_STAGES;
    for (int i = 0; i < k; ++i) ms[i] = s.stage_ms[i];
    return k;
}

int pocr_slot_stage_ms(pocr_engine *e, int32_t slot, float *ms, int32_t cap) {
    if (!e || !ms || slot < 0 || slot >= POCR_NUM_SLOTS) return 0;
    const Slot &s = e->slot[slot];
    if (!s.have_ms) return 0;
    int k = cap < POCR_NUM_STAGES ? cap : POCR_NUM_STAGES;
    for (int i = 0; i < k; ++i) ms[i] = s.stage_ms[i];
    return k;
}

int pocr_debug_read(pocr_engine *e, int32_t what, float *out, size_t cap, size_t *n_floats) {
    if (!e) return fail("engine is NULL");
    Slot &s = e->slot[e->last_slot];
    if (!s.staged) return fail("nothing has been run");
    HIP_TRY(hipSetDevice(e->device));
    const size_t rows = (size_t)s.rows;
    const float *src = nullptr;
    size_t sz = 0;
    if (what >= 0 && what < 9) { src = s.act[what].as<float>(); sz = (size_t)s.act_elems[what]; }
    else if (what == 9) { src = s.feat.as<float>(); sz = rows * e->cfg.conv_out; }
    else if (e->cfg.arch == POCR_ARCH_SA && what == 10) { src = nullptr; return fail("activation 10 (LayerNorm+PE) is not retained"); }
    else if ((e->cfg.arch == POCR_ARCH_SA || e->cfg.arch == POCR_ARCH_S2S) && what >= 11 && what < 11 + e->cfg.sa_layers) { src = s.sa_y[what - 11].as<float>(); sz = rows * e->cfg.conv_out; }
    else if (e->cfg.arch == POCR_ARCH_BLSTM && what >= 10 && what < 10 + e->cfg.lstm_layers) { src = s.lstm_y[what - 10].as<float>(); sz = rows * 2 * e->cfg.lstm_hidden; }
    else return fail("unknown activation id %d", what);
    if (n_floats) *n_floats = sz;
    const size_t k = cap < sz ? cap : sz;
    if (out && k) {
        HIP_TRY(hipStreamSynchronize(s.seq_stream));
        if (what == 0 && e->fuse12) {       // conv1's activation does not exist in the fused mode: compute it now, from the crops still staged
            HIP_TRY(hipStreamSynchronize(s.stream));
            if (launch_pad_fill(e, s, s.stream, 0, -1) || launch_conv1(e, s, s.stream)) return 1;
        }
        HIP_TRY(hipMemcpyAsync(out, src, k * sizeof(float), hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipStreamSynchronize(s.stream));
        const bool seq_p2 = (what == 9 && s.feat_is_p2) || (e->cfg.arch == POCR_ARCH_BLSTM && what >= 10 && what - 10 < s.lstm_p2_layers);
        if ((e->p2 && what >= 0 && what < 9) || seq_p2) {
            // conv activations are kept pre-split (conv_bf16x3.hpp "P2": per pixel and 32-channel chunk 32 x f16 h, 32 x f16 l);
            // the caller gets the values they stand for, x = h + l / 2048, in NHWC order
            const int C = what < 9 ? s.act_c[what] : what == 9 ? e->cfg.conv_out : 2 * e->cfg.lstm_hidden;
            std::vector<float> px(C);
            for (size_t p0 = 0; p0 + C <= k; p0 += C) {
                const _Float16 *raw = reinterpret_cast<const _Float16 *>(out + p0);
                for (int c = 0; c < C; ++c) {
                    const _Float16 h = raw[(c >> 5) * 64 + (c & 31)], l = raw[(c >> 5) * 64 + 32 + (c & 31)];
                    px[c] = (float)h + (float)l * (1.0f / 2048.0f);
                }
                memcpy(out + p0, px.data(), (size_t)C * sizeof(float));
            }
        }
    }
    return 0;
}

}  // extern "C"

#include "parsenet_host.hpp"
#include "crop_host.hpp"
