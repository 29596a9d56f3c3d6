// masked_store_probe.hip - are raw-buffer STORES that are masked by an out-of-range vector offset harmless on gfx950?
//
// Background (csrc/conv_bf16x3.hpp conv_epilogue_staged): the staged epilogue once masked the lanes of a
// tile that lie outside the image by giving them an out-of-range buffer offset (mark 0x80000000 or 0x7FFF0000 in the VGPR offset,
// the tile's position in the SGPR offset) instead of branching around the store.  With three launches in flight on FRESH engines a
// few low planes in the activations of a launch running at the same time came out wrong.  Loads masked the same way are fine.
//
// Part 1 (fully mapped arena): three streams store tiles into their own images of one canary-filled arena, out-of-image lanes
//   masked by the mark; the arena also covers base + mark + tile offset of every image.  Checks: every in-image unit holds its
//   pattern, every other byte of the arena still holds the canary - i.e. the hardware drops the masked stores.
//   Variants: mark 0x80000000 / 0x7FFF0000 / num_records, scalar offset zero / non-zero; control: a descriptor whose range covers the mark.
// Part 2 (the address a dropped store points at is NOT mapped, and gets mapped while the stores run): buffer A is written with
//   masked stores whose dropped lanes point into the virtual range a later hipMalloc returns (found by allocating and freeing it
//   once); while that kernel runs, the host allocates that range (buffer B) and another stream fills and verifies B.
//   Control: the same kernel with a branch around the store.
//
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/masked_store_probe tools/masked_store_probe.hip
// run:   tools/bin/masked_store_probe [rounds=8]
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kCanary = 0xC5C5C5C5u;

__device__ __forceinline__ u32x4 pattern(unsigned image, unsigned byte_off) {
    const unsigned a = image * 0x9E3779B9u + byte_off * 0x85EBCA6Bu + 0x1234567u;
    return (u32x4){a, a ^ 0xA5A5A5A5u, byte_off, image | 0x80000000u};
}

struct TileArgs {
    char *base;             // the image (line) this launch writes: base of the buffer descriptor
    unsigned num_records;   // bytes of the image (descriptor range)
    int Ho, Wout;           // image rows / columns (pixels)
    unsigned pix_bytes;     // bytes per pixel (multiple of 16)
    unsigned mark;          // out-of-range mark of a masked lane (VGPR offset)
    int mode;               // 0: masked by the mark, 1: branch around the store
    int use_soffset;        // 1: tile position in the scalar offset (as the epilogue does); 0: everything in the vector offset
    unsigned image_id;
    int reps;               // a workgroup stores its tile this many times (keeps the kernel running)
};

// the store loop of conv_epilogue_staged: a TH x TW tile of pixels, 16-byte units, 256 threads, lanes outside the image masked
template <int TH, int TW>
__global__ __launch_bounds__(256) void tile_store_kernel(TileArgs a) {
    const int tiles_w = (a.Wout + TW - 1) / TW;
    const int h0 = (blockIdx.x / tiles_w) * TH, w0 = (blockIdx.x % tiles_w) * TW;
    const int upp = (int)(a.pix_bytes / 16u), units = TH * TW * upp;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.base, 0, (int)a.num_records, 0x00020000);
    const unsigned tile_off = __builtin_amdgcn_readfirstlane((int)(((unsigned)h0 * (unsigned)a.Wout + (unsigned)w0) * a.pix_bytes));
    const int rows_ok = a.Ho - h0, cols_ok = a.Wout - w0;
    for (int rep = 0; rep < a.reps; ++rep)
        for (int g = threadIdx.x; g < units; g += 256) {
            const int sp = g / upp, unit = g % upp, row = sp / TW, col = sp % TW;
            const bool ok = row < rows_ok && col < cols_ok;
            const unsigned in_tile = (unsigned)(row * a.Wout + col) * a.pix_bytes + (unsigned)unit * 16u;
            const u32x4 val = pattern(a.image_id, tile_off + in_tile);
            const unsigned voff = a.use_soffset ? in_tile : tile_off + in_tile;
            const unsigned soff = a.use_soffset ? tile_off : 0u;
            if (a.mode == 0) __builtin_amdgcn_raw_buffer_store_b128(val, rs, (int)(ok ? voff : a.mark), (int)soff, 0);
            else if (ok) __builtin_amdgcn_raw_buffer_store_b128(val, rs, (int)voff, (int)soff, 0);
        }
}

// masked LOADS (the halo's zero padding): lanes outside the image must read zeros
template <int TH, int TW>
__global__ __launch_bounds__(256) void tile_load_kernel(TileArgs a, unsigned long long *bad) {
    const int tiles_w = (a.Wout + TW - 1) / TW;
    const int h0 = (blockIdx.x / tiles_w) * TH, w0 = (blockIdx.x % tiles_w) * TW;
    const int upp = (int)(a.pix_bytes / 16u), units = TH * TW * upp;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.base, 0, (int)a.num_records, 0x00020000);
    const unsigned tile_off = __builtin_amdgcn_readfirstlane((int)(((unsigned)h0 * (unsigned)a.Wout + (unsigned)w0) * a.pix_bytes));
    const int rows_ok = a.Ho - h0, cols_ok = a.Wout - w0;
    unsigned long long wrong = 0;
    for (int g = threadIdx.x; g < units; g += 256) {
        const int sp = g / upp, unit = g % upp, row = sp / TW, col = sp % TW;
        const bool ok = row < rows_ok && col < cols_ok;
        const unsigned in_tile = (unsigned)(row * a.Wout + col) * a.pix_bytes + (unsigned)unit * 16u;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? in_tile : a.mark), (int)tile_off, 0);
        const u32x4 want = ok ? pattern(a.image_id, tile_off + in_tile) : (u32x4){0u, 0u, 0u, 0u};
        wrong += (v[0] != want[0]) + (v[1] != want[1]) + (v[2] != want[2]) + (v[3] != want[3]);
    }
    if (wrong) atomicAdd(bad, wrong);
}

__global__ void fill_kernel(u32x4 *p, size_t n16, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = (u32x4){v, v, v, v};
}

// arena check: unit u of the arena belongs to image k if it lies inside [off_k, off_k + bytes_k); else canary.
struct ImageDesc { size_t off; unsigned bytes; unsigned id; };
__global__ void check_kernel(const u32x4 *arena, size_t n16, const ImageDesc *imgs, int n_img, unsigned long long *bad_pattern,
                             unsigned long long *bad_canary, unsigned long long *first_bad) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const size_t byte = i * 16;
        u32x4 want = (u32x4){kCanary, kCanary, kCanary, kCanary};
        bool in_img = false;
        for (int k = 0; k < n_img; ++k)
            if (byte >= imgs[k].off && byte < imgs[k].off + imgs[k].bytes) { want = pattern(imgs[k].id, (unsigned)(byte - imgs[k].off)); in_img = true; }
        const u32x4 v = arena[i];
        if (v[0] != want[0] || v[1] != want[1] || v[2] != want[2] || v[3] != want[3]) {
            atomicAdd(in_img ? bad_pattern : bad_canary, 1ull);
            atomicMin(first_bad, (unsigned long long)byte);
        }
    }
}

// part 2 victim: fill B with a pattern, then verify it (separate launches on the victim's stream)
__global__ void victim_fill(u32x4 *p, size_t n16, unsigned id) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = pattern(id, (unsigned)(i * 16));
}
__global__ void victim_check(const u32x4 *p, size_t n16, unsigned id, unsigned long long *bad, unsigned long long *first_bad) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const u32x4 want = pattern(id, (unsigned)(i * 16)), v = p[i];
        if (v[0] != want[0] || v[1] != want[1] || v[2] != want[2] || v[3] != want[3]) { atomicAdd(bad, 1ull); atomicMin(first_bad, (unsigned long long)(i * 16)); }
    }
}

static unsigned long long *dev_counters(int n) {
    unsigned long long *p;
    CHECK(hipMalloc(&p, n * sizeof(unsigned long long)));
    return p;
}

// ---- part 1
static int part1(int rounds) {
    constexpr int TH = 5, TW = 16;
    const unsigned pix_bytes = 512;                         // 128 channels in the P2 layout
    const int Ho = 23, Wout = 181;                          // neither a multiple of the tile: masked rows and columns
    const unsigned img_bytes = (unsigned)Ho * Wout * pix_bytes;      // 2.1 MB
    struct Variant { const char *name; unsigned mark; int use_soffset; int big_records; };
    const Variant variants[] = {
        {"mark 0x80000000, tile in the scalar offset", 0x80000000u, 1, 0},
        {"mark 0x7FFF0000, tile in the scalar offset", 0x7FFF0000u, 1, 0},
        {"mark = num_records, tile in the scalar offset", 0u, 1, 0},
        {"mark 0x80000000, no scalar offset", 0x80000000u, 0, 0},
        {"CONTROL: mark 0x7FFF0000 with num_records 0xFFFFFFFF (the mark is IN range)", 0x7FFF0000u, 1, 1},
    };
    // arena: images of stream s at s * 64 MiB, their "shadows" (where base + mark points) at + mark: 3 x 64 MiB + 2 GiB + slack
    const size_t arena_bytes = ((size_t)3 << 26) + ((size_t)1 << 31) + ((size_t)1 << 27);
    char *arena;
    CHECK(hipMalloc(&arena, arena_bytes));
    hipStream_t st[3];
    for (auto &s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *cnt = dev_counters(4);
    ImageDesc *d_imgs;
    CHECK(hipMalloc(&d_imgs, 8 * sizeof(ImageDesc)));
    int failures = 0;
    for (const Variant &v : variants) {
        unsigned long long tot_pat = 0, tot_can = 0, tot_ld = 0, first = ~0ull;
        for (int r = 0; r < rounds; ++r) {
            hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st[0], reinterpret_cast<u32x4 *>(arena), arena_bytes / 16, kCanary);
            CHECK(hipDeviceSynchronize());
            ImageDesc imgs[3];
            const int tiles = ((Ho + TH - 1) / TH) * ((Wout + TW - 1) / TW);
            for (int s = 0; s < 3; ++s) {
                TileArgs a{};
                // image of stream s (the arena reaches beyond base + mark + image: a store that is not dropped lands on canary bytes)
                const size_t off = (size_t)s << 26;
                a.base = arena + off; a.num_records = v.big_records ? 0xFFFFFFFFu : img_bytes;
                a.Ho = Ho; a.Wout = Wout; a.pix_bytes = pix_bytes;
                a.mark = v.mark ? v.mark : a.num_records;
                a.mode = 0; a.use_soffset = v.use_soffset; a.image_id = (unsigned)(r * 16 + s + 1); a.reps = 40;
                imgs[s] = ImageDesc{off, img_bytes, a.image_id};
                hipLaunchKernelGGL((tile_store_kernel<TH, TW>), dim3(tiles), dim3(256), 0, st[s], a);
            }
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(d_imgs, imgs, sizeof(imgs), hipMemcpyHostToDevice));
            unsigned long long init[4] = {0, 0, ~0ull, 0};
            CHECK(hipMemcpy(cnt, init, sizeof(init), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, st[0], reinterpret_cast<const u32x4 *>(arena), arena_bytes / 16, d_imgs, 3, cnt, cnt + 1, cnt + 2);
            // masked loads of the same images
            for (int s = 0; s < 3; ++s) {
                TileArgs a{};
                a.base = arena + imgs[s].off; a.num_records = v.big_records ? 0xFFFFFFFFu : img_bytes; a.Ho = Ho; a.Wout = Wout; a.pix_bytes = pix_bytes;
                a.mark = v.mark ? v.mark : a.num_records; a.image_id = imgs[s].id;
                if (!v.big_records)      // (with the wide range a masked load really reads what lies at base + mark: not zeros by construction)
                    hipLaunchKernelGGL((tile_load_kernel<TH, TW>), dim3(tiles), dim3(256), 0, st[0], a, cnt + 3);
            }
            CHECK(hipDeviceSynchronize());
            unsigned long long got[4];
            CHECK(hipMemcpy(got, cnt, sizeof(got), hipMemcpyDeviceToHost));
            tot_pat += got[0]; tot_can += got[1]; tot_ld += got[3];
            if (got[2] < first) first = got[2];
        }
        const bool expect_bad = v.big_records;     // the control: a descriptor range that covers the mark must NOT drop the stores
        const bool ok = expect_bad ? (tot_can > 0) : (tot_pat == 0 && tot_can == 0 && tot_ld == 0);
        printf("part 1  %-72s rounds %d: in-image units wrong %llu, canary units overwritten %llu%s, masked-load words wrong %llu  -> %s\n", v.name, rounds,
               tot_pat, tot_can, tot_can ? (std::string(" (first at arena byte ") + std::to_string(first) + ")").c_str() : "", tot_ld,
               expect_bad ? (ok ? "control ok (stores inside the descriptor's range do land)" : "CONTROL FAILED") : (ok ? "masked stores dropped" : "MASKED STORES NOT HARMLESS"));
        if (!ok) ++failures;
    }
    CHECK(hipFree(arena)); CHECK(hipFree(cnt)); CHECK(hipFree(d_imgs));
    for (auto &s : st) CHECK(hipStreamDestroy(s));
    return failures;
}

// ---- part 2
static int part2(int rounds) {
    constexpr int TH = 5, TW = 16;
    const unsigned pix_bytes = 512;
    const int Ho = 23, Wout = 181;
    const unsigned img_bytes = (unsigned)Ho * Wout * pix_bytes;
    const size_t a_bytes = (size_t)64 << 20, b_bytes = (size_t)512 << 20;
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned long long *cnt = dev_counters(2);
    int failures = 0;
    for (int mode = 0; mode < 2; ++mode) {
        unsigned long long tot_bad = 0, first = ~0ull;
        int usable = 0, same_va = 0;
        for (int r = 0; r < rounds; ++r) {
            char *A = nullptr, *B = nullptr;
            CHECK(hipMalloc(&B, b_bytes));                 // where a buffer of this size lands (the allocator hands out descending addresses:
            CHECK(hipMalloc(&A, a_bytes));                 // A, allocated second, lies BELOW it - reachable by a positive 32-bit offset) ...
            const char *b_expected = B;
            CHECK(hipFree(B));                             // ... and unmapped again
            const long long delta = b_expected - A;
            if (delta <= (long long)a_bytes || delta + (long long)img_bytes >= (1ll << 32)) {
                if (r == 0) printf("part 2  (B lands at A %+lld bytes: not reachable by a 32-bit offset - round skipped)\n", delta);
                CHECK(hipFree(A));
                continue;
            }
            ++usable;
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, s1, reinterpret_cast<u32x4 *>(A), a_bytes / 16, kCanary);
            CHECK(hipDeviceSynchronize());
            // the attacker: masked stores whose dropped lanes point at b_expected + (0 .. img_bytes): unmapped right now
            TileArgs a{};
            a.base = A; a.num_records = img_bytes; a.Ho = Ho; a.Wout = Wout; a.pix_bytes = pix_bytes;
            a.mark = (unsigned)delta; a.mode = mode; a.use_soffset = 1; a.image_id = 7; a.reps = 6000;      // ~tens of ms
            const int tiles = ((Ho + TH - 1) / TH) * ((Wout + TW - 1) / TW);
            hipLaunchKernelGGL((tile_store_kernel<TH, TW>), dim3(tiles), dim3(256), 0, s1, a);
            // the victim: the allocation that maps that range, filled and verified by another stream while the attacker runs
            CHECK(hipMalloc(&B, b_bytes));
            if (B == b_expected) ++same_va;
            unsigned long long init[2] = {0, ~0ull};
            CHECK(hipMemcpyAsync(cnt, init, sizeof(init), hipMemcpyHostToDevice, s2));
            for (int k = 0; k < 3; ++k) {
                hipLaunchKernelGGL(victim_fill, dim3(2048), dim3(256), 0, s2, reinterpret_cast<u32x4 *>(B), b_bytes / 16, (unsigned)(100 + k));
                hipLaunchKernelGGL(victim_check, dim3(2048), dim3(256), 0, s2, reinterpret_cast<const u32x4 *>(B), b_bytes / 16, (unsigned)(100 + k), cnt, cnt + 1);
            }
            const bool attacker_running = hipStreamQuery(s1) == hipErrorNotReady;
            CHECK(hipDeviceSynchronize());
            unsigned long long got[2];
            CHECK(hipMemcpy(got, cnt, sizeof(got), hipMemcpyDeviceToHost));
            tot_bad += got[0];
            if (got[1] < first) first = got[1];
            if (r == 0) printf("part 2  (%s: A %p, B %p = A + 0x%llx, attacker still running when the victim was enqueued: %s)\n", mode ? "branch" : "mask", (void *)A, (void *)B,
                               (unsigned long long)delta, attacker_running ? "yes" : "no");
            CHECK(hipFree(A)); CHECK(hipFree(B));
        }
        printf("part 2  %-8s stores next to an allocation that maps the range their dropped lanes point at: %d usable rounds (%d re-used the address), victim units wrong %llu%s\n",
               mode ? "branched" : "masked", usable, same_va, tot_bad, tot_bad ? (std::string(" (first at byte ") + std::to_string(first) + ")").c_str() : "");
        if (mode == 0 && tot_bad) ++failures;
    }
    CHECK(hipFree(cnt));
    CHECK(hipStreamDestroy(s1)); CHECK(hipStreamDestroy(s2));
    return failures;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("masked_store_probe on %s (%s), %d rounds per variant\n", prop.name, prop.gcnArchName, rounds);
    const int f1 = part1(rounds);
    const int f2 = part2(rounds);
    printf("summary: part 1 %s, part 2 %s\n", f1 ? "FAILED" : "clean", f2 ? "victim corrupted with masked stores" : "clean");
    return (f1 || f2) ? 1 : 0;
}
