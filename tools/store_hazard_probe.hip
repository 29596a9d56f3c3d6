// store_hazard_probe.hip - does gfx950 need a wait state between `buffer_store_dwordx4 ... sN offen` (SGPR offset) and a VALU write
// of the store's data registers?
//
// Why: the library variant whose staged conv epilogue masks out-of-image lanes by an out-of-range offset instead of a branch
// (tools/masked_store_repro.sh) gives wrong DWORDS in its own output under load (tools/three_in_flight.py: runs of two f16 channels,
// sometimes in the high plane).  Its ISA differs from the shipped (branching) one in one respect that matters: the scheduler put
//       buffer_store_dwordx4 v[2:5], v22, s[4:7], s11 offen
//       v_lshrrev_b32_e32 v2, 5, v1                              <- VALU write of data register 0, zero wait states
// back to back, where the branching form has `s_or_b64 exec` and address arithmetic on OTHER registers behind the store.  The gfx9
// rule (a store of more than 64 bits must be followed by one wait state before a VALU write of its data registers) is applied by
// LLVM's hazard recogniser only when the store has NO SGPR offset.  This probe issues exactly that pair from inline asm - with /
// without an SGPR offset, with 0 / 1 / 2 wait states, all lanes in range / a quarter of them out of range - on three streams that
// keep the memory pipeline backed up, and counts stored units whose dword 0 is the value of the LATER v_mov.
//
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/store_hazard_probe tools/store_hazard_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kGarbage = 0xDEADBEEFu;

// VARIANT bits: 1 = SGPR offset in the store, 2 = one wait state (s_nop 0) behind it, 4 = two (s_nop 1), 8 = a quarter of the lanes out of range
#define HAZARD_ASM(SOFF_TXT, NOP_TXT)                                                                                   \
    asm volatile("v_mov_b32 v4, %0\n\tv_mov_b32 v5, %1\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\t"                     \
                 "s_nop 4\n\t"                                                                                         \
                 "buffer_store_dwordx4 v[4:7], %4, %5, " SOFF_TXT " offen\n\t" NOP_TXT                                  \
                 "v_mov_b32 v4, %7\n\t"                                                                                \
                 "s_nop 4\n\t"                                                                                         \
                 :                                                                                                     \
                 : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(voff), "s"(rsv), "s"(soff_u), "v"(kGarbage)                 \
                 : "v4", "v5", "v6", "v7", "memory")

// the same pair with a GLOBAL store (FLAT encoding: the case LLVM always pads - two wait states on gfx940+): SGPR base + VGPR offset
#define HAZARD_ASM_GLOBAL(NOP_TXT)                                                                                     \
    asm volatile("v_mov_b32 v4, %0\n\tv_mov_b32 v5, %1\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\t"                     \
                 "s_nop 4\n\t"                                                                                         \
                 "global_store_dwordx4 %4, v[4:7], %5\n\t" NOP_TXT                                                     \
                 "v_mov_b32 v4, %6\n\t"                                                                                \
                 "s_nop 4\n\t"                                                                                         \
                 :                                                                                                     \
                 : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(goff), "s"(gbase), "v"(kGarbage)                            \
                 : "v4", "v5", "v6", "v7", "memory")

template <int VARIANT>
__global__ __launch_bounds__(256) void hazard_kernel(unsigned *buf, unsigned bytes_per_wg, int iters, unsigned id) {
    // the workgroup's region as a raw buffer; iteration `it` stores 256 consecutive 16-byte units (a wave instruction = eight 128-byte lines)
    char *base = reinterpret_cast<char *>(buf) + (size_t)blockIdx.x * bytes_per_wg;
    const unsigned long long ba = reinterpret_cast<unsigned long long>(base);      // raw buffer descriptor: base, stride 0, num_records, DATA_FORMAT 32
    const u32x4 rsv = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(ba >> 32) & 0xffffu)),
                       (unsigned)__builtin_amdgcn_readfirstlane((int)bytes_per_wg), 0x00020000u};
    for (int it = 0; it < iters; ++it) {
        const unsigned unit = (unsigned)it * 256u + threadIdx.x;
        const unsigned off = unit * 16u;
        const bool oor = (VARIANT & 8) && (threadIdx.x & 3) == 3;
        // SGPR-offset variants: the iteration's position goes through the scalar offset (as the tile position does in the epilogue)
        const unsigned soff_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((VARIANT & 1) ? (unsigned)it * 4096u : 0u));
        const unsigned voff = oor ? 0x80000000u : ((VARIANT & 1) ? threadIdx.x * 16u : off);
        const unsigned d0 = unit * 0x9E3779B9u + id, d1 = unit ^ 0x5555AAAAu, d2 = unit, d3 = id;
        if constexpr ((VARIANT & 16) != 0) {
            const unsigned goff = off;                                  // (global store: no range check, every lane in range)
            const unsigned long long gbase = ba;
            if constexpr ((VARIANT & 6) == 0) HAZARD_ASM_GLOBAL("");
            else if constexpr ((VARIANT & 6) == 2) HAZARD_ASM_GLOBAL("s_nop 0\n\t");
            else HAZARD_ASM_GLOBAL("s_nop 1\n\t");
        } else
        if constexpr ((VARIANT & 7) == 1) HAZARD_ASM("%6", "");
        else if constexpr ((VARIANT & 7) == 3) HAZARD_ASM("%6", "s_nop 0\n\t");
        else if constexpr ((VARIANT & 7) == 5) HAZARD_ASM("%6", "s_nop 1\n\t");
        else if constexpr ((VARIANT & 7) == 0) HAZARD_ASM("0", "");
        else if constexpr ((VARIANT & 7) == 2) HAZARD_ASM("0", "s_nop 0\n\t");
        else HAZARD_ASM("0", "s_nop 1\n\t");
    }
}

// dword 0 of every stored unit: the pattern, the later v_mov's value (the hazard), or something else; out-of-range lanes' units keep the fill
__global__ void check_kernel(const u32x4 *buf, size_t units_per_wg, int n_wg, unsigned id, int oor_quarter, unsigned long long *cnt) {
    const size_t total = units_per_wg * n_wg;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const unsigned unit = (unsigned)(i % units_per_wg);
        const u32x4 v = buf[i];
        const bool masked = oor_quarter && (unit & 3) == 3;
        const unsigned d0 = unit * 0x9E3779B9u + id;
        if (masked) { if (v[0] != 0x11111111u || v[1] != 0x11111111u) atomicAdd(cnt + 3, 1ull); continue; }
        if (v[0] == d0 && v[1] == (unit ^ 0x5555AAAAu) && v[2] == unit && v[3] == id) atomicAdd(cnt + 0, 1ull);
        else if (v[0] == kGarbage && v[1] == (unit ^ 0x5555AAAAu) && v[2] == unit && v[3] == id) atomicAdd(cnt + 1, 1ull);
        else atomicAdd(cnt + 2, 1ull);
    }
}
__global__ void fill_kernel(u32x4 *p, size_t n16, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = (u32x4){v, v, v, v};
}

template <int VARIANT>
static void run_variant(const char *name, int rounds) {
    const int n_wg = 2048, iters = 64, n_streams = 3;
    const unsigned bytes_per_wg = (unsigned)iters * 4096u;
    const size_t units_per_wg = (size_t)iters * 256, bytes = (size_t)n_wg * bytes_per_wg;
    hipStream_t st[n_streams];
    unsigned *buf[n_streams];
    unsigned long long *cnt;
    CHECK(hipMalloc(&cnt, 4 * sizeof(unsigned long long)));
    for (int s = 0; s < n_streams; ++s) { CHECK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking)); CHECK(hipMalloc(&buf[s], bytes)); }
    unsigned long long tot[4] = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        for (int s = 0; s < n_streams; ++s) hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, st[s], reinterpret_cast<u32x4 *>(buf[s]), bytes / 16, 0x11111111u);
        CHECK(hipDeviceSynchronize());
        for (int s = 0; s < n_streams; ++s)
            hipLaunchKernelGGL((hazard_kernel<VARIANT>), dim3(n_wg), dim3(256), 0, st[s], buf[s], bytes_per_wg, iters, (unsigned)(r * 8 + s + 1));
        CHECK(hipDeviceSynchronize());
        for (int s = 0; s < n_streams; ++s) {
            CHECK(hipMemsetAsync(cnt, 0, 4 * sizeof(unsigned long long), st[0]));      // (same stream as the check: the streams are non-blocking)
            hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, st[0], reinterpret_cast<const u32x4 *>(buf[s]), units_per_wg, n_wg, (unsigned)(r * 8 + s + 1), (VARIANT & 8) ? 1 : 0, cnt);
            CHECK(hipDeviceSynchronize());
            unsigned long long got[4];
            CHECK(hipMemcpy(got, cnt, sizeof(got), hipMemcpyDeviceToHost));
            for (int k = 0; k < 4; ++k) tot[k] += got[k];
        }
    }
    const unsigned long long expect = (unsigned long long)rounds * n_streams * n_wg * units_per_wg;
    const unsigned long long seen = tot[0] + tot[1] + tot[2], masked = (VARIANT & 8) ? expect / 4 : 0;
    printf("%-78s units right %llu, dword 0 = the LATER v_mov's value %llu (%.2f %%), otherwise wrong %llu, masked units written %llu%s\n", name, tot[0], tot[1],
           100.0 * (double)tot[1] / (double)(expect - masked), tot[2], tot[3], seen + masked == expect ? "" : "  [COUNT MISMATCH]");
    for (int s = 0; s < n_streams; ++s) { CHECK(hipFree(buf[s])); CHECK(hipStreamDestroy(st[s])); }
    CHECK(hipFree(cnt));
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 4;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("store_hazard_probe on %s, %d rounds x 3 streams x 2048 workgroups x 64 stores per lane\n", prop.gcnArchName, rounds);
    run_variant<1>("SGPR offset, v_mov of data register 0 directly behind the store", rounds);
    run_variant<3>("SGPR offset, one wait state (s_nop 0) between them", rounds);
    run_variant<5>("SGPR offset, two wait states (s_nop 1)", rounds);
    run_variant<9>("SGPR offset, directly behind, a quarter of the lanes out of range", rounds);
    run_variant<11>("SGPR offset, one wait state, a quarter of the lanes out of range", rounds);
    run_variant<13>("SGPR offset, two wait states, a quarter of the lanes out of range", rounds);
    run_variant<0>("no SGPR offset (the case LLVM pads), directly behind", rounds);
    run_variant<2>("no SGPR offset, one wait state", rounds);
    run_variant<8>("no SGPR offset, directly behind, a quarter of the lanes out of range", rounds);
    run_variant<10>("no SGPR offset, one wait state, a quarter of the lanes out of range", rounds);
    run_variant<4>("no SGPR offset, two wait states (what LLVM emits on gfx940+)", rounds);
    run_variant<16>("global_store_dwordx4 (SGPR base), v_mov directly behind", rounds);
    run_variant<18>("global_store_dwordx4 (SGPR base), one wait state", rounds);
    run_variant<20>("global_store_dwordx4 (SGPR base), two wait states (what LLVM emits)", rounds);
    return 0;
}
