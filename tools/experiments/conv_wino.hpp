// conv_wino.hpp — the deep-K 3x3 layers of the backbone as a width-wise Winograd F(2,3) on the f16x2 matrix-pipe arithmetic.
//
// Replaces aten::conv2d + ReLU / LeakyReLU + max_pool2d + batch_norm of the reference's VGG blocks
// (pero_ocr/ocr_engine/transformer.py:51-72, 86-144) for the layers where the product blocks dominate: the part runs at its
// power set point (profiles/r04_power_cap.txt), so the time of an MFMA-bound layer is the NUMBER of matrix operations, and
// F(2,3) along the width needs 4 instead of 6 products per two output columns and (tap row, input channel):
//     d = x[2p-1 .. 2p+2] (one image row, one channel), g = W[dy][0..2]
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3            (input transform, this kernel, fp32)
//     U0 = g0        U1 = (g0 + g1 + g2) / 2       U2 = (g0 - g1 + g2) / 2       U3 = g2      (weights, host, float64)
//     M_k[r][p][co] = sum_{dy, ci} V_k[r + dy][p][ci] * U_k[dy][ci][co]     (four GEMMs on the matrix pipe, f16x2)
//     y[2p] = M0 + M1 + M2        y[2p+1] = M1 - M2 - M3                    (output transform, epilogue, fp32)
// i.e. 12 instead of 18 "taps" per pair of output columns.  Numerics: oracle/winograd_numerics.py (the transformed operands are
// rounded to f16x2's 22 bits like every other operand of the stack; on c3's worst chunk the logits stay 1.1e-4 from float64).
//
// Work split.  A workgroup of EIGHT waves owns TH rows x 16 column pairs (= 32 output columns) x 64 output channels; wave
// w = (k = w & 3, nh = w >> 2) accumulates plane k for the channel half nh (TH x 2 accumulator tiles, main + cross terms: the
// register budget of the direct kernel's wave).  The four planes of one halo row are a lane-for-lane image of the MFMA A
// operand (lane = (pair, channel octet)), so the transform needs no shuffles: per 32-channel chunk, wave j < TH + 2 loads the
// four pixels 2p-1 .. 2p+2 of halo row j for its lane's (pair, octet) straight from the P2 activations (no LDS hop), rebuilds
// the fp32 values, forms V0..V3, splits them again and writes the eight 1 KB fragments (plane, h / l) into the LDS buffer of
// the NEXT chunk while the current one multiplies; one barrier per chunk.  Main loop = the halo-row streaming of
// conv_bf16x3.hpp: the fragment of halo row j feeds output rows j, j-1, j-2 (dy = 0, 1, 2); weights straight from L2 into
// registers, the next chunk's set requested while this one computes.  The output transform goes through LDS once per tile
// (all accumulators -> [plane][row][pair][channel] fp32 -> thread = (row, pair, channel octet)): bias, activation, BatchNorm,
// pooling (the Winograd pair IS the width-pool pair), range note, f16x2 split and 16-byte P2 stores.
//
// A tile's two halves (8 pairs each) are described separately (WinoTile): a line whose width is an odd number of half tiles
// (c2: 144 columns = 4.5 tiles) shares its last tile with another line instead of multiplying zeros.
// STATUS (round 5): an EXPERIMENT, not part of libpocr_hip.so.  Correct (tools/conv_wino_bench.hip: as close to float64 as the direct
// kernel, every layer), but it only TIES the direct kernel on conv9 (2.0 ms both) and loses on the shallower layers: what bounds
// it is not the matrix pipe (MFMA busy 0.43 against the direct kernel's 0.81) but the round trip of the activation loads the
// transform needs one chunk ahead (profiles/r05_winograd.txt: per workgroup 31 us without them, 43.5 with, whatever their
// volume, pattern or L2 residency) - with the register file full (accumulators for four planes + two weight sets) there is no room
// for a second staging set, and s_waitcnt vmcnt counts in order.  DESIGN.md section 9 has the numbers and what was tried.
#pragma once
#include "../../pero_ocr_amd/csrc/conv_bf16x3.hpp"

namespace pocr {

// Pixel tile: two HALF tiles of 8 output-column pairs (16 columns) each, which may belong to different lines; line = -1: no such
// half; ht_wt = (first row h0 << 16) | half-tile index (columns 16 wt .. 16 wt + 15)
struct WinoTile { int32_t line[2]; int32_t ht_wt[2]; };
struct WinoArgs : ConvArgs {
    const WinoTile *wtiles;  // n_ptiles of them; line_w / in_off / out_off as for ConvArgs::tiles
    uint32_t x_bytes;        // size of the input tensor (its loads are bounds-checked buffer loads; < 0xF0000000)
};
inline size_t wino_grid_blocks(const WinoArgs &a) {    // (conv_grid_blocks with the Winograd tile count)
    const size_t P = (size_t)a.n_ptiles;
    const int tn = a.tiles_n;
    if (tn >= 2 && tn <= 8 && (tn & (tn - 1)) == 0) {
        const int G = a.xcd_g > 1 && tn % a.xcd_g == 0 ? a.xcd_g : 1;
        const size_t groups = 8 * G / tn;
        return (P + groups - 1) / groups * G * 8;
    }
    return P * tn;
}

#ifndef POCR_TRACE_STAMP                 // (the library's phase stamps were removed in round 6; the experiment's two call sites stay no-ops)
#define POCR_TRACE_STAMP(k) do { } while (0)
#endif
#ifndef POCR_WINO_XF_A
#define POCR_WINO_XF_A 1               // unit (halo row) of the chunk's MFMA stream behind which the next chunk's transform runs: waves 0..3
#endif
#ifndef POCR_WINO_XF_B
#define POCR_WINO_XF_B 4               // ... waves 4..7
#endif
#ifndef POCR_WINO_DBG
#define POCR_WINO_DBG 0                // tools/conv_wino_bench.hip: 1 no transform arithmetic, 2 no activation loads, 4 no weight loads, 8 no A reads, 16 no epilogue, 32 no barrier in the loop (results are then wrong)
#endif
#ifndef POCR_WINO_CM_IN
#define POCR_WINO_CM_IN 0              // experiment: input in the chunk-major layout [row][32-channel chunk][pixel][128 B] instead of P2's [row][pixel][chunk][128 B]
#endif
#ifndef POCR_WINO_WSPREAD
#define POCR_WINO_WSPREAD 0            // 1: the next chunk's weight loads spread over the units (two per unit) instead of four in each of the first three
#endif
#ifndef POCR_WINO_NSX
#define POCR_WINO_NSX 2                // experiment: 1 = one channel tile per wave (half the accumulators and weights: frees registers; results of the other tile are garbage)
#endif
#ifndef POCR_WINO_DEPTH
#define POCR_WINO_DEPTH 1              // chunks the transform's activation loads run ahead of their use (2 needs a second staging set: 32 registers)
#endif
#ifndef POCR_WINO_ROT
#define POCR_WINO_ROT 0                // experiment: workgroup b starts its K loop at chunk (b >> 3) % nchunks (the CUs of an XCD then stream different weights at any time)
#endif
#ifndef POCR_WINO_AHEAD
#define POCR_WINO_AHEAD 1              // A fragments requested this many units before use
#endif

template <int TH, int POOLH, int POOLW, int ACT, bool BN>
__global__ __launch_bounds__(512, 2) void conv3x3_wino_kernel(WinoArgs a) {
    constexpr int NS = POCR_WINO_NSX, NROW = TH + 2, NT = 64, WU = 128, DEP = POCR_WINO_DEPTH;
    constexpr int VROW = 8;                             // rows per plane in LDS: every wave transforms "its" row - wave j >= NROW a row of zeros nobody reads - so the loop has no branch
    constexpr int V_U = 4 * VROW * 128;                 // 16-byte units per V buffer: [plane 4][halo row][h | l][lane 64]
    constexpr int PST = 68, M_DW = 4 * TH * 16 * PST;   // epilogue image: [plane][row][pair] x 68 dwords (64 channels + 4: the four pair groups of a wave store to different banks)
    static_assert(M_DW * 4 <= 2 * V_U * 16, "the output transform reuses the V buffers");
    static_assert(POOLH == 1 || TH % 2 == 0, "H-pool needs an even tile height");
    static_assert(NROW <= 8, "one wave per halo row");
    __shared__ u32x4 lds[2 * V_U];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int k = wave & 3, nh = wave >> 2;

    // block -> (channel tile, pixel tile), as conv3x3_bf16x3_kernel
    int nt, ptile;
    {
        const int tn = a.tiles_n, P = a.n_ptiles;
        if (tn >= 2 && tn <= 8 && (tn & (tn - 1)) == 0) {
            const int G = a.xcd_g > 1 && tn % a.xcd_g == 0 ? a.xcd_g : 1, xg = tn / G;
            const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3, groups = 8 / xg;
            nt = (xcd % xg) * G + kb % G;
            ptile = (kb / G) * groups + xcd / xg;
            if (ptile >= P) return;
        } else {
            int b = blockIdx.x;
            const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = b & 7, kb = b >> 3;
            b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kb;
            nt = b % tn;
            ptile = b / tn;
        }
    }
    POCR_TRACE_STAMP(0);
    const WinoTile wtile = a.wtiles[ptile];
    const int nchunks = a.cin >> 5;
#if POCR_WINO_ROT
    const int rot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 3) % nchunks);
    auto rc = [&](int c) { const int t = c + rot; return t >= nchunks ? t - nchunks : t; };
#else
    auto rc = [&](int c) { return c; };
#endif

    // ---- transform role: wave j < NROW owns halo row j; lane (li = pair, kq = channel octet)
    // The activations come through BUFFER loads: a pixel outside the image (conv padding, tile overhang, an empty half tile) gets
    // an offset beyond the descriptor's range and the load unit returns zeros - no branch, no select, and the loads in flight
    // stay countable (s_waitcnt vmcnt(n)).  Per lane: the byte offsets of its four pixels; the chunk is the scalar offset.
    const bool xf_wave = wave < NROW;
    constexpr unsigned kOutOfRange = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)min(a.x_bytes ? a.x_bytes : kOutOfRange, kOutOfRange), 0x00020000);
    unsigned xoff[4], xcm = 0;
    {
        const int half = li >> 3, pp = li & 7;
#if POCR_WINO_DBG & 64
        const int ln = (blockIdx.x >> 3) & 7, hw = 0;     // every workgroup reads one of eight tiles: the activation loads hit L2
#else
        const int ln = wtile.line[half], hw = wtile.ht_wt[half];
#endif
        const int lnc = ln >= 0 ? ln : 0;
        const int Win = a.line_w[lnc];
        const int hi = (hw >> 16) - 1 + wave, px0 = (hw & 0xffff) * 16 + 2 * pp - 1;
        const bool row_ok = xf_wave && ln >= 0 && hi >= 0 && hi < a.H;
#if POCR_WINO_CM_IN
        const unsigned base = (unsigned)(a.in_off[lnc] * 4) + (unsigned)(hi * Win * a.cin * 4 + px0 * 128 + kq * 16);
        xcm = (unsigned)Win * 128u;
#pragma unroll
        for (int i = 0; i < 4; ++i) xoff[i] = (row_ok && px0 + i >= 0 && px0 + i < Win) ? base + (unsigned)(i * 128) : kOutOfRange;
#else
        const unsigned base = (unsigned)(a.in_off[lnc] * 4) + (unsigned)((hi * Win + px0) * a.cin * 4 + kq * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) xoff[i] = (row_ok && px0 + i >= 0 && px0 + i < Win) ? base + (unsigned)(i * a.cin * 4) : kOutOfRange;
#if POCR_WINO_DBG & 128
        // same bytes, CONTIGUOUS: every load instruction of a wave reads 1 KB (8 whole lines), the wave's eight loads 8 KB of its row (results wrong)
#pragma unroll
        for (int i = 0; i < 4; ++i) xoff[i] = (unsigned)(a.in_off[lnc] * 4) + (unsigned)(max(hi, 0) * Win * a.cin * 4) + (unsigned)(((hw & 0xffff) * 16 * 4 + i * 16) * 128) + (unsigned)lane * 16u;
#endif
#endif
    }
    unsigned long long lat_sum = 0; unsigned lat_n = 0; (void)lat_sum; (void)lat_n;
    unsigned dummy = 0; (void)dummy;
    u32x4 sh_[DEP][4], sl_[DEP][4];
    auto ldX = [&](int chunk, int set = 0) {
        u32x4 (&sh)[4] = sh_[set], (&sl)[4] = sl_[set];
#if POCR_WINO_DBG & 8192
        const unsigned long long ti0 = wall_clock64();   // how long does the ISSUE of the eight loads take (no wait for data)?
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if POCR_WINO_DBG & 2
            sh[i] = (u32x4){(unsigned)chunk, 0u, (unsigned)i, 0u}; sl[i] = sh[i];
#else
#if POCR_WINO_CM_IN
            const unsigned o = xoff[i] == kOutOfRange ? kOutOfRange : xoff[i] + (unsigned)chunk * xcm;
            sh[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)o, 0, 0);
            sl[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)o + 64, 0, 0);
#else
            sh[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)xoff[i], rc(chunk) * 128, 0);
#if POCR_WINO_DBG & 512
            sl[i] = sh[i] ^ 0x00010001u;                  // half the activation bytes (results wrong)
#else
            sl[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)xoff[i] + 64, rc(chunk) * 128, 0);
#endif
#endif
#endif
        }
#if POCR_WINO_DBG & 8192
        asm volatile("" ::: "memory");
        lat_sum += wall_clock64() - ti0; ++lat_n;
#endif
#if POCR_WINO_DBG & 1024
        {   // how long does this burst of eight loads take from issue to the last byte?  (everything older is drained first)
            const unsigned long long t0 = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lat_sum += wall_clock64() - t0; ++lat_n;
        }
#endif
    };
    // v_fma_mix reads f16 halves in place (no unpacking) and writes a rounded f16 half in place (no packing).  Plane by plane, so
    // that few values are alive at a time: rebuild 2^11 x of the pixels a plane needs (one op per value), one add per value,
    // two ops per value for its h / l halves, two 16-byte LDS stores per plane.
    auto xform = [&](int buf, int set = 0) {
        u32x4 (&sh)[4] = sh_[set], (&sl)[4] = sl_[set];
        const float s_up = kF16x2Scale, s_dn = 1.0f / kF16x2Scale, s_neg = -kF16x2Scale;
        u32x4 *dst = lds + buf * V_U + wave * 128 + lane;
#if POCR_WINO_DBG & 2048
        // the loads are waited for and touched (one xor per register), but what goes to LDS are constants
#pragma unroll
        for (int q = 0; q < 4; ++q) { dummy ^= sh[q][0] ^ sl[q][1]; dst[q * VROW * 128] = (u32x4){(unsigned)buf, 1u, 2u, 3u}; dst[q * VROW * 128 + 64] = (u32x4){(unsigned)buf, 5u, 6u, 7u}; }
#elif POCR_WINO_DBG & 1
#pragma unroll
        for (int q = 0; q < 4; ++q) { dst[q * VROW * 128] = sh[q]; dst[q * VROW * 128 + 64] = sl[q]; }
#else
        auto rebuild = [&](float (&x)[8], const u32x4 &h, const u32x4 &l) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const unsigned hd = h[m], ld = l[m];
                asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(x[2 * m]) : "v"(hd), "s"(s_up), "v"(ld));
                asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(x[2 * m + 1]) : "v"(hd), "s"(s_up), "v"(ld));
            }
        };
        auto emit = [&](int q, const float (&v)[8]) {
            u32x4 oh, ol;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                unsigned a, b;
                asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(a) : "v"(v[2 * m]), "s"(s_dn));
                asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(a) : "v"(v[2 * m + 1]), "s"(s_dn));
                asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(b) : "v"(a), "s"(s_neg), "v"(v[2 * m]));
                asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(b) : "v"(a), "s"(s_neg), "v"(v[2 * m + 1]));
                oh[m] = a; ol[m] = b;
            }
            dst[q * VROW * 128] = oh;
            dst[q * VROW * 128 + 64] = ol;
        };
        float xa[8], xb[8], v[8];
        rebuild(xa, sh[0], sl[0]);                       // x0
        rebuild(xb, sh[2], sl[2]);                       // x2
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = xa[c] - xb[c];
        emit(0, v);
        rebuild(xa, sh[1], sl[1]);                       // x1
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = xa[c] + xb[c];
        emit(1, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = xb[c] - xa[c];
        emit(2, v);
        rebuild(xb, sh[3], sl[3]);                       // x3
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = xa[c] - xb[c];
        emit(3, v);
#endif
    };

    // ---- multiply role: plane k, channel tiles 4 nt + 2 nh + {0, 1}
    f32x4 acc[TH][NS], acc2[TH][NS];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int n = 0; n < NS; ++n) { acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const size_t chunk_stride = (size_t)a.cout16 * WU, tap_stride = (size_t)nchunks * chunk_stride;
    // weights: buffer loads too - one descriptor, the lane's 16 bytes as the vector offset, the (tap, chunk, channel tile) piece as
    // the scalar offset: no 64-bit address registers per lane
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wfrag), 0, (int)min((size_t)12 * tap_stride * 16, (size_t)0x7fffffff), 0x00020000);
    const int wlane = lane * 16;
    const int wbase = __builtin_amdgcn_readfirstlane((int)((((size_t)nt * 4 + nh * 2) * WU + (size_t)(k * 3) * tap_stride) * 16));
    constexpr int AH = POCR_WINO_AHEAD, RING = AH + 1;
    u32x4 bw[2][3][NS][2], ar[RING][2];
    auto ldW = [&](u32x4 (&dst)[NS][2], int chunk, int dy) {         // (past the last chunk: the last one again - read, never used)
        const int so = wbase + (int)(((size_t)dy * tap_stride + (size_t)rc(min(chunk, nchunks - 1)) * chunk_stride) * 16);
#pragma unroll
        for (int n = 0; n < NS; ++n) {
#if POCR_WINO_DBG & 4
            dst[n][0] = (u32x4){(unsigned)chunk, (unsigned)dy, 0u, 0u}; dst[n][1] = dst[n][0]; (void)so;
#else
            dst[n][0] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + n * WU * 16, so, 0);
            dst[n][1] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + n * WU * 16 + 1024, so, 0);
#endif
        }
    };
    auto ldW1 = [&](u32x4 (&dst)[2], int chunk, int dy, int n) {
        const int so = wbase + (int)(((size_t)dy * tap_stride + (size_t)rc(min(chunk, nchunks - 1)) * chunk_stride + n * WU) * 16);
#if POCR_WINO_DBG & 4
        dst[0] = (u32x4){(unsigned)chunk, (unsigned)dy, 0u, 0u}; dst[1] = dst[0]; (void)so;
#else
        dst[0] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so, 0);
        dst[1] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + 1024, so, 0);
#endif
    };
    auto rdA = [&](u32x4 (&dst)[2], int buf, int j) {
        const u32x4 *p = lds + buf * V_U + (k * VROW + j) * 128 + lane;
#if POCR_WINO_DBG & 8
        dst[0] = (u32x4){(unsigned)j, 0u, 0u, 0u}; dst[1] = dst[0]; (void)p;
#else
        dst[0] = p[0]; dst[1] = p[64];
#endif
    };

    ldX(0, 0);
    if constexpr (DEP == 2) ldX(nchunks > 1 ? 1 : 0, 1);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) ldW(bw[0][dy], 0, dy);
    xform(0, 0);
    ldX(min(DEP, nchunks - 1), 0);                     // DEP 1: chunk 1 into the only set; DEP 2: chunk 2 into set 0 (set 1 holds chunk 1)
    __syncthreads();
    POCR_TRACE_STAMP(1);
    // The two waves of a SIMD (w and w + 4: same plane, the two channel halves) run their transform - a burst of ~130 vector
    // instructions and eight 1 KB loads - at DIFFERENT points of the chunk (XF = unit behind which it runs), so that one multiplies
    // while the other transforms and the load unit sees the requests of four waves at a time, not eight.  Two copies of the loop
    // (the choice is wave-uniform and made once): inside a copy there is no branch, so the loads in flight stay countable.
    auto main_loop = [&](auto xf_c) {
        constexpr int XF = decltype(xf_c)::value;
        for (int c0 = 0; c0 < nchunks; c0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                // two chunks per trip: buffer and weight-set parity are static
                const int chunk = c0 + u;
                if (chunk >= nchunks) break;             // (uniform)
                const int buf = u;
#pragma unroll
                for (int q = 0; q < AH; ++q) rdA(ar[q % RING], buf, q);
#pragma unroll
                for (int j = 0; j < NROW; ++j) {
                    if (j + AH < NROW) rdA(ar[(j + AH) % RING], buf, j + AH);
                    // the next chunk's weights, one (dy, channel tile) piece = two 1 KB loads per unit
#if POCR_WINO_WSPREAD
#pragma unroll
                    for (int pc = (j * 3 * NS) / NROW; pc < ((j + 1) * 3 * NS) / NROW; ++pc) ldW1(bw[u ^ 1][pc / NS][pc % NS], chunk + 1, pc / NS, pc % NS);
#else
                    if (j < 3) ldW(bw[u ^ 1][j], chunk + 1, j);      // all of them BEFORE the transform's loads: s_waitcnt vmcnt counts in order
#endif
                    const u32x4 ah = ar[j % RING][0], al = ar[j % RING][1];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int r = j - dy;
                        if (r < 0 || r >= TH) continue;
                        u32x4 (&bc)[NS][2] = bw[u][dy];
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc2[r][n] = mfma16_f16(al, bc[n][0], acc2[r][n]);
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc[r][n] = mfma16_f16(ah, bc[n][0], acc[r][n]);
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc2[r][n] = mfma16_f16(ah, bc[n][1], acc2[r][n]);
                    }
                    if (j == XF) {
                        // the other buffer: its last readers passed the barrier of the previous chunk.  After the last chunk: a
                        // transform of the clamped re-read that nobody uses (keeps the loads in flight the same in every trip)
                        // (two chunks per trip and chunk = c0 + u: the set of chunk + 1 is static, (u + 1) & 1, when c0 is even)
                        xform(buf ^ 1, DEP == 2 ? (u + 1) & 1 : 0);
                        ldX(min(chunk + 1 + DEP, nchunks - 1), DEP == 2 ? (u + 1) & 1 : 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#if !(POCR_WINO_DBG & 32)
                __syncthreads();
#endif
            }
        }
    };
    if (nh == 0) main_loop(std::integral_constant<int, POCR_WINO_XF_A>{});
    else main_loop(std::integral_constant<int, POCR_WINO_XF_B>{});
    POCR_TRACE_STAMP(2);
#if POCR_WINO_DBG & 2048
    if (dummy == 0x12345u) a.y[tid] = 1.f;
#endif
#if (POCR_WINO_DBG & (1024 | 8192)) && defined(POCR_BF16X3_TRACE)
    if (lane == 0 && wave == 2 && blockIdx.x < (1u << 15)) { g_conv_trace[blockIdx.x * 8 + 6] = lat_sum; g_conv_trace[blockIdx.x * 8 + 7] = lat_n; }
#endif
#if POCR_WINO_DBG & 16
    if (acc[0][0][0] == 123.456f) a.y[tid] = acc[1][1][1] + acc2[2][0][0];
    return;
#endif

    // ---- output transform through LDS
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int n = 0; n < NS; ++n) acc[r][n] += acc2[r][n] * (1.0f / kF16x2Scale);
    float *mlds = reinterpret_cast<float *>(lds);
    {
        float *mw = mlds + (size_t)(k * TH * 16 + 4 * kq) * PST + nh * 32 + li;
#pragma unroll
        for (int r = 0; r < TH; ++r)
#pragma unroll
            for (int n = 0; n < NS; ++n)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) mw[(r * 16 + rr) * PST + n * 16] = acc[r][n][rr];
    }
    __syncthreads();
    constexpr int RG = TH / POOLH;                      // row groups per tile
    constexpr int NIT = (RG * 128 + 511) / 512;
    unsigned rmax = 0u;
    const int co0 = nt * NT + (tid & 7) * 8;
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(a.bias + co0), b1 = *reinterpret_cast<const f32x4 *>(a.bias + co0 + 4);
    const float bias[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    float sc[8], sf[8];
    if constexpr (BN) {
        const f32x4 s0 = *reinterpret_cast<const f32x4 *>(a.bn_scale + co0), s1 = *reinterpret_cast<const f32x4 *>(a.bn_scale + co0 + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4 *>(a.bn_shift + co0), t1 = *reinterpret_cast<const f32x4 *>(a.bn_shift + co0 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { sc[c] = s0[c]; sc[c + 4] = s1[c]; sf[c] = t0[c]; sf[c + 4] = t1[c]; }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 512, oct = e & 7, p = (e >> 3) & 15, rg = e >> 7;
        if (rg >= RG) break;
        const int half = p >> 3, pp = p & 7;
        const int ln = wtile.line[half], hw = wtile.ht_wt[half];
        if (ln < 0) continue;
        const int Wo = a.line_w[ln], h0 = hw >> 16, wc = (hw & 0xffff) * 16 + 2 * pp;
        float y[2][8];                                  // [column of the pair][channel], after bias / activation / BatchNorm / H-pool
#pragma unroll
        for (int ph = 0; ph < POOLH; ++ph) {
            const int r = rg * POOLH + ph;
            f32x4 m[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 *src = reinterpret_cast<const f32x4 *>(mlds + (size_t)((q * TH + r) * 16 + p) * PST + oct * 8);
                m[q][0] = src[0]; m[q][1] = src[1];
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float m0 = m[0][c >> 2][c & 3], m1 = m[1][c >> 2][c & 3], m2 = m[2][c >> 2][c & 3], m3 = m[3][c >> 2][c & 3];
                float t0 = apply_act(((m0 + m1) + m2) + bias[c], ACT), t1 = apply_act(((m1 - m2) - m3) + bias[c], ACT);
                if constexpr (BN) { t0 = t0 * sc[c] + sf[c]; t1 = t1 * sc[c] + sf[c]; }
                y[0][c] = ph ? fmaxf(y[0][c], t0) : t0;
                y[1][c] = ph ? fmaxf(y[1][c], t1) : t1;
            }
        }
        if (h0 + rg * POOLH >= a.Ho || wc >= Wo) continue;
        const int ho = (h0 + rg * POOLH) / POOLH, Wout = Wo / POOLW;
        char *prow = reinterpret_cast<char *>(a.y + a.out_off[ln]) + p2_channel_bytes(co0);
        const size_t pix_bytes = (size_t)a.out_stride * 4;
#pragma unroll
        for (int col = 0; col < (POOLW == 2 ? 1 : 2); ++col) {
            f16x8 hh, ll;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float v = POOLW == 2 ? fmaxf(y[0][c], y[1][c]) : y[col][c];
                range_note(rmax, v);
                _Float16 h_, l_;
                split2_scalar(v, h_, l_);
                hh[c] = h_; ll[c] = l_;
            }
            const int wo = POOLW == 2 ? wc / 2 : wc + col;
            u32x4 *d = reinterpret_cast<u32x4 *>(prow + ((size_t)ho * Wout + wo) * pix_bytes);
            d[0] = __builtin_bit_cast(u32x4, hh);
            d[4] = __builtin_bit_cast(u32x4, ll);
        }
    }
    range_publish(a.range_max, rmax, lane);
#ifdef POCR_BF16X3_TRACE
    POCR_TRACE_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    POCR_TRACE_STAMP(4);
#endif
}

}  // namespace pocr
