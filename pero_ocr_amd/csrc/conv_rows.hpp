// conv_rows.hpp — the recogniser's 3x3 convolutions conv3 .. conv9 in the default f16x2 / P2 mode (round 5).  The main loop is the
// halo-row streaming loop of conv_bf16x3.hpp ("ROWS": same fragments, same MFMAs in the same order - the two kernels give the same
// bits; tools/conv_wino_bench.hip and the GPU suite check that); what is different is everything around it, from taking the
// one-tile kernel apart (profiles/r05_conv_rows.txt):
//   * the weights are the MFMA's A operand (mfma_conv_f16<true>): the result is [channel][pixel], a lane holds four consecutive
//     channels of one pixel, and the tile's output is put together in LDS and stored as whole 128-byte lines (conv_epilogue_staged:
//     the epilogue's 32-byte pieces had cost conv3 a third of its time - the number of partial-line writes, not their bytes);
//   * weights and activations come by buffer loads: one per-lane offset register each, (tap, chunk) as a scalar offset - no 64-bit
//     vector address arithmetic in the loop (196-240 registers against 250); halo pixels outside the image carry an out-of-range
//     mark and read as zeros.  STORES are never masked that way (conv_epilogue_staged says why);
//   * channel constants (bias, batch-norm scale / shift) wait in LDS.
// PERS = true is the PERSISTENT form: a grid of two workgroups per CU, each walking pixel tiles of ONE channel tile.  With one tile
// per workgroup every tile starts with a chain of round trips that nothing inside the workgroup hides - tile descriptor -> line
// descriptors -> first halo tile and weight fragments -> LDS -> barrier: 4.5-5.3 us per tile in every layer (conv3 5.2 of 13.8 us,
// conv9 4.4 of 55).  Here the descriptors of tile t + 1 are scalar loads issued when tile t starts, chunk 0 of tile t + 1 is requested
// between the main loop and the epilogue of tile t and lands while the epilogue runs, and the weight stream wraps to chunk 0 in the
// last chunk by itself (same channel tile): the next tile starts with one LDS write and one barrier.  It needs an even number of
// 32-channel chunks (the weight-set parity then lines up: cin = 64 .. 512) and a block -> tile mapping that keeps a workgroup on one
// channel tile (pocr_hip.hip: launch_conv_rows).  Measured: the prologue disappears from the stamps and the main loop grows by as much -
// the two workgroups of a CU had covered each other's prologue already; conv3 alone gains 8 % (shipped), the deeper layers lose 2-4 %.
#pragma once
#include "conv_bf16x3.hpp"

namespace pocr {

// PERS = false: one block per workgroup (the tile loop runs once and everything that looks at a next tile is compiled out)
template <int TH, int MW, int NS, int WM, int POOLH, int POOLW, int ACT, bool BN, int MINW, bool PERS = false>
__global__ __launch_bounds__(256, MINW) void conv3x3_rows_kernel(ConvArgs a) {
    constexpr int NWAVE = 4, WN = NWAVE / WM, KC = 32, WU = 2 * 64;
    static_assert(MW % WM == 0, "column strips must divide among the M waves");
    constexpr int TW = 16 * MW, MWW = MW / WM, MS = TH * MWW, NT = NS * WN * 16, NTHR = 256;
    constexpr int HH = TH + 2, HW = TW + 2, NP = HH * HW, NPPAD = (NP + 15) / 16 * 16;
    constexpr int PS = 4 * NPPAD, A_U = 2 * PS;         // 16-byte units per plane / per A tile ([plane][octet][pixel])
    constexpr int NP8 = (NP + 7) / 8 * 8, A_LD = (8 * NP8 + NTHR - 1) / NTHR;
    static_assert(POOLH == 1 || TH % 2 == 0, "H-pool needs an even tile height");
    // (the A buffers double as the staging area of the epilogue: conv_epilogue_staged)
    constexpr int LDS_U = 2 * A_U > conv_stage_units(TH, TW, POOLH, POOLW, NT) ? 2 * A_U : conv_stage_units(TH, TW, POOLH, POOLW, NT);
    __shared__ u32x4 ldsA[LDS_U];
    __shared__ float cst[(BN ? 3 : 1) * NT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int wm = wave % WM, wn = wave / WM;
    const unsigned total = a.nblocks > 0 ? (unsigned)a.nblocks : gridDim.x;
    const int nchunks = a.cin / KC;
    const bool chain = (nchunks & 1) == 0;              // (uniform) tile-to-tile prefetch possible
    constexpr unsigned kOut = 0x7FFF0000u;              // out-of-range mark of a load (with the chunk's scalar offset on top it stays below 2^31 and above any line image)

    // virtual block -> (channel tile, pixel tile), as conv3x3_bf16x3_kernel; false: no such tile
    const int tn = a.tiles_n;
    const int P = a.tiles ? a.n_ptiles : a.tiles_w * a.tiles_h * a.n;
    auto map = [&](unsigned vb, int &nt, int &ptile) -> bool {
        if (tn >= 2 && tn <= 8 && (tn & (tn - 1)) == 0) {
            const int G = a.xcd_g > 1 && tn % a.xcd_g == 0 ? a.xcd_g : 1, xg = tn / G;
            const int xcd = vb & 7, k = vb >> 3, groups = 8 / xg;
            nt = (xcd % xg) * G + k % G;
            ptile = (k / G) * groups + xcd / xg;
        } else {
            const int nb = (int)total, q = nb >> 3, r = nb & 7, xcd = vb & 7, k = vb >> 3;
            const int b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
            nt = b % tn;
            ptile = b / tn;
        }
        return ptile < P;
    };
    struct Tile { int h0, w0, Win; size_t in_base, out_base; };
    auto uniform64 = [](size_t v) -> size_t {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return ((size_t)hi << 32) | lo;
    };
    auto desc = [&](int ptile, Tile &t) {
        int wt, ht, img;
        if (a.tiles) {
            const PixelTile pt = a.tiles[ptile];
            img = pt.line; ht = pt.ht_wt >> 16; wt = pt.ht_wt & 0xffff;
            t.Win = a.line_w[img];
            t.in_base = (size_t)a.in_off[img];
            t.out_base = (size_t)a.out_off[img];
        } else {
            wt = ptile % a.tiles_w;
            ht = (ptile / a.tiles_w) % a.tiles_h;
            img = ptile / (a.tiles_w * a.tiles_h);
            t.Win = a.W;
            t.in_base = (size_t)img * a.H * a.W * a.cin;
            t.out_base = (size_t)img * (a.Ho / POOLH) * (a.Wo / POOLW) * a.out_stride;
        }
        // (everything here is uniform over the workgroup; said explicitly, or the buffer descriptors built from it are taken for
        // divergent and every buffer access becomes a loop over the lanes' descriptor values)
        t.h0 = __builtin_amdgcn_readfirstlane(ht * TH); t.w0 = __builtin_amdgcn_readfirstlane(wt * TW);
        t.Win = __builtin_amdgcn_readfirstlane(t.Win);
        t.in_base = uniform64(t.in_base); t.out_base = uniform64(t.out_base);
    };
    // the input image of a tile's line as a buffer: halo pixels outside the image get an out-of-range offset and read as zeros
    auto in_rsrc = [&](const Tile &t) {
        const size_t bytes = (size_t)a.H * t.Win * a.cin * 4;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x) + t.in_base, 0, (int)(bytes < kOut ? bytes : kOut - 1), 0x00020000);
    };
    // Staging slot e = tid + r * 256: eight consecutive lanes take the same 16-byte unit u = (plane, octet) of eight consecutive
    // halo pixels (their ds_write_b128 fill 128 contiguous bytes of one LDS row), the eight lane groups of a wave the eight units
    // of those pixels (each pixel's 128-byte chunk is read whole by one wave instruction).
    // LDS slot (16-byte units) of staging slot r: pixel p0 + 32 r, unit u - linear in r, the same for every tile
    const int st_u = (tid & 63) >> 3, st_p0 = (tid >> 6) * 8 + (tid & 7);
    const int st_lds0 = (st_u >> 2) * PS + (st_u & 3) * NPPAD + st_p0;
    unsigned xbo[A_LD];                                 // byte offset of the slot's unit in chunk 0 of the line's image, or out of range
    auto offsets = [&](const Tile &t) {
#pragma unroll
        for (int r = 0; r < A_LD; ++r) {
            const int e = tid + r * NTHR, l = e & 63, p = (e >> 6) * 8 + (l & 7), u = l >> 3;
            const int hr = p / HW, wc = p % HW;
            const int hi = t.h0 - 1 + hr, wi = t.w0 - 1 + wc;
            const bool ok = p < NP && hi >= 0 && hi < a.H && wi >= 0 && wi < t.Win;
            xbo[r] = ok ? ((unsigned)(hi * t.Win + wi) * (unsigned)(a.cin >> 2) + (unsigned)u) * 16u : kOut;
        }
    };
    u32x4 ra[A_LD];
    auto ldA = [&](__amdgpu_buffer_rsrc_t rs, int chunk) {
#pragma unroll
        for (int r = 0; r < A_LD; ++r) ra[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)xbo[r], chunk * 128, 0);
    };
    auto stA = [&](int abuf) {
#pragma unroll
        for (int r = 0; r < A_LD; ++r)
            if (st_p0 + 32 * r < NP) ldsA[abuf * A_U + st_lds0 + 32 * r] = ra[r];
    };

    int nt, ptile;
    unsigned vb = blockIdx.x;
    if (!map(vb, nt, ptile)) return;
    Tile cur, nx;
    desc(ptile, cur);
    nx = cur;

    // weight fragments: buffer loads - one per-lane offset register, (tap, chunk) as a scalar offset, the channel sub-tile as the
    // instruction's immediate: no 64-bit vector address arithmetic in the loop
    const size_t chunk_stride = (size_t)a.cout16 * WU, tap_stride = (size_t)nchunks * chunk_stride;      // 16-byte units
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wfrag), 0, (int)(9 * tap_stride * 16), 0x00020000);
    const int wlane = (int)((((size_t)nt * (NT / 16) + wn * NS) * WU + lane) * 16);
    constexpr int NROW = TH + 2, NU = NROW * MWW, AH = kRowAhead, RING = AH + 1;
    static_assert(NU >= 3 && AH >= 1 && AH <= NU, "row streaming: units per column offset");
    u32x4 bw[2][3][NS][2], ar[RING][2];
    auto ldW = [&](u32x4 (&dst)[NS][2], int chunk, int tap) {       // past the last chunk: chunk 0 again - the next tile's (same channel tile)
        const int so = (int)(((size_t)tap * tap_stride + (size_t)(chunk < nchunks ? chunk : 0) * chunk_stride) * 16);
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            dst[n][0] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane + n * WU * 16, so, 0);
            dst[n][1] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane + n * WU * 16 + 1024, so, 0);
        }
    };
    auto rdA = [&](u32x4 (&dst)[2], int abuf, int dx, int unit) {
        const u32x4 *p = ldsA + abuf * A_U + (unit / MWW) * HW + dx + li + kq * NPPAD + (wm * MWW + unit % MWW) * 16;
        dst[0] = p[0]; dst[1] = p[PS];
    };

    // channel constants of the epilogue (bias, batch-norm scale / shift of the NT channels) wait in LDS: no registers in the main loop
    for (int c = tid; c < NT; c += NTHR) {
        const int co = nt * NT + c;
        cst[c] = a.bias[co];
        if constexpr (BN) { cst[NT + c] = a.bn_scale[co]; cst[2 * NT + c] = a.bn_shift[co]; }
    }
    unsigned rmax = 0u;                                 // f16x2 range guard (conv_igemm.hpp: range_note), published once per workgroup

    __amdgpu_buffer_rsrc_t rs = in_rsrc(cur);
    bool pre = false;                                   // chunk 0 of `cur` is in LDS buffer 0 and bw[0] holds its first weight set
    for (;;) {
        const unsigned vbn = vb + gridDim.x;
        int nt_n = 0, ptile_n = 0;
        const bool more = PERS && vbn < total && map(vbn, nt_n, ptile_n);
        const bool has_next = chain && more;
        if (more) desc(ptile_n, nx);                    // scalar loads: in flight behind this tile's main loop
        if (!pre) {
            offsets(cur);
            ldA(rs, 0);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) ldW(bw[0][dy], 0, dy * 3);
            stA(0);
            __syncthreads();
        }
        f32x4 acc[MS][NS], acc2[MS][NS];
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int s = 0; s < NS; ++s) { acc[m][s] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[m][s] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

        for (int c0 = 0; c0 < nchunks; c0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {               // two chunks = six groups: the weight-set parity is static
                const int chunk = c0 + u;
                if (chunk >= nchunks) break;            // (uniform)
                const int abuf = chunk & 1;
                const int lchunk = chunk + 1 < nchunks ? chunk + 1 : chunk;    // (past the last chunk: this one again - read, never used; the load counts stay uniform)
#pragma unroll
                for (int q = 0; q < AH; ++q) rdA(ar[q % RING], abuf, 0, q);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int par = (u * 3 + dx) & 1;
                    const int ndx = dx == 2 ? 0 : dx + 1, nchunk = dx == 2 ? chunk + 1 : chunk;
#pragma unroll
                    for (int q = 0; q < NU; ++q) {
                        const int qq = dx * NU + q, pq = qq + AH;
                        if (pq < 3 * NU) rdA(ar[pq % RING], abuf, pq / NU, pq % NU);
                        if (q < 3) ldW(bw[par ^ 1][q], nchunk, q * 3 + ndx);
                        if (dx == 0 && q == (kRowLdaQ < NU ? kRowLdaQ : NU - 1)) ldA(rs, lchunk);
                        const int j = q / MWW, mw = q % MWW;
                        const u32x4 ah = ar[qq % RING][0], al = ar[qq % RING][1];
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int r = j - dy;
                            if (r < 0 || r >= TH) continue;
                            const int m = r * MWW + mw;
                            u32x4 (&bc)[NS][2] = bw[par][dy];
#pragma unroll
                            for (int n = 0; n < NS; ++n) acc2[m][n] = mfma_conv_f16<true>(al, bc[n][0], acc2[m][n]);
#pragma unroll
                            for (int n = 0; n < NS; ++n) acc[m][n] = mfma_conv_f16<true>(ah, bc[n][0], acc[m][n]);
#pragma unroll
                            for (int n = 0; n < NS; ++n) acc2[m][n] = mfma_conv_f16<true>(ah, bc[n][1], acc2[m][n]);
                        }
                        __builtin_amdgcn_sched_barrier(0);   // units stay in source order: reads of unit q + AH, then the MFMAs of unit q
                    }
                    if (dx == kRowStaDx) stA(abuf ^ 1);    // the other A buffer: its last readers passed the barrier of the previous chunk
                }
                __syncthreads();
            }
        }

        // the NEXT tile's chunk 0 is requested here and lands while the epilogue computes and stores (the weight fragments of its first
        // group are in bw[0] already: the stream wrapped in the last chunk)
        if (has_next) { offsets(nx); rs = in_rsrc(nx); ldA(rs, 0); }
        // ---- epilogue of `cur` (conv_bf16x3.hpp: conv_epilogue_staged)
        conv_epilogue_staged<TH, MWW, NS, WM, POOLH, POOLW, ACT, BN, NT, TW>(acc, acc2, cst, cst + NT, cst + 2 * NT, reinterpret_cast<char *>(ldsA),
                                                                             a.y + cur.out_base, cur.h0, cur.w0, cur.Win, a.Ho, a.out_stride, nt, rmax);
        if (!more) break;
        cur = nx; vb = vbn; nt = nt_n;                  // (the mapping keeps nt; an odd chunk count re-stages from scratch)
        pre = has_next;
        if (pre) { stA(0); __syncthreads(); }           // (buffer 0: its last readers passed the barrier of the second-last chunk)
        else { rs = in_rsrc(cur); __syncthreads(); }
    }
    range_publish(a.range_max, rmax, lane);
}

}  // namespace pocr
