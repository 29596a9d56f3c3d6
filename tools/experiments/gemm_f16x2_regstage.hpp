// gemm_f16x2.hpp — the GEMM-shaped layers (LSTM input projections, encoder linears, the aggregation conv) on the f16 matrix
// pipe in the f16x2 arithmetic of conv_bf16x3.hpp (fp32 operands as two f16 planes, three MFMAs per 32-deep product block),
// as a PERSISTENT kernel of 256 x 128 tiles whose two wave groups alternate between a memory phase and a multiply phase.
//
// Replaces (same arithmetic, same accumulation order, bit-identical results) conv3x3_bf16x3_kernel's GEMM mode for
//   torch.nn.LSTM's W_ih x + b for all frames at once, nn.TransformerEncoderLayer's linears (transformer.py:366-385) and the
//   (H/8) x 1 aggregation conv (transformer.py:351-355) - the layers whose K is one tap deep, where that kernel's main loop
//   sat at 37 % MFMA issue (profiles/r03_gemm_pipe.txt: a barrier per 48 MFMAs with all four waves in lockstep, 128 x 128
//   tiles, 27 % of a K = 512 tile's life in prologue and epilogue).
//
// Operands.  A: activations in the P2 layout ([row][K/32][32 x f16 h | 32 x f16 l], written by the producer's epilogue):
//   a row's 32-channel chunk is one 128-byte line.  B: weights in fragment order wsplit[K/32][N/16][plane][lane][8 x f16]
//   (build_wsplit): the (chunk, 16-column tile, plane) piece is 1 KB in exactly the order the MFMA's B operand lanes read it.
// Tile: 256 rows x 128 columns per workgroup, 8 waves as 4 (rows) x 2 (columns): a wave owns 64 x 64 = 4 x 4 MFMA tiles x
//   (main, cross) accumulators = 128 registers.
// Stage = one 32-deep chunk = 32 KB of A + 16 KB of B: six 16-byte pieces per thread, requested TWO stages ahead into
//   registers, written to LDS one stage ahead (two stage buffers, 96 KB: one workgroup per CU).
//   A image in LDS: row-major [256][128 B] with the 16-byte unit index XOR-ed by (row >> 1) & 7 - applied when the piece is
//   FETCHED (lane l of a row's eight lanes fetches unit l ^ swz), so eight lanes still read one whole 128-byte line and store
//   128 contiguous bytes, and the MFMA A fragment (ds_read_b128: 16 rows x one unit per 16-lane group) touches every bank
//   once.  B image: the fragments themselves.
//   (First version, kept in the history and in profiles/r04_gemm_dma_*: the pieces copied by the load unit itself,
//   global_load_lds_dwordx4, three buffers, counted vmcnt across raw barriers - bit-identical too, but an LDS-DMA piece costs
//   the CU's load path ~50 cycles per KB whatever the L2 hit rate, row pitch or XCD mapping: ~20 B/clk/CU against the 31 this
//   tile needs at full MFMA rate.)
// Persistent: the grid is one workgroup per CU; a workgroup walks its tiles as ONE flat stream of stages, so the loads of the
//   next tile's first stages are in flight while the current tile's epilogue converts and stores (302 MB of fp32 for the c2
//   projection).
// Tile order: XCD x (blocks b % 8 == x, observed placement; a speed matter only) owns a group of nb column tiles and every
//   (8 nb / N-tiles)-th row tile; its workgroups take consecutive (row tile, column tile) pairs, so the column tiles of a row
//   tile run at the same time on the XCD whose L2 holds that A tile.
// GATHER (aggregation conv): row m of the GEMM is frame t of line i, stage p = (chunk c = p / taps, tap = p % taps) lies at
//   in_off[i] + ((tap * W_i + t) * cin + 32 c) * 4 bytes of the conv activation: only the source addresses change.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_igemm.hpp"
#include "conv_bf16x3.hpp"

namespace pocr {

struct GemmP2Args {
    const void *a;            // P2 activations
    const void *w;            // wsplit [nk][N16][2][64][8 x f16]
    const float *bias;        // [N16 * 16]
    void *y;                  // fp32 [M][ldy] or P2 with the same pixel pitch (ldy * 4 bytes)
    int32_t M, nk, N16, n_valid, ldy;
    int64_t lda;              // bytes between rows of `a` (plain GEMM; >= nk * 128)
    int32_t mt_total, nt_total;          // 256-row / 128-column tiles
    int32_t nb;                          // column tiles per XCD block (divides nt_total; nt_total / nb divides 8)
    // GATHER
    const int32_t *row_line, *row_t, *line_w;
    const int64_t *in_off;    // element (float) offsets of the lines in `a`
    int32_t cpt, ntap, cin;   // chunks per tap, taps, input channels: stage p = (chunk p / ntap, tap p % ntap) - the K order of conv3x3_bf16x3_kernel's tap loops
    unsigned *range_flag;     // [1]: bit pattern of max |y| (atomicMax) - the f16x2 range guard; NULL = off
};

#ifndef POCR_GEMM_DBG
#define POCR_GEMM_DBG 0              // tools/gemm_bench.hip ablations (results wrong, time only): 1 no staging in the loop, 2 no fragment reads, 4 no MFMAs
#endif
constexpr int kGemmBM = 256, kGemmBN = 128, kGemmThreads = 512;
constexpr int kGemmStageU = 3072;                 // 16-byte units per stage: 2048 of A, 1024 of B
constexpr int kGemmBiasMax = 3072;                // bias columns kept in LDS
constexpr int kGemmLdsU = 2 * kGemmStageU + kGemmBiasMax / 4;

template <int ACT, bool P2OUT, bool GATHER>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_f16x2_kernel(GemmP2Args a) {
    __shared__ u32x4 lds[kGemmLdsU];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int wm = wave & 3, wn = wave >> 2;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    // XCD x works on the column tiles of group x % NG (nb tiles) and on the row tiles (x / NG), (x / NG) + MG, ...;
    // NG = 1: every XCD sees every column tile
    const int NG = a.nt_total / a.nb, MG = 8 / NG, ngroup = xcd % NG, mgroup = xcd / NG;
    const int cnt_m = (a.mt_total - mgroup + MG - 1) / MG;
    const int q_total = cnt_m * a.nb;
    const int iters = slot < q_total ? (q_total - slot + per_xcd - 1) / per_xcd : 0;
    if (iters == 0) return;
    const int nk = a.nk, total = iters * nk;

    {   // bias -> LDS
        float *bl = reinterpret_cast<float *>(lds + 2 * kGemmStageU);
        for (int c = tid; c < a.N16 * 16 && c < kGemmBiasMax; c += kGemmThreads) bl[c] = a.bias[c];
    }

    // ---- source addresses of this thread's six 16-byte pieces of a stage.  A piece j: row (j * 8 + wave) * 8 + (lane >> 3) of
    // the tile, unit (lane & 7) ^ swz(row) of the row's 128-byte chunk, stored to LDS slot (row, lane & 7).
    const char *abase = static_cast<const char *>(a.a);
    const char *wbase = static_cast<const char *>(a.w);
    unsigned aoff[4];                                 // byte offsets from abase (activations < 4 GB)
    unsigned atap[GATHER ? 4 : 1];
    int p_it = 0, p_k = 0, p_tap = 0, p_c = 0;        // (tile iteration, stage) of the next stage to request; GATHER: its (tap, chunk)
    const char *wcol = nullptr;
    auto tile_of = [&](int it, int &m0, int &n16) {
        const int q = slot + it * per_xcd;
        m0 = ((q / a.nb) * MG + mgroup) * kGemmBM;
        n16 = (ngroup * a.nb + q % a.nb) * (kGemmBN / 16);
    };
    auto tile_addr = [&](int it) {
        int m0, n16;
        tile_of(it, m0, n16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (j * 8 + wave) * 8 + (lane >> 3);
            const int m = min(m0 + r, a.M - 1);       // rows past the end: a valid row again, never stored
            const int u = (lane & 7) ^ ((r >> 1) & 7);
            if constexpr (GATHER) {
                const int line = a.row_line[m], t = a.row_t[m], W = a.line_w[line];
                aoff[j] = (unsigned)(((size_t)a.in_off[line] + (size_t)t * a.cin) * 4 + u * 16);
                atap[j] = (unsigned)((size_t)W * a.cin * 4);
            } else {
                aoff[j] = (unsigned)((size_t)m * a.lda + u * 16);
            }
        }
        wcol = wbase + (size_t)n16 * 2048 + (size_t)wave * 1024 + lane * 16;
    };
    f32x4 ra[4], rb[2];
    auto ld = [&]() {                                 // request stage (p_it, p_k) into the staging registers, advance
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned off;
            if constexpr (GATHER) off = aoff[j] + (unsigned)p_tap * atap[j] + (unsigned)p_c * 128u;
            else off = aoff[j] + (unsigned)p_k * 128u;
            ra[j] = *reinterpret_cast<const f32x4 *>(abase + off);
        }
        const char *wsrc = wcol + (size_t)(GATHER ? p_tap * a.cpt + p_c : p_k) * a.N16 * 2048;
        rb[0] = *reinterpret_cast<const f32x4 *>(wsrc);                 // (tile, plane) pieces wave and wave + 8
        rb[1] = *reinterpret_cast<const f32x4 *>(wsrc + 8 * 1024);
        if constexpr (GATHER) { if (++p_tap == a.ntap) { p_tap = 0; ++p_c; } }
        if (++p_k == nk) {
            p_k = 0; p_tap = 0; p_c = 0;
            if (p_it + 1 < iters) { ++p_it; tile_addr(p_it); }
            else { p_k = nk - 1; p_tap = a.ntap - 1; p_c = a.cpt - 1; }     // past the end: the last stage again (read, never used)
        }
    };
    auto st = [&](int buf) {
        u32x4 *dst = lds + buf * kGemmStageU;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(j * 8 + wave) * 64 + lane] = __builtin_bit_cast(u32x4, ra[j]);
        dst[2048 + wave * 64 + lane] = __builtin_bit_cast(u32x4, rb[0]);
        dst[2048 + (8 + wave) * 64 + lane] = __builtin_bit_cast(u32x4, rb[1]);
    };

    f32x4 acc[4][4], acc2[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) { acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // fragment read addresses (16-byte units inside a stage): A row wm * 64 + mt * 16 + li, unit (plane * 4 + kq) ^ swz(li)
    const int sw = (li >> 1) & 7;
    const int a_h = (wm * 64 + li) * 8 + (kq ^ sw), a_l = (wm * 64 + li) * 8 + ((4 + kq) ^ sw);
    const int b_u = 2048 + (wn * 4) * 128 + lane;

    tile_addr(0);
    ld();
    st(0);
    ld();
    int c_it = 0, c_k = 0;
    unsigned rmax = 0u;                               // largest |output| as a bit pattern: non-negative floats order like integers, inf / NaN above every finite value

    // Two wave groups, half a stage apart (waves w and w + 4 share a SIMD): while group A multiplies stage g, group B writes
    // its staged pieces of stage g + 1 to LDS, requests stage g + 2 and reads the fragments of stage g; then B multiplies and A
    // stages / reads.  Each SIMD's matrix pipe is fed by one wave at a time and the other wave's memory phase hides behind it
    // (with all eight waves in lockstep every phase of a stage runs exposed).  A stage has two barriers - before the memory
    // phase and before the multiply phase - and group B passes one more at the start (A at the end): barrier instance 2 g + 1
    // separates A's memory / multiply phases of stage g and B's multiply of g - 1 / memory of g.
    //   RAW: stage g + 1 is written to LDS in the memory phase of stage g - A's before instance 2 g + 1, B's (retired by
    //        lgkmcnt(0)) before 2 g + 2 - and first read by A behind instance 2 g + 2.
    //   WAR: it goes to the buffer of stage g - 1, whose last reads (B's memory phase of g - 1, retired before instance 2 g)
    //        precede every write (A's behind instance 2 g, B's behind 2 g + 1).  Two buffers.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wn == 1) __builtin_amdgcn_s_barrier();

    auto stage = [&](int buf) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#if !(POCR_GEMM_DBG & 1)
        st(buf ^ 1);
        ld();
#endif
        const u32x4 *S = lds + buf * kGemmStageU;
        u32x4 bh[4], bl[4], ah[4], al[4];
#if POCR_GEMM_DBG & 2
#pragma unroll
        for (int n = 0; n < 4; ++n) { bh[n] = (u32x4){(unsigned)(buf + n), 1u, 2u, (unsigned)lane}; bl[n] = bh[n] ^ 5u; ah[n] = bh[n] ^ 9u; al[n] = bh[n] ^ 17u; }
        (void)S;
#else
#pragma unroll
        for (int n = 0; n < 4; ++n) { bh[n] = S[b_u + n * 128]; bl[n] = S[b_u + n * 128 + 64]; }
#pragma unroll
        for (int m = 0; m < 4; ++m) { ah[m] = S[a_h + m * 128]; al[m] = S[a_l + m * 128]; }
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#if POCR_GEMM_DBG & 4
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) { acc[m][n][0] += __builtin_bit_cast(float, ah[m][0] ^ bh[n][0]); acc2[m][n][0] += __builtin_bit_cast(float, al[m][0] ^ bl[n][0]); }
#else
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int n = 0; n < 4; ++n) acc2[m][n] = mfma16_f16(al[m], bh[n], acc2[m][n]);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = mfma16_f16(ah[m], bh[n], acc[m][n]);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc2[m][n] = mfma16_f16(ah[m], bl[n], acc2[m][n]);
        }
#endif
        if (++c_k == nk) {
            // ---- epilogue of tile c_it: + bias, activation, store; the next tile's first stages are already on their way
            c_k = 0;
            int m0, n16;
            tile_of(c_it, m0, n16);
            ++c_it;
            const float *bl_ = reinterpret_cast<const float *>(lds + 2 * kGemmStageU);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int co = (n16 + wn * 4 + n) * 16 + li;
                const float bias = bl_[min(co, kGemmBiasMax - 1)];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = apply_act(acc[m][n][r] + acc2[m][n][r] * (1.0f / kF16x2Scale) + bias, ACT);
                        rmax = max(rmax, __builtin_bit_cast(unsigned, v[r]) & 0x7fffffffu);
                    }
                    acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    quad_transpose(v, lane);
                    const int row = m0 + wm * 64 + m * 16 + kq * 4 + (li & 3), c4 = co - (li & 3);
                    if (row < a.M) {
                        if constexpr (P2OUT) {
                            u32x2 hh, ll;
                            split2_quad((f32x4){v[0], v[1], v[2], v[3]}, hh, ll);
                            u32x2 *d = reinterpret_cast<u32x2 *>(static_cast<char *>(a.y) + (size_t)row * a.ldy * 4 + p2_channel_bytes(c4));
                            d[0] = hh; d[8] = ll;
                        } else {
                            float *d = static_cast<float *>(a.y) + (size_t)row * a.ldy + c4;
                            if (c4 + 3 < a.n_valid && (a.ldy & 3) == 0) {
                                *reinterpret_cast<f32x4 *>(d) = (f32x4){v[0], v[1], v[2], v[3]};
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (c4 + k < a.n_valid) d[k] = v[k];
                            }
                        }
                    }
                }
            }
        }
    };

    for (int g = 0; g < total; g += 2) {
        stage(0);
        if (g + 1 < total) stage(1);
    }
    if (wn == 0) __builtin_amdgcn_s_barrier();
    if (a.range_flag) {
        // f16x2 range guard (pocr_hip.hip): the largest |output| of this launch, as a bit pattern
        unsigned m = rmax;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
        if (lane == 0) atomicMax(a.range_flag, m);
    }
}

inline int gemm_f16x2_grid(int M, int N, int n_cus) {
    const int mt = (M + kGemmBM - 1) / kGemmBM, nt = N / kGemmBN;
    const long tiles = (long)mt * nt;
    int g = n_cus / 8 * 8;
    while (g > 8 && (long)(g - 8) >= tiles) g -= 8;      // fewer tiles than workgroups: shrink in whole XCD rounds
    return g;
}

}  // namespace pocr
