// gemm_f16x2.hpp — the GEMM-shaped layers (LSTM input projections, encoder linears, the aggregation conv) on the f16 matrix
// pipe in the f16x2 arithmetic of conv_bf16x3.hpp (fp32 operands as two f16 planes, three MFMAs per 32-deep product block),
// as a PERSISTENT kernel whose operands are copied HBM/L2 -> LDS by the load unit itself (global_load_lds_dwordx4).
//
// Replaces (same arithmetic, same accumulation order, bit-identical results) conv3x3_bf16x3_kernel's GEMM mode for
//   torch.nn.LSTM's W_ih x + b for all frames at once, nn.TransformerEncoderLayer's linears (transformer.py:366-385) and the
//   (H/8) x 1 aggregation conv (transformer.py:351-355) - the layers whose K is one tap deep, where that kernel's main loop
//   sat at 37 % MFMA issue (profiles/r03_gemm_pipe.txt: a barrier per 48 MFMAs, operands staged through registers by
//   ds_write_b128, 27 % of a K = 512 tile's life in prologue and epilogue).
//
// Operands.  A: activations in the P2 layout ([row][K/32][32 x f16 h | 32 x f16 l], written by the producer's epilogue):
//   a row's 32-channel chunk is one 128-byte line.  B: weights in fragment order wsplit[K/32][N/16][plane][lane][8 x f16]
//   (build_wsplit): the (chunk, 16-column tile, plane) piece is 1 KB in exactly the order the MFMA's B operand lanes read it.
// Tile: 256 rows x 128 columns per workgroup, 8 waves as 4 (rows) x 2 (columns): a wave owns 64 x 64 = 4 x 4 MFMA tiles x
//   (main, cross) accumulators = 128 registers; two waves per SIMD cover each other's LDS reads and barrier waits.
// Stage = one 32-deep chunk = 32 KB of A + 16 KB of B, three stage buffers (144 KB LDS, one workgroup per CU); every wave
//   issues 6 LDS-DMA pieces of 1 KB per stage, two stages ahead of the MFMAs that use them; ONE raw s_barrier per stage, the
//   DMA queue is never drained (s_waitcnt vmcnt(N) with N = pieces of the next stage [+ the epilogue's stores]).
//   A image in LDS: row-major [256][128 B] with the 16-byte unit index XOR-ed by (row >> 1) & 7 - applied to the SOURCE address of
//   the DMA (the LDS side of a DMA is lane-linear), so a wave instruction still reads 8 whole 128-byte lines, and the MFMA A
//   fragment (ds_read_b128: 16 rows x one unit per 16-lane group) touches every bank once.  B image: the fragments themselves.
// Persistent: the grid is one workgroup per CU; a workgroup walks its tiles as ONE flat stream of stages, so the DMA of the
//   next tile's first stages is in flight while the current tile's epilogue converts and stores (302 MB of fp32 for the c2
//   projection: the stores of one tile drain under the next tile's main loop instead of idling the matrix pipe).
// Tile order: XCD x (blocks b % 8 == x, observed placement; a speed matter only) owns the row tiles x, x + 8, ...; its 32
//   workgroups take consecutive (row tile, column tile) pairs, so the 16 column tiles of a row tile run at the same time on the
//   XCD whose L2 holds that A tile, and the weights (4 MB for 512 x 2048) stay L2-resident.
// GATHER (aggregation conv): row m of the GEMM is frame t of line i, K chunk kc = tap * cpt + c lies at
//   in_off[i] + ((tap * W_i + t) * cin + 32 c) * 4 bytes of the conv activation: only the DMA source addresses change.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
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
    unsigned *range_flag;     // [8]: the f16x2 range guard (conv_igemm.hpp: range_publish); NULL = off
};

constexpr int kGemmBM = 256, kGemmBN = 128, kGemmThreads = 512;
constexpr int kGemmStageU = 3072;                 // 16-byte units per stage: 2048 of A, 1024 of B
constexpr int kGemmBiasMax = 3072;                // bias columns kept in LDS (an ordinary global load inside the stream would drain the DMA queue)
constexpr int kGemmLdsU = 3 * kGemmStageU + kGemmBiasMax / 4;

constexpr int kGemmAAux = 2;                      // cache policy of the A pieces: 2 = nt (streamed: the weights, re-read by every row tile, keep their place in L2)
// (measured and dropped in round 4, profiles/r04_gemm_split_issue.txt, r04_gemm_dma_*: DMA requests between the MFMA groups instead of in
// the memory phase, s_setprio around a stage's MFMAs, default cache policy for A)
// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4); AUX: the cache policy bits
template <int AUX, class TG, class TL>
__device__ __forceinline__ void glds16(const TG *gptr, TL *lptr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), (__attribute__((address_space(3))) void *)(lptr), 16, 0, AUX);
}

template <int ACT, bool P2OUT, bool GATHER>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_f16x2_kernel(GemmP2Args a) {
    __shared__ u32x4 lds[kGemmLdsU];                  // ONE object: a second __shared__ array makes hipcc drain the DMA queue before every ds_read
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform by construction: keeps the LDS-DMA bases (M0) and the tile bookkeeping in SGPRs
    const int wm = wave & 3, wn = wave >> 2;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    // XCD x works on the column tiles of group x % NG (nb tiles: their weights stay in its L2) and on the row tiles
    // (x / NG), (x / NG) + MG, ...; NG = 1: every XCD sees every column tile
    const int NG = a.nt_total / a.nb, MG = 8 / NG, ngroup = xcd % NG, mgroup = xcd / NG;
    const int cnt_m = (a.mt_total - mgroup + MG - 1) / MG;
    const int q_total = cnt_m * a.nb;
    const int iters = slot < q_total ? (q_total - slot + per_xcd - 1) / per_xcd : 0;
    if (iters == 0) return;
    const int nk = a.nk, total = iters * nk;

    {   // bias -> LDS (plain loads, before any DMA is in flight)
        float *bl = reinterpret_cast<float *>(lds + 3 * kGemmStageU);
        for (int c = tid; c < a.N16 * 16 && c < kGemmBiasMax; c += kGemmThreads) bl[c] = a.bias[c];
    }

    // ---- DMA source addresses of this thread's pieces: A piece j covers rows (j * 8 + wave) * 8 + (lane >> 3), slot lane & 7
    const char *abase = static_cast<const char *>(a.a);
    const char *wbase = static_cast<const char *>(a.w);
    const char *arow[4];
    size_t atap[GATHER ? 4 : 1];
    int p_it = 0, p_k = 0, p_tap = 0, p_c = 0;        // (tile iteration, stage) of the next stage to request; GATHER: its (tap, chunk)
    const char *wcol = nullptr;
    auto tile_of = [&](int it, int &m0, int &n16) __attribute__((always_inline)) {
        const int q = slot + it * per_xcd;
        m0 = ((q / a.nb) * MG + mgroup) * kGemmBM;
        n16 = (ngroup * a.nb + q % a.nb) * (kGemmBN / 16);
    };
    auto tile_addr = [&](int it) __attribute__((always_inline)) {
        int m0, n16;
        tile_of(it, m0, n16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (j * 8 + wave) * 8 + (lane >> 3);
            const int m = min(m0 + r, a.M - 1);       // rows past the end: a valid row again, never stored
            const int u = (lane & 7) ^ ((r >> 1) & 7);
            if constexpr (GATHER) {
                const int line = a.row_line[m], t = a.row_t[m], W = a.line_w[line];
                arow[j] = abase + ((size_t)a.in_off[line] + (size_t)t * a.cin) * 4 + u * 16;
                atap[j] = (size_t)W * a.cin * 4;
            } else {
                arow[j] = abase + (size_t)m * a.lda + u * 16;
            }
        }
        wcol = wbase + (size_t)n16 * 2048 + (size_t)wave * 1024 + lane * 16;
    };
    // one stage = 4 A pieces + 2 B pieces per wave; issue_a(buf, j) requests A piece j, issue_b the two B pieces, advance()
    // moves on to the next stage of the flat stream
    auto issue_a = [&](int buf, int j) __attribute__((always_inline)) {
        u32x4 *dst = lds + buf * kGemmStageU;
        const char *src;
        if constexpr (GATHER) src = arow[j] + p_tap * atap[j] + (size_t)p_c * 128;
        else src = arow[j] + (size_t)p_k * 128;
        glds16<kGemmAAux>(src, dst + (j * 8 + wave) * 64);
    };
    auto issue_b = [&](int buf) __attribute__((always_inline)) {
        u32x4 *dst = lds + buf * kGemmStageU;
        const char *wsrc = wcol + (size_t)(GATHER ? p_tap * a.cpt + p_c : p_k) * a.N16 * 2048;
        glds16<0>(wsrc, dst + 2048 + wave * 64);                    // (tile, plane) pieces wave and wave + 8
        glds16<0>(wsrc + 8 * 1024, dst + 2048 + (8 + wave) * 64);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if constexpr (GATHER) { if (++p_tap == a.ntap) { p_tap = 0; ++p_c; } }
        if (++p_k == nk) {
            p_k = 0; p_tap = 0; p_c = 0;
            if (p_it + 1 < iters) { ++p_it; tile_addr(p_it); }
            else { p_k = nk - 1; p_tap = a.ntap - 1; p_c = a.cpt - 1; }     // past the end: the last stage again (read, never used) - the DMA count per stage stays constant
        }
    };
    auto issue = [&](int buf) __attribute__((always_inline)) {                       // a whole stage at once (prologue)
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_a(buf, j);
        issue_b(buf);
        advance();
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
    issue(0);
    issue(1);
    int c_it = 0, c_k = 0;
    bool stores_young = false;                        // the previous stage ended with a full tile's epilogue: NST stores are younger than the pieces to wait for
    unsigned rmax = 0u;                               // largest |output| as a bit pattern: non-negative floats order like integers, inf / NaN above every finite value
    constexpr int NST = P2OUT ? 32 : 16;              // store instructions of one epilogue

    // Two wave groups, half a stage apart (waves w and w + 4 share a SIMD): while group A multiplies stage g, group B reads the
    // fragments of stage g from LDS and requests stage g + 2; then B multiplies and A reads stage g + 1.  Each SIMD's matrix pipe
    // is fed by one wave at a time, and the other wave's LDS / DMA phase hides behind it (with ONE phase per stage all eight
    // waves read, wait and multiply in lockstep: 262-296 TF; profiles/r04_gemm_persistent.txt).  A stage has two barriers -
    // before the read phase and before the multiply phase - and group B passes one more at the start (A at the end), so that
    // barrier instance 2 g + 1 separates A's read / multiply of stage g and B's multiply of g - 1 / read of g.
    //   RAW: the pieces of stage g + 1 are retired by their issuers (vmcnt) before the multiply barrier of stage g, i.e. before
    //        barrier instance 2 g + 1 (A) / 2 g + 2 (B); the first reader of stage g + 1 (A) starts behind instance 2 g + 2.
    //   WAR: stage g + 2 goes to the buffer of stage g - 1, whose last reads (B's, retired by lgkmcnt(0) before its multiply
    //        barrier = instance 2 g) precede every request of stage g + 2 (A: behind instance 2 g; B: behind 2 g + 1).
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // stage 0 has landed
    if (wn == 1) __builtin_amdgcn_s_barrier();

    // The epilogue of a tile (+ bias, activation, store) runs at the START of the next stage's memory phase - while the
    // other wave group multiplies - not behind the tile's last MFMAs, where the other group would wait at the barrier for it.
    // The WEIGHTS are the MFMA's A operand and the activations its B operand (same products, same sums: the bits of
    // conv3x3_bf16x3_kernel), so a lane ends up with four consecutive COLUMNS 4 kq + r of ONE row li of each 16 x 16 tile:
    // one 16-byte store per tile and lane, no transpose (the quad transpose of the first version cost four cross-lane
    // shuffles per tile: ~9 k cycles per wave and tile, exposed twice).
    auto epilogue = [&]() __attribute__((always_inline)) -> bool {
        int m0, n16;
        tile_of(c_it, m0, n16);
        ++c_it;
        const float *bl_ = reinterpret_cast<const float *>(lds + 3 * kGemmStageU);
        // a tile inside the matrix whose columns are all stored with 16-byte stores: straight-line code, exactly NST store
        // instructions per wave (what the counted vmcnt wait of the stage assumes); any other tile: per-lane guards, and the
        // plain vmcnt(6) of the stage also waits for its (older) stores
        const bool full = m0 + kGemmBM <= a.M && (n16 + kGemmBN / 16) * 16 <= a.n_valid && (P2OUT || (a.ldy & 3) == 0);
        char *yrow = static_cast<char *>(a.y) + (size_t)(m0 + wm * 64 + li) * a.ldy * 4;
        auto body = [&](auto fullc) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int c4 = (n16 + wn * 4 + n) * 16 + kq * 4;
                const f32x4 bias = *reinterpret_cast<const f32x4 *>(bl_ + min(c4, kGemmBiasMax - 4));
                char *ycol = yrow + (P2OUT ? p2_channel_bytes(c4) : (size_t)c4 * 4);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float vr = apply_act(acc[m][n][r] + acc2[m][n][r] * (1.0f / kF16x2Scale) + bias[r], ACT);
                        v[r] = vr;
                        rmax = max(rmax, __builtin_bit_cast(unsigned, vr) & 0x7fffffffu);
                    }
                    acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    char *d = ycol + (size_t)m * 16 * a.ldy * 4;
                    if (FULL || m0 + wm * 64 + m * 16 + li < a.M) {
                        if constexpr (P2OUT) {
                            if (FULL || c4 < a.n_valid) {       // (n_valid % 32 == 0: a quad of columns is valid or not as a whole)
                                u32x2 hh, ll;
                                split2_quad(v, hh, ll);
                                reinterpret_cast<u32x2 *>(d)[0] = hh; reinterpret_cast<u32x2 *>(d)[8] = ll;
                            }
                        } else if (FULL || (c4 + 3 < a.n_valid && (a.ldy & 3) == 0)) {
                            *reinterpret_cast<f32x4 *>(d) = v;
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (c4 + k < a.n_valid) reinterpret_cast<float *>(d)[k] = v[k];
                        }
                    }
                }
            }
        };
        if (full) body(std::true_type{});
        else body(std::false_type{});
        return full;
    };
    bool pending = false;                             // the tile whose last stage was just multiplied still has its epilogue to run

    auto stage = [&](int buf) __attribute__((always_inline)) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (pending) { stores_young = epilogue(); pending = false; }
        // fragment reads first, the six DMA requests behind them (in the shadow of the reads' latency)
        const u32x4 *S = lds + buf * kGemmStageU;
        u32x4 bh[4], bl[4], ah[4], al[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { bh[n] = S[b_u + n * 128]; bl[n] = S[b_u + n * 128 + 64]; }
#pragma unroll
        for (int m = 0; m < 4; ++m) { ah[m] = S[a_h + m * 128]; al[m] = S[a_l + m * 128]; }
        {
            const int nb_ = buf == 0 ? 2 : buf - 1;
            issue(nb_);
        }
        // the pieces of stage g + 1: all but the 6 just requested - and, behind a full tile's epilogue, its NST stores, which
        // are younger than the pieces of stage g + 1 and older than those of g + 2 (so the NEXT stage's wait covers them)
        // (the six pieces requested in THIS memory phase are the only ones younger than the stage waited for, plus a full epilogue's
        //  NST stores)
        if (stores_young) {
            if constexpr (P2OUT) {
                asm volatile("s_waitcnt vmcnt(38) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(22) lgkmcnt(0)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        }
        stores_young = false;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int n = 0; n < 4; ++n) acc2[m][n] = mfma16_f16(bh[n], al[m], acc2[m][n]);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = mfma16_f16(bh[n], ah[m], acc[m][n]);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc2[m][n] = mfma16_f16(bl[n], ah[m], acc2[m][n]);
        }
        if (++c_k == nk) { c_k = 0; pending = true; }
    };

    for (int g = 0; g < total; g += 3) {
        stage(0);
        if (g + 1 < total) stage(1);
        if (g + 2 < total) stage(2);
    }
    if (wn == 0) __builtin_amdgcn_s_barrier();
    if (pending) (void)epilogue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-requested last stages: nothing may land in LDS after the workgroup has gone
    range_publish(a.range_flag, rmax, lane);
    (void)NST;
}

// Workgroups of a launch: one per CU, less where that changes nothing - every XCD walks ceil(row tiles / 8) x column tiles in
// rounds of its workgroups, and the smallest number of workgroups per XCD that needs no more ROUNDS finishes at the same time
// while leaving CUs to whatever else runs on the GPU (the other launch's convolutions): the aggregation conv of config 2 has
// 576 tiles = 2.25 rounds of 256 workgroups -> 192 workgroups x 3 tiles.
inline int gemm_f16x2_grid(int M, int N, int n_cus) {
    const int mt = (M + kGemmBM - 1) / kGemmBM, nt = N / kGemmBN;
    const int per_xcd_max = n_cus / 8 > 0 ? n_cus / 8 : 1;
    const long q = (long)((mt + 7) / 8) * nt;                         // tiles of the busiest XCD
    const long rounds = (q + per_xcd_max - 1) / per_xcd_max;
    const long per_xcd = rounds > 0 ? (q + rounds - 1) / rounds : 1;
    return (int)(8 * (per_xcd > 0 ? per_xcd : 1));
}

}  // namespace pocr
