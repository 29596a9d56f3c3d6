// decoder.hpp — greedy autoregressive transformer decoding with cached keys/values (SURVEY.md section 8
// row f-3).  Replaces the loop of TransformerEngineLineOCR.transcribe_batch,
// pero_ocr/ocr_engine/transformer_ocr_engine.py:49-89, and what it calls per step:
//   Decoder.infer / DecoderLayer.infer                      pero_ocr/ocr_engine/transformer.py:466-484, 413-456
//   CustomMultiheadAttention.cached_forward (self + memory)  pero_ocr/ocr_engine/transformer.py:183-303
//   dec_embeder / pos_encoder / dec_out_proj / argmax        transformer_ocr_engine.py:65-73
//
// One decoding step works on ONE new position of every line, so all of its matrix products are
// [n lines] x [K] x [N] with n <= a few hundred: weight-streaming ("skinny") GEMMs, not compute-bound
// ones, and two attention reads over cached keys/values.  Kernels of a step:
//   skinny_gemm_kernel   y = act(x W^T + b) on v_mfma_f32_16x16x4_f32; workgroup tile (16 RM) x (16 CN),
//                        its four waves split K and meet in LDS (same scheme as lstm_step_kernel)
//   dec_attention_kernel softmax((q d^-1/2) K^T) V for one (line, head): keys/values come either from the
//                        self-attention cache [pos][line][3E] or from the projected encoder output
//                        [row][2E] (ragged: line i owns rows row_off[i] .. + line_T[i])
//   s2s_init_kernel / s2s_sample_kernel   start token, arg-max sampling, "alive" bookkeeping per reference
//                        batch, embedding + positional encoding of the next input
// Lines of several reference batches are decoded together; a batch ends exactly where the reference's
// loop ends for it (all its lines have produced the boundary symbol, or the length limit), and the
// kernels skip finished batches.  `stop` points at the number of unfinished batches: every kernel
// returns at once when it is zero, so steps enqueued past the end cost only their launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_igemm.hpp"
#include "ctc.hpp"

namespace pocr {

struct SkinnyArgs {
    const float *x;          // [M][K], row stride ldx
    const float *wfrag;      // [K/16][cout16][64 lanes][4] = W[16 s + (lane & 15)][16 g + 4 (lane >> 4) + j]
    const float *bias;       // [cout16 * 16]
    float *y;                // [M][cout_valid], row stride ldy
    int64_t ldx, ldy;
    int32_t M, K, cout16, cout_valid;
    const int32_t *stop;
    // LayerNorm prologue (template LNE > 0, K == LNE): the GEMM input is x = LN(ln_a + ln_b) * gamma + beta,
    // computed by every workgroup for its own 16 rows into LDS (x is ignored); the workgroups of the first
    // column tile also store it to ln_out, where later kernels pick it up as their residual input.
    // Replaces the norm1 / norm2 / norm3 calls of DecoderLayer.infer (transformer.py:431, 443, 446).
    const float *ln_a, *ln_b, *gamma, *beta;
    float *ln_out;
    float eps;
};

template <int RM, int CN, bool RELU, int LNE = 0, int NW = 4>
__global__ __launch_bounds__(NW * 64) void skinny_gemm_kernel(SkinnyArgs a) {
    static_assert(LNE == 0 || RM == 1, "the LayerNorm prologue works on one 16-row tile");
    constexpr int XP = LNE + 4;                          // LDS row pitch of the normalised rows
    __shared__ float xs[LNE > 0 ? 16 * XP : 4];
    // the stop flag is fetched together with the first operands and tested after the K loop: one memory
    // round trip less on the critical path of every decoding step
    const int go = a.stop ? *a.stop : 1;
    __shared__ float part[NW * RM * CN * 256];           // [wave][fragment][lane][reg]; NW waves split K (8 for long K)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.y * 16 * RM, cf0 = blockIdx.x * CN;
    const int KG = a.K / 16;
    f32x4 acc[RM][CN];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int c = 0; c < CN; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float *xr[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) xr[r] = a.x + (size_t)min(row0 + 16 * r + li, a.M - 1) * a.ldx + 4 * kq;
    if constexpr (LNE > 0) {
        // one wave per row, same arithmetic as layernorm_kernel (encoder.hpp)
        constexpr int NV = LNE / 64;
        float gam[NV], bet[NV];                                    // fetched alongside the rows, not after the reductions
#pragma unroll
        for (int k = 0; k < NV; ++k) { gam[k] = a.gamma[lane + 64 * k]; bet[k] = a.beta[lane + 64 * k]; }
#pragma unroll
        for (int rr = wave; rr < 16; rr += NW) {                   // 4 rows per wave, unrolled: all 8 NV loads in flight at once
            const int row = min(row0 + rr, a.M - 1);
            const float *pa = a.ln_a + (size_t)row * LNE, *pb = a.ln_b + (size_t)row * LNE;
            float v[NV], sum = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) { v[k] = pa[lane + 64 * k] + pb[lane + 64 * k]; sum += v[k]; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
            const float mean = sum / (float)LNE;
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) { const float dlt = v[k] - mean; sq += dlt * dlt; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
            const float rstd = 1.0f / sqrtf(sq / (float)LNE + a.eps);
            const bool store = blockIdx.x == 0 && row0 + rr < a.M && go != 0;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int e = lane + 64 * k;
                const float o = (v[k] - mean) * rstd * gam[k] + bet[k];
                xs[rr * XP + e] = o;
                if (store) a.ln_out[(size_t)(row0 + rr) * LNE + e] = o;
            }
        }
        __syncthreads();
    }
    // A fragment of k-group g, row tile r: from LDS after the LayerNorm prologue, else from global memory
    auto lda = [&](int r, int g) -> f32x4 {
        if constexpr (LNE > 0) return *reinterpret_cast<const f32x4 *>(&xs[li * XP + 4 * kq + 16 * g]);
        else return *reinterpret_cast<const f32x4 *>(xr[r] + 16 * g);
    };
    const f32x4 *wf = reinterpret_cast<const f32x4 *>(a.wfrag) + lane;
    // U k-groups per iteration: the step is a chain of L2 round trips, so all operand loads of a batch are issued
    // before its first MFMA (K = 512: one batch per wave; the registers bound U)
    constexpr int U = (RM + CN <= 3) ? 8 : 4;
    // the epilogue's bias is requested now, not after the reduction
    float bias_v[(RM * CN * 64 + NW * 64 - 1) / (NW * 64)];
#pragma unroll
    for (int it = 0; it < (RM * CN * 64 + NW * 64 - 1) / (NW * 64); ++it) {
        const int item = tid + it * NW * 64;
        bias_v[it] = item < RM * CN * 64 ? a.bias[(cf0 + (item >> 6) % CN) * 16 + (item & 15)] : 0.f;
    }
    int kg = wave;
    for (; kg + NW * (U - 1) < KG; kg += NW * U) {
        f32x4 av[U][RM], bv[U][CN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = kg + NW * u;
#pragma unroll
            for (int r = 0; r < RM; ++r) av[u][r] = lda(r, g);
#pragma unroll
            for (int c = 0; c < CN; ++c) bv[u][c] = wf[((size_t)g * a.cout16 + cf0 + c) * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < RM; ++r)
#pragma unroll
                    for (int c = 0; c < CN; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r][j], bv[u][c][j], acc[r][c], 0, 0, 0);
    }
    for (; kg < KG; kg += NW) {
        f32x4 av[RM], bv[CN];
#pragma unroll
        for (int r = 0; r < RM; ++r) av[r] = lda(r, kg);
#pragma unroll
        for (int c = 0; c < CN; ++c) bv[c] = wf[((size_t)kg * a.cout16 + cf0 + c) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < RM; ++r)
#pragma unroll
                for (int c = 0; c < CN; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][j], bv[c][j], acc[r][c], 0, 0, 0);
    }
    if (go == 0) return;
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int c = 0; c < CN; ++c)
            *reinterpret_cast<f32x4 *>(&part[((wave * RM * CN + r * CN + c) * 64 + lane) * 4]) = acc[r][c];
    __syncthreads();
    // D layout of a 16x16 fragment: lane -> column lane & 15, rows 4 (lane >> 4) + reg
#pragma unroll
    for (int it = 0; it < (RM * CN * 64 + NW * 64 - 1) / (NW * 64); ++it) {
        const int item = tid + it * NW * 64;
        if (item >= RM * CN * 64) break;
        const int f = item >> 6, ln = item & 63;
        f32x4 v = *reinterpret_cast<const f32x4 *>(&part[((0 * RM * CN + f) * 64 + ln) * 4]);
#pragma unroll
        for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4 *>(&part[((w * RM * CN + f) * 64 + ln) * 4]);
        const int r = f / CN, c = f % CN;
        const int col = (cf0 + c) * 16 + (ln & 15);
        const float b = bias_v[it];
        if (col < a.cout_valid) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = row0 + 16 * r + 4 * (ln >> 4) + q;
                float o = v[q] + b;
                if (RELU) o = fmaxf(o, 0.f);
                if (row < a.M) a.y[(size_t)row * a.ldy + col] = o;
            }
        }
    }
}

constexpr int DEC_MAX_KEYS = 1024;       // keys one query can attend to (T <= 960 frames; decoded length <= 961)

struct DecAttnArgs {
    const float *q;          // query of line i, head h: q + i * ldq + h * D
    const float *k, *v;      // key / value of line i, position p, head h: base + line_base(i) + p * pos_stride + h * D
    float *out;              // [n][E]
    int64_t ldq, pos_stride, line_stride;
    const int32_t *row_off;  // memory attention: line_base = row_off[i] * pos_stride; NULL: line_base = i * line_stride
    const int32_t *line_len; // keys per line (memory attention); NULL: `len` for every line
    const int32_t *line_done;                    // lines of finished batches are skipped
    const int32_t *stop;
    int32_t len, E;
    float scale;
};

// MEMORY only names the instantiation (keys / values = the encoder output of the line, row_off set) so that a kernel
// trace tells the HBM-bound memory attention from the short self attention over the cache; the code is the same.
template <int D, bool MEMORY = false>
__global__ __launch_bounds__(256) void dec_attention_kernel(DecAttnArgs a) {
    static_assert(D == 32 || D == 64 || D == 128, "head dim");
    // A key / value row of one head is D floats = LPR lanes x 16 bytes; a wave instruction therefore fetches RPW whole
    // rows, fully coalesced (one lane per 16 bytes instead of one lane per row).
    constexpr int LPR = D / 4, RPW = 64 / LPR;
    const int head = blockIdx.x, line = blockIdx.y;
    const int go = a.stop ? *a.stop : 1, done = a.line_done[line];      // fetched alongside q
    __shared__ float sc[DEC_MAX_KEYS];
    __shared__ float red[8];
    __shared__ f32x4 opart[4][LPR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane % LPR, g = lane / LPR;
    const int S = a.line_len ? a.line_len[line] : a.len;
    const size_t base = (a.row_off ? (size_t)a.row_off[line] * a.pos_stride : (size_t)line * a.line_stride) + (size_t)head * D + 4 * li;
    f32x4 qv = *reinterpret_cast<const f32x4 *>(a.q + (size_t)line * a.ldq + head * D + 4 * li);
    if (go == 0 || done) return;
    qv *= a.scale;                                                      // q * d^-1/2 first (transformer.py:268)
    // ---- scores: wave w takes the keys p = (w + 4 i) * RPW + g
    constexpr int US = 4;
    for (int p0 = wave * RPW; p0 < S; p0 += 4 * RPW * US) {
        f32x4 kv[US];
#pragma unroll
        for (int u = 0; u < US; ++u) {
            const int p = p0 + 4 * RPW * u + g;
            kv[u] = *reinterpret_cast<const f32x4 *>(a.k + base + (size_t)min(p, S - 1) * a.pos_stride);
        }
#pragma unroll
        for (int u = 0; u < US; ++u) {
            const int p = p0 + 4 * RPW * u + g;
            float dot = kv[u][0] * qv[0] + kv[u][1] * qv[1] + kv[u][2] * qv[2] + kv[u][3] * qv[3];
#pragma unroll
            for (int off = 1; off < LPR; off <<= 1) dot += __shfl_xor(dot, off, 64);
            if (li == 0 && p < S) sc[p] = dot;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int p = tid; p < S; p += 256) m = fmaxf(m, sc[p]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int p = tid; p < S; p += 256) {
        const float e = expf(sc[p] - m);
        sc[p] = e;
        sum += e;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    // ---- weighted sum of the values, same key -> lane mapping; a lane accumulates its 4 outputs
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int UV = 8;
    for (int p0 = wave * RPW; p0 < S; p0 += 4 * RPW * UV) {
        f32x4 vv[UV];
        float pr[UV];
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            const int p = p0 + 4 * RPW * u + g;
            pr[u] = p < S ? sc[p] : 0.f;
            vv[u] = *reinterpret_cast<const f32x4 *>(a.v + base + (size_t)min(p, S - 1) * a.pos_stride);
        }
#pragma unroll
        for (int u = 0; u < UV; ++u) acc += vv[u] * pr[u];
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        acc[0] += __shfl_xor(acc[0], off, 64); acc[1] += __shfl_xor(acc[1], off, 64);
        acc[2] += __shfl_xor(acc[2], off, 64); acc[3] += __shfl_xor(acc[3], off, 64);
    }
    if (g == 0) opart[wave][li] = acc;
    __syncthreads();
    if (tid < LPR) {
        const f32x4 o = (opart[0][tid] + opart[1][tid] + opart[2][tid] + opart[3][tid]) / sum;
        *reinterpret_cast<f32x4 *>(a.out + (size_t)line * a.E + head * D + 4 * tid) = o;
    }
}

struct S2sState {
    int32_t *tokens;          // [n][S_cap] sample of every step
    int32_t *alive;           // [n]   line has not produced the boundary symbol yet (transformer_ocr_engine.py:72-73)
    int32_t *batch_done;      // [n_batches]
    int32_t *line_done;       // [n]   the line's batch is finished
    int32_t *steps;           // [n_batches] decoding steps the reference's loop runs for this batch (= rows of its logits)
    int32_t *remaining;       // unfinished batches
    const int32_t *batch_first;   // [n_batches + 1] first line of every batch
    const int32_t *limit;     // [n_batches] padded width / 4 (transformer_ocr_engine.py:77)
    const float *embed;       // [C][E]
    const float *pe;          // [>= S_cap + 1][E]
    float *x;                 // [n][E] decoder input of the next step
    int32_t n, n_batches, S_cap, C, E, boundary;
};

__global__ __launch_bounds__(256) void s2s_init_kernel(S2sState st) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *st.remaining = st.n_batches;
    if (i < st.n_batches) { st.batch_done[i] = 0; st.steps[i] = 0; }
    if (i < st.n) { st.alive[i] = 1; st.line_done[i] = 0; }
    for (size_t k = i; k < (size_t)st.n * st.E; k += (size_t)gridDim.x * 256) {
        const int e = (int)(k % st.E);
        st.x[k] = st.embed[(size_t)st.boundary * st.E + e] + st.pe[e];       // start token = boundary symbol, position 0
    }
}

// One workgroup per reference batch.  logits_step = logits + s * C, row stride ld (= S_cap * C).
__global__ __launch_bounds__(256) void s2s_sample_kernel(S2sState st, const float *logits_step, int64_t ld, int s) {
    if (*st.remaining == 0) return;
    const int b = blockIdx.x;
    if (st.batch_done[b]) return;
    __shared__ int any_alive[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int any = 0;
    for (int line = st.batch_first[b] + wave; line < st.batch_first[b + 1]; line += 4) {
        const float *row = logits_step + (size_t)line * ld;
        float bv = 0.f;
        int bi = 0x7fffffff;
        bool have = false;
        for (int c = lane; c < st.C; c += 64) {
            const float v = row[c];
            if (!have || argmax_better(v, c, bv, bi)) { bv = v; bi = c; have = true; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            const bool oh = __shfl_xor((int)have, off, 64) != 0;
            if (oh && (!have || argmax_better(ov, oi, bv, bi))) { bv = ov; bi = oi; have = true; }
        }
        const int al = st.alive[line] && bi != st.boundary;
        if (lane == 0) {
            st.tokens[(size_t)line * st.S_cap + s] = bi;
            st.alive[line] = al;
        }
        any |= al;
        const float *em = st.embed + (size_t)bi * st.E, *pe = st.pe + (size_t)(s + 1) * st.E;
        float *x = st.x + (size_t)line * st.E;
        for (int e = lane; e < st.E; e += 64) x[e] = em[e] + pe[e];
    }
    if (lane == 0) any_alive[wave] = any;
    __syncthreads();
    const bool alive = any_alive[0] | any_alive[1] | any_alive[2] | any_alive[3];
    if (!alive || s + 1 > st.limit[b]) {              // transformer_ocr_engine.py:74-80 (len(partial_transcripts) = s + 1)
        for (int line = st.batch_first[b] + tid; line < st.batch_first[b + 1]; line += 256) st.line_done[line] = 1;
        if (tid == 0) {
            st.batch_done[b] = 1;
            st.steps[b] = s + 1;
            atomicSub(st.remaining, 1);
        }
    }
}

}  // namespace pocr
