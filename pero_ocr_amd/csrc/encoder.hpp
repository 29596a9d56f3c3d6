// encoder.hpp — self-attention encoder kernels (the "self-attention replaces BiLSTM" variant,
// BASELINE config 4).  Replaces LineSelfAttentionEncoder.forward,
// pero_ocr/ocr_engine/transformer.py:366-385: LayerNorm(E, 1e-5) -> + sinusoidal PE (:316-332) ->
// nb_layers x nn.TransformerEncoderLayer (post-norm, ReLU FFN, dropout 0, no attention mask).
// The four projections of a layer run on conv_igemm_kernel in GEMM mode; this file has the rest:
//   layernorm_kernel   y = LN(a [+ b]) * gamma + beta [+ pe[t]]      one wavefront per row
//   attention_kernel   softmax(Q K^T / sqrt(d)) V per (line, head)    one wavefront per 16 queries
//
// attention on v_mfma_f32_16x16x4_f32, no LDS, flash-style online softmax over 16-key blocks.
// It computes TRANSPOSED tiles so that no cross-lane data movement is needed between the two GEMMs:
//   S^T[key][query] = sum_d K[key][d] Q[query][d]       (A = K rows, B = Q^T; D: col = query, row = key)
//   O^T[d][query]  += sum_key V[key][d] P^T[key][query] (A = V^T, B = P^T)
// In the D layout lane (g = lane>>4, q = lane&15) holds S^T[key = 4g + r][q] in register r; the MFMA B
// operand wants lane (k = lane>>4, q) to supply row k.  Summation order over keys is free, so MFMA
// step r multiplies exactly the keys {4g + r}: register r of every lane IS the B operand of step r.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_igemm.hpp"
#include "conv_bf16x3.hpp"

namespace pocr {

// rows x E; one wave per row, E % 64 == 0 not required (strided loop).
__global__ __launch_bounds__(256) void layernorm_kernel(const float *a, const float *b, const float *gamma,
                                                        const float *beta, const float *pe, float *y, int rows,
                                                        int E, int T, float eps, const int32_t *row_t,
                                                        const int32_t *stop, void *y_p2 = nullptr, unsigned *range_max = nullptr) {
    // y_p2: a second copy of the output in the P2 (pre-split f16x2) layout - the input of the next projection GEMM
    // (gemm_f16x2.hpp); the fp32 copy stays the residual of the next LayerNorm
    const int go = stop ? *stop : 1;            // decoder steps enqueued past the end of the last batch (decoder.hpp);
                                                // tested after the row has been requested
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *pa = a + (size_t)row * E;
    const float *pb = b ? b + (size_t)row * E : nullptr;
    constexpr int MAXV = 16;                    // E <= 1024
    float v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int e = lane + 64 * k;
        float x = 0.f;
        if (e < E) { x = pa[e]; if (pb) x += pb[e]; }
        v[k] = x;
        sum += x;
    }
    if (go == 0) return;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)E;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int e = lane + 64 * k;
        const float d = e < E ? v[k] - mean : 0.f;
        sq += d * d;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = 1.0f / sqrtf(sq / (float)E + eps);
    // frame index of this row inside its line: row_t[row] for ragged batches, row % T for uniform ones
    const float *ppe = pe ? pe + (size_t)(row_t ? row_t[row] : row % T) * E : nullptr;
    float *py = y + (size_t)row * E;
    unsigned rmax = 0u;                         // f16x2 range guard of the P2 copy (conv_igemm.hpp)
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int e = lane + 64 * k;
        if (e < E) {
            float o = (v[k] - mean) * rstd * gamma[e] + beta[e];
            if (ppe) o += ppe[e];
            py[e] = o;
            range_note(rmax, o);
            if (y_p2) {
                _Float16 h, l;
                split2_scalar(o, h, l);
                _Float16 *d = reinterpret_cast<_Float16 *>(static_cast<char *>(y_p2) + (size_t)row * E * 4 + p2_channel_bytes(e));
                d[0] = h; d[32] = l;
            }
        }
    }
    range_publish(range_max, rmax, lane);
}

// qkv [n][T][3E] (q | k | v, head h = columns h*D .. h*D+D-1 of each part) -> out [n][T][E]
// P2OUT: the output in the P2 layout (it only feeds the output projection GEMM)
template <int D, bool P2OUT = false>
__global__ __launch_bounds__(64) void attention_kernel(const float *qkv, float *out, int T_uniform, int E, float scale,
                                                       const int32_t *line_T, const int32_t *row_off, unsigned *range_max = nullptr) {
    static_assert(D % 16 == 0 && D <= 128, "head dim must be a multiple of 16");
    constexpr int DG = D / 16;
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    const int qb = blockIdx.x, head = blockIdx.y, line = blockIdx.z;
    const int T = line_T ? line_T[line] : T_uniform;
    const size_t row0 = row_off ? (size_t)row_off[line] : (size_t)line * T_uniform;
    const float *base = qkv + row0 * 3 * E + head * D;
    const int q0 = qb * 16;
    if (q0 >= T) return;                    // the grid is sized for the longest line

    // B operand of S^T = K Q^T: lane (k = g, j = li) holds Q[q0 + li][16*dg + 4*g + 0..3]
    f32x4 qf[DG];
    {
        const int qi = min(q0 + li, T - 1);
        const float *qp = base + (size_t)qi * 3 * E + 4 * g;
#pragma unroll
        for (int dg = 0; dg < DG; ++dg) qf[dg] = *reinterpret_cast<const f32x4 *>(qp + 16 * dg);
    }
    f32x4 o[DG];
#pragma unroll
    for (int dt = 0; dt < DG; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < T; k0 += 16) {
        // ---- S^T tile: A operand lane (i = li (key), k = g) = K[k0 + li][16*dg + 4*g + 0..3]
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const int ki = min(k0 + li, T - 1);
            const float *kp = base + E + (size_t)ki * 3 * E + 4 * g;
#pragma unroll
            for (int dg = 0; dg < DG; ++dg) {
                const f32x4 kf = *reinterpret_cast<const f32x4 *>(kp + 16 * dg);
#pragma unroll
                for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j], qf[dg][j], s, 0, 0, 0);
            }
        }
        // ---- online softmax over keys (rows of the tile): this lane holds keys k0 + 4g + r, column q = li
        float p[4], tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = (k0 + 4 * g + r < T) ? s[r] * scale : -INFINITY;
            tmax = fmaxf(tmax, p[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);            // finite: every block holds at least one valid key
        const float alpha = expf(m_run - m_new);           // exp(-inf) = 0 on the first block
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] = expf(p[r] - m_new); psum += p[r]; }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // ---- O^T += V^T P^T: step r uses keys k0 + 4g' + r (g' = the operand lane's k index)
#pragma unroll
        for (int dt = 0; dt < DG; ++dt) {
            o[dt] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int vi = min(k0 + 4 * g + r, T - 1);          // masked keys have p = 0
                const float vf = base[2 * E + (size_t)vi * 3 * E + 16 * dt + li];   // A lane (i = li (d), k = g)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, p[r], o[dt], 0, 0, 0);
            }
        }
    }
    // ---- O^T[d = 16*dt + 4g + r][q = li] / l  ->  out[line][q0 + li][head*D + 16*dt + 4g + 0..3]
    unsigned rmax = 0u;                     // f16x2 range guard of the P2 output (conv_igemm.hpp)
    if (q0 + li < T) {
        const float inv = 1.0f / l_run;
        float *op = out + (row0 + q0 + li) * E + head * D + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DG; ++dt) {
            if constexpr (P2OUT) {
                u32x2 hh, ll;
                const f32x4 ov = o[dt] * inv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float t = ov[r]; range_note(rmax, t); }
                split2_quad(ov, hh, ll);
                u32x2 *d = reinterpret_cast<u32x2 *>(reinterpret_cast<char *>(out) + (row0 + q0 + li) * (size_t)E * 4 + p2_channel_bytes(head * D + 16 * dt + 4 * g));
                d[0] = hh; d[8] = ll;
            } else {
                const f32x4 ov = o[dt] * inv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float t = ov[r]; range_note(rmax, t); }
                *reinterpret_cast<f32x4 *>(op + 16 * dt) = ov;
            }
        }
    }
    range_publish(range_max, rmax, lane);
}

}  // namespace pocr
