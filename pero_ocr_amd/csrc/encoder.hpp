// encoder.hpp — self-attention encoder kernels (the "self-attention replaces BiLSTM" variant,
// BASELINE config 4).  Replaces LineSelfAttentionEncoder.forward,
// pero_ocr/ocr_engine/transformer.py:366-385: LayerNorm(E, 1e-5) -> + sinusoidal PE (:316-332) ->
// nb_layers x nn.TransformerEncoderLayer (post-norm, ReLU FFN, dropout 0, no attention mask).
// The four projections of a layer run on conv_igemm_kernel in GEMM mode; this file has the rest:
//   layernorm_kernel   y = LN(a [+ b]) * gamma + beta [+ pe[t]]      one wavefront per row
//   attention_kernel   softmax(Q K^T / sqrt(d)) V per (line, head)    one wavefront per 16 queries (fp32 MFMA, fp32 q | k | v)
//   attention_f16x2_kernel  the same on f16x2 MFMAs from P2 q | k | v      four wavefronts per 128 queries, K / V through LDS
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
    if ((E & 511) == 0 && E <= 1024) {
        // E = 512 / 1024 (the encoder's widths): a lane owns whole channel OCTETS - 32-byte loads, and the P2 copy is written as the
        // 16-byte h / l units the GEMM reads instead of two 2-byte stores per value (the kernel ran at 26 % of the HBM rate on c4:
        // 0.8 ms per step for five passes over 53 248 rows, profiles/r04_bench_c4_kernel_stats.txt)
        const int NV = E >> 9;
        f32x4 v0[2], v1[2];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= NV) break;
            const int e = (k * 64 + lane) * 8;
            v0[k] = *reinterpret_cast<const f32x4 *>(pa + e); v1[k] = *reinterpret_cast<const f32x4 *>(pa + e + 4);
            if (pb) { v0[k] += *reinterpret_cast<const f32x4 *>(pb + e); v1[k] += *reinterpret_cast<const f32x4 *>(pb + e + 4); }
#pragma unroll
            for (int c = 0; c < 4; ++c) sum += v0[k][c];
#pragma unroll
            for (int c = 0; c < 4; ++c) sum += v1[k][c];
        }
        if (go == 0) return;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float mean = sum / (float)E;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= NV) break;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float d0 = v0[k][c] - mean, d1 = v1[k][c] - mean; sq += d0 * d0; sq += d1 * d1; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
        const float rstd = 1.0f / sqrtf(sq / (float)E + eps);
        const float *ppe = pe ? pe + (size_t)(row_t ? row_t[row] : row % T) * E : nullptr;
        float *py = y + (size_t)row * E;
        unsigned rmax = 0u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= NV) break;
            const int e = (k * 64 + lane) * 8;
            const f32x4 g0 = *reinterpret_cast<const f32x4 *>(gamma + e), g1 = *reinterpret_cast<const f32x4 *>(gamma + e + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(beta + e), b1 = *reinterpret_cast<const f32x4 *>(beta + e + 4);
            f32x4 o0, o1;
#pragma unroll
            for (int c = 0; c < 4; ++c) { o0[c] = (v0[k][c] - mean) * rstd * g0[c] + b0[c]; o1[c] = (v1[k][c] - mean) * rstd * g1[c] + b1[c]; }
            if (ppe) { o0 += *reinterpret_cast<const f32x4 *>(ppe + e); o1 += *reinterpret_cast<const f32x4 *>(ppe + e + 4); }
            *reinterpret_cast<f32x4 *>(py + e) = o0; *reinterpret_cast<f32x4 *>(py + e + 4) = o1;
#pragma unroll
            for (int c = 0; c < 4; ++c) { range_note(rmax, o0[c]); range_note(rmax, o1[c]); }
            if (y_p2) {
                u32x2 h0, l0, h1, l1;
                split2_quad(o0, h0, l0);
                split2_quad(o1, h1, l1);
                u32x4 *d = reinterpret_cast<u32x4 *>(static_cast<char *>(y_p2) + (size_t)row * E * 4 + p2_channel_bytes(e));
                d[0] = (u32x4){h0[0], h0[1], h1[0], h1[1]};
                d[4] = (u32x4){l0[0], l0[1], l1[0], l1[1]};
            }
        }
        range_publish(range_max, rmax, lane);
        return;
    }
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

// ---- attention in the f16x2 arithmetic of the rest of the stack (conv_bf16x3.hpp): q | k | v arrive in the P2 layout (the
// projection GEMM's P2 output), both products run on v_mfma_f32_16x16x32_f16 - three MFMAs per 32-deep block instead of
// eight fp32 ones - and the probabilities are split in registers.  One workgroup = 128 queries of one (line, head): four
// waves, two 16-query tiles each, so every K / V fragment read from LDS feeds six MFMAs.  Keys go by blocks of 32:
//   S^T = K Q^T      A = K rows from the LDS image [chunk][plane][key][4 units], B = Q fragments held in registers
//   O^T += V^T P^T   A = V^T from the LDS image [plane][d][32 keys], B = P^T = the S^T accumulators themselves
// The reduction index of the second product is free, so position 8 kq + j of a V^T row holds key 16 (j >> 2) + 4 kq + (j & 3):
// then registers {S^T tile 0: r = 0..3, S^T tile 1: r = 0..3} of a lane ARE its eight B-operand values, no data movement.
// V is transposed on its way into LDS: a lane takes the same 8 channels of two consecutive keys (two 16-byte loads) and
// stores eight (key, key + 1) f16 pairs.  The 16-byte units of a 64-byte image row are XORed with 3 on rows 8..15 of a
// tile, which spreads the four non-contiguous 16-lane groups of ds_read_b128 over all banks.  Next block's K / V pieces
// are requested before the current block's products (register staging), one barrier per block.
// exp runs as v_exp_f32 on log2(e)-scaled scores; the running maximum only rescales the accumulators when it moved.
template <int D, int QT>
__global__ __launch_bounds__(256, 2) void attention_f16x2_kernel(const void *qkv, void *out, int T_uniform, int E, float scale2,
                                                                 const int32_t *line_T, const int32_t *row_off, unsigned *range_max) {
    static_assert(D % 32 == 0 && D <= 128, "head dim must be a multiple of 32");
    constexpr int DC = D / 32, DT = D / 16, KB = 32;        // QT: 16-query tiles per wave (2; 1 for D = 128, registers)
    constexpr int K_IMG = KB * 64 + 64;          // one (chunk, plane) image; + 64 B: the planes' 16-byte stores on different banks
    constexpr int K_BYTES = DC * 2 * K_IMG;
    constexpr int V_IMG = D * 64 + 64;
    constexpr int BUF = K_BYTES + 2 * V_IMG;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int head = blockIdx.y, line = blockIdx.z;
    const int T = line_T ? line_T[line] : T_uniform;
    const int qw = blockIdx.x * (4 * QT * 16);
    if (qw >= T) return;                         // the grid is sized for the longest line (uniform for the workgroup)
    const size_t row0 = row_off ? (size_t)row_off[line] : (size_t)line * T_uniform;
    const size_t pitch = (size_t)E * 12;          // bytes of a q | k | v row
    const char *base = static_cast<const char *>(qkv) + row0 * pitch;
    const size_t q_off = (size_t)(head * D / 32) * 128, k_off = q_off + (size_t)E * 4, v_off = q_off + (size_t)E * 8;
    const int q0 = qw + wave * (QT * 16);
    const bool active = q0 < T;

    u32x4 qh[QT][DC], ql[QT][DC];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const char *qp = base + (size_t)min(q0 + 16 * t + li, T - 1) * pitch + q_off + kq * 16;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            qh[t][c] = *reinterpret_cast<const u32x4 *>(qp + c * 128);
            ql[t][c] = *reinterpret_cast<const u32x4 *>(qp + c * 128 + 64);
        }
    }
    // staging assignment: K - DC 16-byte pieces per thread; V - key pairs, (128 DC) items over 256 threads
    constexpr int VI = (128 * DC + 255) / 256;
    u32x4 kreg[DC], vreg[VI][2];
    auto request = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DC; ++i) {
            const int p = tid + 256 * i, key = p / (DC * 8), w = p % (DC * 8);
            kreg[i] = *reinterpret_cast<const u32x4 *>(base + (size_t)min(k0 + key, T - 1) * pitch + k_off + (w >> 3) * 128 + (w & 7) * 16);
        }
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int it = tid + 256 * i;
            if (128 * DC >= 256 || it < 128 * DC) {
                const int m = it & 15, cb = it >> 4;
                const size_t o = v_off + (size_t)(cb >> 3) * 128 + (size_t)(cb & 7) * 16;
                vreg[i][0] = *reinterpret_cast<const u32x4 *>(base + (size_t)min(k0 + 2 * m, T - 1) * pitch + o);
                vreg[i][1] = *reinterpret_cast<const u32x4 *>(base + (size_t)min(k0 + 2 * m + 1, T - 1) * pitch + o);
            }
        }
    };
    auto deposit = [&](char *buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DC; ++i) {
            const int p = tid + 256 * i, key = p / (DC * 8), w = p % (DC * 8), u = w & 3;
            *reinterpret_cast<u32x4 *>(buf + (w >> 2) * K_IMG + key * 64 + ((u ^ ((key & 8) ? 3 : 0)) * 16)) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int it = tid + 256 * i;
            if (128 * DC >= 256 || it < 128 * DC) {
                const int m = it & 15, cb = it >> 4, kq8 = cb & 3, plane = (cb >> 2) & 1, c = cb >> 3;
                const int unit = ((m >> 1) & 3) ^ ((kq8 & 1) ? 3 : 0);
                char *d = buf + K_BYTES + plane * V_IMG + (c * 32 + kq8 * 8) * 64 + unit * 16 + (m >> 3) * 8 + (m & 1) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {     // dword j of the two pieces = channels 2 j, 2 j + 1 of keys 2 m, 2 m + 1
                    const unsigned a = vreg[i][0][j], b = vreg[i][1][j];
                    *reinterpret_cast<unsigned *>(d + (2 * j) * 64) = (a & 0xffffu) | (b << 16);
                    *reinterpret_cast<unsigned *>(d + (2 * j + 1) * 64) = (a >> 16) | (b & 0xffff0000u);
                }
            }
        }
    };

    f32x4 o1[QT][DT], o2[QT][DT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { o1[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; o2[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) { m_run[t] = -INFINITY; l_run[t] = 0.f; }
    const int swz = (li & 8) ? 3 : 0;
    const int frag = li * 64 + ((kq ^ swz) * 16);         // this lane's 16 bytes inside a 16-row tile of either image

    request(0);
    deposit(smem);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < T; k0 += KB) {
        const bool more = k0 + KB < T;
        if (more) request(k0 + KB);
        const char *buf = smem + cur * BUF;
        if (active) {
            // ---- S^T, two key tiles x QT query tiles
            f32x4 s1[QT][2], s2[QT][2];
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) { s1[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; s2[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    const u32x4 kh = *reinterpret_cast<const u32x4 *>(buf + (2 * c) * K_IMG + kt * 1024 + frag);
                    const u32x4 kl = *reinterpret_cast<const u32x4 *>(buf + (2 * c + 1) * K_IMG + kt * 1024 + frag);
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        s1[t][kt] = mfma16_f16(kh, qh[t][c], s1[t][kt]);
                        s2[t][kt] = mfma16_f16(kh, ql[t][c], s2[t][kt]);
                        s2[t][kt] = mfma16_f16(kl, qh[t][c], s2[t][kt]);
                    }
                }
            // ---- online softmax: this lane holds keys k0 + 16 kt + 4 kq + r of query li
            u32x4 ph[QT], pl[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float p[8], tmax = -INFINITY;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = (s1[t][j >> 2][j & 3] + s2[t][j >> 2][j & 3] * (1.0f / kF16x2Scale)) * scale2;
                    p[j] = (more || k0 + 16 * (j >> 2) + 4 * kq + (j & 3) < T) ? v : -INFINITY;
                    tmax = fmaxf(tmax, p[j]);
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run[t], tmax);         // finite: every block holds at least one valid key
                const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);     // 0 on the first block
                float psum = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { p[j] = __builtin_amdgcn_exp2f(p[j] - m_new); psum += p[j]; }
                psum += __shfl_xor(psum, 16, 64);
                psum += __shfl_xor(psum, 32, 64);
                l_run[t] = l_run[t] * alpha + psum;
                m_run[t] = m_new;
                if (__any(alpha != 1.0f)) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) { o1[t][dt] *= alpha; o2[t][dt] *= alpha; }
                }
                u32x2 h0, l0, h1, l1;
                split2_quad((f32x4){p[0], p[1], p[2], p[3]}, h0, l0);
                split2_quad((f32x4){p[4], p[5], p[6], p[7]}, h1, l1);
                ph[t] = (u32x4){h0[0], h0[1], h1[0], h1[1]};
                pl[t] = (u32x4){l0[0], l0[1], l1[0], l1[1]};
            }
            // ---- O^T += V^T P^T
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const u32x4 vh = *reinterpret_cast<const u32x4 *>(buf + K_BYTES + dt * 1024 + frag);
                const u32x4 vl = *reinterpret_cast<const u32x4 *>(buf + K_BYTES + V_IMG + dt * 1024 + frag);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    o1[t][dt] = mfma16_f16(vh, ph[t], o1[t][dt]);
                    o2[t][dt] = mfma16_f16(vh, pl[t], o2[t][dt]);
                    o2[t][dt] = mfma16_f16(vl, ph[t], o2[t][dt]);
                }
            }
        }
        if (more) deposit(smem + (cur ^ 1) * BUF);
        __syncthreads();
        cur ^= 1;
    }
    // ---- O^T[d = 16 dt + 4 kq + r][q = li] / l  ->  out (P2) [row q][head * D + 16 dt + 4 kq + 0..3]
    unsigned rmax = 0u;
    if (active) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int q = q0 + 16 * t + li;
            if (q < T) {
                const float inv = 1.0f / l_run[t];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const f32x4 ov = (o1[t][dt] + o2[t][dt] * (1.0f / kF16x2Scale)) * inv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float x = ov[r]; range_note(rmax, x); }
                    u32x2 hh, ll;
                    split2_quad(ov, hh, ll);
                    u32x2 *d = reinterpret_cast<u32x2 *>(static_cast<char *>(out) + (row0 + q) * (size_t)E * 4 + p2_channel_bytes(head * D + 16 * dt + 4 * kq));
                    d[0] = hh; d[8] = ll;
                }
            }
        }
    }
    range_publish(range_max, rmax, lane);
}

}  // namespace pocr
