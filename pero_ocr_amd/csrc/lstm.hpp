// lstm.hpp — one time step of a bidirectional LSTM layer (torch.nn.LSTM semantics: gate
// order i,f,g,o; gates = W_ih x + b_ih + W_hh h + b_hh; c' = f*c + i*g; h' = o*tanh(c')).
// The reference runs it inside its opaque TorchScript model (pytorch_ocr_engine.py:66-69;
// aten::lstm / mkldnn_rnn_layer in the CPU profile).
//
// The input projections W_ih x + (b_ih + b_hh) for ALL time steps are hoisted into one
// MFMA GEMM (conv_igemm_kernel as a 1x1 conv) -> xproj [n][T][2*4H] (dir-major).
// This kernel does the serial part for step s: fwd direction at t = s, bwd at t = T-1-s.
//
// Work split: grid = (H/16 unit groups, ceil(n/16) line slices, 2 directions); a workgroup
// computes the 4 gates of 16 hidden units for 16 lines: a [16 lines] x [4 x 16 gate columns]
// x [K = H] GEMM on v_mfma_f32_16x16x4_f32.  Its 4 waves split K; partial sums meet in LDS,
// then thread (line, unit) applies the gate non-linearities.  h ping-pongs between two
// HBM buffers (kernel boundary = the step barrier); W_hh is read from L2 in fragment order
//   whh_frag[dir][unit_group][k/16][gate][lane][j] = W_hh[gate*H + 16*ug + (lane&15)][16*kg + 4*(lane>>4) + j].
// The step is latency-bound (a chain of L2 round trips), so every global load of the step -
// h fragments, W_hh fragments, the xproj gate pre-activations and c - is issued up front
// (KPW = k-groups per wave is a template parameter so the loop unrolls).
#pragma once
#include <hip/hip_runtime.h>
#include "conv_igemm.hpp"
#include "conv_bf16x3.hpp"

namespace pocr {

struct LstmStepArgs {
    const float *xproj;     // [n][T][8H]   (dir, gate, unit)
    const float *whh_frag;  // [2][H/16][H/16][4][64][4]
    const void *whh2;       // f16x2 fragments (lstm_gemm_f16x2 below) or NULL: the recurrent GEMM on the fp32 MFMA
    const float *h_in;      // [2][npad][H]
    float *h_out;           // [2][npad][H]
    float *c;               // [2][npad][H]  (in place)
    float *y;               // [n][T][2H]    layer output, fwd in [0,H), bwd in [H,2H)
    const int32_t *dims;    // optional device pointer to {n, npad}: overrides the two fields below, so that a
                            // captured hipGraph of the T step launches can be replayed for any chunk size
    // ragged batches: per-line frame counts and first rows (frames of line i are rows row_off[i] ..
    // row_off[i] + line_T[i] of xproj / y); slice_T[s] = max line_T over the 16 lines of slice s.
    // NULL -> every line has T frames and line i starts at row i * T.
    const int32_t *line_T;
    const int32_t *row_off;
    const int32_t *slice_T;
    int32_t n, npad, T, H, step;
    int32_t y_p2;           // the layer output in the P2 (pre-split f16x2) layout: it only feeds the next layer's input projection (gemm_f16x2.hpp)
};

// Gate non-linearities on the hardware transcendentals (v_exp_f32, v_rcp_f32: ~1 ulp each): absolute error ~1e-7, the
// size of one fp32 rounding of a value in (-1, 1) - the library expf / tanhf they replace were 15 % of a recurrence step
// (1229 of 8456 cycles, lstm_resident.hpp).  tanh(x) = 1 - 2 / (1 + e^2x) saturates to +-1 exactly for large |x|.
// Every operation is spelled out (no fp contraction left to the compiler): lstm_step_kernel and lstm_resident_kernel must
// produce the same bits.
#pragma clang fp contract(off)
__device__ __forceinline__ float sigmoid_f32(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f32(float x) { return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)), 1.0f); }
// gates i, f, g, o (pre-activations: recurrent part + input projection) -> new cell and hidden state
__device__ __forceinline__ void lstm_cell(const float (&gate)[4], const float (&xg)[4], float cprev, float &cn, float &hn) {
    const float gi = sigmoid_f32(gate[0] + xg[0]);
    const float gf = sigmoid_f32(gate[1] + xg[1]);
    const float gg = tanh_f32(gate[2] + xg[2]);
    const float go = sigmoid_f32(gate[3] + xg[3]);
    cn = __builtin_fmaf(gf, cprev, gi * gg);
    hn = go * tanh_f32(cn);
}
#pragma clang fp contract(fast)

// ---- the recurrent GEMM on the f16 matrix pipe (conv_bf16x3.hpp "f16x2": operands as two f16 planes, three MFMAs per
// 32-deep block): 24 v_mfma_f32_16x16x32_f16 per wave and step at H = 256 instead of 64 v_mfma_f32_16x16x4_f32 (384 against
// 2048 cycles of a step's ~7600).  W_hh is split on the host:
//   whh2[dir][unit group][K block = k / 32][gate][plane][lane][8 x f16] = plane of W_hh[gate*H + 16*ug + (lane&15)][32*blk + 8*(lane>>4) + j];
// h_{s-1} is split by the wave that reads it (8 values per lane and block).  K blocks are dealt to the four waves round-robin
// (wave w: blocks w, w + 4, ...); both recurrence kernels call these two functions, so their results agree bit for bit.
template <int KPW> struct LstmW2 { u32x4 b[(2 * KPW + 3) / 4][4][2]; };

template <int KPW>
__device__ __forceinline__ void lstm_load_w_f16x2(const void *whh2, int dir, int ug, int wave, int lane, LstmW2<KPW> &w) {
    constexpr int NB = 2 * KPW, QB = (NB + 3) / 4, KGT = 4 * KPW;
    const u32x4 *base = reinterpret_cast<const u32x4 *>(whh2) + ((size_t)(dir * KGT + ug) * NB) * 4 * 2 * 64 + lane;
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int blk = wave + 4 * q;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                w.b[q][g][p] = blk < NB ? base[((size_t)(blk * 4 + g) * 2 + p) * 64] : (u32x4){0u, 0u, 0u, 0u};
    }
}

// The MFMA part: x0 / x1 = the lane's 8 values of h_{s-1} per K block (wave w: blocks w, w + 4, ...), split here.  acc = this wave's partial gates.
template <int KPW>
__device__ __forceinline__ void lstm_mfma_f16x2(const f32x4 (&x0)[(2 * KPW + 3) / 4], const f32x4 (&x1)[(2 * KPW + 3) / 4], int wave,
                                                const LstmW2<KPW> &w, f32x4 (&acc)[4]) {
    constexpr int NB = 2 * KPW, QB = (NB + 3) / 4;
    f32x4 main_[4], cross[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { main_[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; cross[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        if (wave + 4 * q >= NB) continue;                 // (wave-uniform: H = 64 has two blocks for four waves)
        u32x2 h01, l01, h23, l23;
        split2_quad(x0[q], h01, l01);
        split2_quad(x1[q], h23, l23);
        const u32x4 ah = {h01[0], h01[1], h23[0], h23[1]}, al = {l01[0], l01[1], l23[0], l23[1]};
#pragma unroll
        for (int g = 0; g < 4; ++g) cross[g] = mfma16_f16(al, w.b[q][g][0], cross[g]);
#pragma unroll
        for (int g = 0; g < 4; ++g) main_[g] = mfma16_f16(ah, w.b[q][g][0], main_[g]);
#pragma unroll
        for (int g = 0; g < 4; ++g) cross[g] = mfma16_f16(ah, w.b[q][g][1], cross[g]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = main_[g] + cross[g] * (1.0f / kF16x2Scale);
}

// hrow: the lane's line of h_{s-1} (H floats) in global memory; NT: read with L1-bypassing loads (resident kernel).
template <int KPW, bool NT>
__device__ __forceinline__ void lstm_gemm_f16x2(const float *hrow, int wave, int kq, const LstmW2<KPW> &w, f32x4 (&acc)[4]) {
    constexpr int NB = 2 * KPW, QB = (NB + 3) / 4;
    f32x4 x0[QB], x1[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int blk = min(wave + 4 * q, NB - 1);
        const f32x4 *p = reinterpret_cast<const f32x4 *>(hrow + blk * 32 + kq * 8);
        if constexpr (NT) { x0[q] = __builtin_nontemporal_load(p); x1[q] = __builtin_nontemporal_load(p + 1); }
        else { x0[q] = p[0]; x1[q] = p[1]; }
    }
    lstm_mfma_f16x2<KPW>(x0, x1, wave, w, acc);
}

// layer output y[row][2H]: fp32, or split once here into the two f16 planes the next projection's MFMAs read (P2, conv_bf16x3.hpp)
__device__ __forceinline__ void lstm_store_y(float *y, size_t row, int H2, int col, float hn, bool p2) {
    if (p2) {
        _Float16 h, l;
        split2_scalar(hn, h, l);
        _Float16 *d = reinterpret_cast<_Float16 *>(reinterpret_cast<char *>(y) + row * (size_t)H2 * 4 + p2_channel_bytes(col));
        d[0] = h; d[32] = l;
    } else {
        y[row * (size_t)H2 + col] = hn;
    }
}

// KPW > 0: H == 64 * KPW, fully unrolled.  KPW == 0: generic H (multiple of 16).
template <int KPW>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepArgs a) {
    __shared__ float part[4 * 4 * 64 * 4];      // [wave][gate][lane][reg]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int ug = blockIdx.x, slice = blockIdx.y, dir = blockIdx.z;
    if (a.dims) { a.n = a.dims[0]; a.npad = a.dims[1]; }
    if (slice * 16 >= a.npad) return;           // replayed graphs are sized for a bucket of slices
    if (a.slice_T && a.step >= a.slice_T[slice]) return;      // every line of this slice has finished
    const int H = KPW > 0 ? 64 * KPW : a.H, KGT = H / 16;

    // epilogue operands first: they come from HBM (xproj is streamed, never cached)
    const int u = tid & 15, i = tid >> 4;
    const int line = slice * 16 + i;
    const int unit = ug * 16 + u;
    const size_t sidx = ((size_t)dir * a.npad + line) * H + unit;
    float xg[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    const int Ti = line < a.n ? (a.line_T ? a.line_T[line] : a.T) : 0;
    const bool live = a.step < Ti;                              // shorter lines simply stop updating
    const int t = dir == 0 ? a.step : Ti - 1 - a.step;         // the backward pass starts at the line's own last frame
    const size_t row = (a.row_off ? (size_t)a.row_off[min(line, a.n - 1)] : (size_t)line * a.T) + t;
    if (live) {
        const float *xp = a.xproj + row * (8 * H) + (size_t)dir * 4 * H + unit;
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[g] = xp[(size_t)g * H];
        cprev = a.c[sidx];
    }

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float *hrow = a.h_in + ((size_t)dir * a.npad + slice * 16 + li) * H;
    const f32x4 *wf = reinterpret_cast<const f32x4 *>(a.whh_frag) + ((size_t)(dir * KGT + ug) * KGT) * 4 * 64 + lane;
    if (KPW > 0 && a.whh2) {
        if constexpr (KPW > 0) {
            LstmW2<KPW> w2;
            lstm_load_w_f16x2<KPW>(a.whh2, dir, ug, wave, lane, w2);
            lstm_gemm_f16x2<KPW, false>(hrow, wave, kq, w2, acc);
        }
    } else if constexpr (KPW > 0) {
        f32x4 av[KPW], bv[KPW][4];
#pragma unroll
        for (int q = 0; q < KPW; ++q) {
            const int kg = wave + 4 * q;
            av[q] = *reinterpret_cast<const f32x4 *>(hrow + kg * 16 + kq * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[q][g] = wf[((size_t)kg * 4 + g) * 64];
        }
#pragma unroll
        for (int q = 0; q < KPW; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][j], bv[q][g][j], acc[g], 0, 0, 0);
    } else {
        for (int kg = wave; kg < KGT; kg += 4) {
            const f32x4 av = *reinterpret_cast<const f32x4 *>(hrow + kg * 16 + kq * 4);
            f32x4 bv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[g] = wf[((size_t)kg * 4 + g) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[g][j], acc[g], 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4 *>(&part[((wave * 4 + g) * 64 + lane) * 4]) = acc[g];
    __syncthreads();

    // thread -> (line i, unit u); D layout: lane = (i/4)*16 + u, reg = i%4
    const int src = (((i >> 2) * 16 + u) * 4) + (i & 3);
    float gate[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float s = part[(0 * 4 + g) * 256 + src];
        s += part[(1 * 4 + g) * 256 + src];
        s += part[(2 * 4 + g) * 256 + src];
        s += part[(3 * 4 + g) * 256 + src];
        gate[g] = s;
    }
    if (live) {
        float cn, hn;
        lstm_cell(gate, xg, cprev, cn, hn);
        a.c[sidx] = cn;
        a.h_out[sidx] = hn;
        lstm_store_y(a.y, row, 2 * H, dir * H + unit, hn, a.y_p2 != 0);
    } else {
        a.h_out[sidx] = 0.f;    // padding rows of the last slice and finished lines: value is never used
    }
}

}  // namespace pocr
