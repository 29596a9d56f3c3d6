// EXPERIMENTS, not part of the product (moved out of pero_ocr_amd/csrc/lstm.hpp in round 3): two repackings of the BiLSTM
// recurrence step that were measured and are slower than the shipped lstm_step_kernel, alone and next to a conv backbone
// (profiles/r02_lstm_persistent_experiment.txt).  Kept for reference; include after csrc/lstm.hpp (LstmStepArgs).
#pragma once
#include "../../pero_ocr_amd/csrc/lstm.hpp"

namespace pocr {

// Split-K like lstm_step_kernel, but every wave applies its quarter of the W_hh fragments (held in registers) to SL
// consecutive 16-line slices: 1/SL of the workgroups and of the W_hh traffic from L2 for the same latency chain
// (the MFMA phase grows by 64 instructions per extra slice, ~0.2 us).  grid = (H/16, ceil(slices/SL), 2).
template <int KPW, int SL>
__global__ __launch_bounds__(256) void lstm_step_multi_kernel(LstmStepArgs a) {
    constexpr int H = 64 * KPW, KGT = H / 16;
    __shared__ float part[SL * 4 * 4 * 64 * 4];      // [slice][wave][gate][lane][reg]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int ug = blockIdx.x, lg = blockIdx.y, dir = blockIdx.z;
    if (a.dims) { a.n = a.dims[0]; a.npad = a.dims[1]; }
    bool sl_live[SL];
    bool any = false;
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        const int sl = lg * SL + q;
        sl_live[q] = sl * 16 < a.npad && !(a.slice_T && a.step >= a.slice_T[sl]);
        any = any || sl_live[q];
    }
    if (!any) return;

    const int u = tid & 15, i = tid >> 4;
    const int unit = ug * 16 + u;
    float xg[SL][4], cprev[SL];
    bool live[SL];
    size_t row[SL], sidx[SL];
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        const int line = (lg * SL + q) * 16 + i;
        sidx[q] = ((size_t)dir * a.npad + line) * H + unit;
        const int Ti = (sl_live[q] && line < a.n) ? (a.line_T ? a.line_T[line] : a.T) : 0;
        live[q] = a.step < Ti;
        const int t = dir == 0 ? a.step : Ti - 1 - a.step;
        row[q] = (a.row_off ? (size_t)a.row_off[min(line, a.n - 1)] : (size_t)line * a.T) + t;
        cprev[q] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[q][g] = 0.f;
        if (live[q]) {
            const float *xp = a.xproj + row[q] * (8 * H) + (size_t)dir * 4 * H + unit;
#pragma unroll
            for (int g = 0; g < 4; ++g) xg[q][g] = xp[(size_t)g * H];
            cprev[q] = a.c[sidx[q]];
        }
    }
    const f32x4 *wf = reinterpret_cast<const f32x4 *>(a.whh_frag) + ((size_t)(dir * KGT + ug) * KGT) * 4 * 64 + lane;
    f32x4 bv[KPW][4], av[SL][KPW];
#pragma unroll
    for (int q = 0; q < KPW; ++q) {
        const int kg = wave + 4 * q;
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[q][g] = wf[((size_t)kg * 4 + g) * 64];
    }
#pragma unroll
    for (int sq = 0; sq < SL; ++sq) {
        const float *hrow = a.h_in + ((size_t)dir * a.npad + (lg * SL + sq) * 16 + li) * H + kq * 4;
#pragma unroll
        for (int q = 0; q < KPW; ++q)
            av[sq][q] = sl_live[sq] ? *reinterpret_cast<const f32x4 *>(hrow + (wave + 4 * q) * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int sq = 0; sq < SL; ++sq) {
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KPW; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sq][q][j], bv[q][g][j], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4 *>(&part[(((sq * 4 + wave) * 4 + g) * 64 + lane) * 4]) = acc[g];
    }
    __syncthreads();
    const int src = (((i >> 2) * 16 + u) * 4) + (i & 3);
#pragma unroll
    for (int sq = 0; sq < SL; ++sq) {
        if (!sl_live[sq]) continue;
        float gate[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float t = part[((sq * 4 + 0) * 4 + g) * 256 + src];
            t += part[((sq * 4 + 1) * 4 + g) * 256 + src];
            t += part[((sq * 4 + 2) * 4 + g) * 256 + src];
            t += part[((sq * 4 + 3) * 4 + g) * 256 + src];
            gate[g] = t;
        }
        if (live[sq]) {
            const float gi = sigmoid_f32(gate[0] + xg[sq][0]);
            const float gf = sigmoid_f32(gate[1] + xg[sq][1]);
            const float gg = tanhf(gate[2] + xg[sq][2]);
            const float go = sigmoid_f32(gate[3] + xg[sq][3]);
            const float cn = gf * cprev[sq] + gi * gg;
            const float hn = go * tanhf(cn);
            a.c[sidx[sq]] = cn;
            a.h_out[sidx[sq]] = hn;
            a.y[row[sq] * (2 * H) + (size_t)dir * H + unit] = hn;
        } else {
            a.h_out[sidx[sq]] = 0.f;
        }
    }
}

// The same step for FOUR 16-line slices per workgroup: grid = (H/16 unit groups, ceil(slices/4), 2 directions), a quarter of
// the workgroups of lstm_step_kernel.  The step is a latency chain either way (~10 us), so what matters next to a conv
// backbone running on the other stream is how much of the chip the chain occupies: here the unit group's 64 KB of W_hh
// fragments are fetched from L2 once per 64 lines (staged in LDS, read by all four waves), each wave owns one slice over
// the whole K (no split-K partial sums, no LDS reduction, no second barrier), and the four gates of a (line, unit) pair
// already sit in one lane's accumulators (D layout: lane = 16 * (line / 4) + unit, register = line % 4).
template <int KPW>                              // H = 64 * KPW
__global__ __launch_bounds__(256) void lstm_step_wide_kernel(LstmStepArgs a) {
    constexpr int H = 64 * KPW, KGT = H / 16;
    __shared__ f32x4 wlds[KGT * 256];           // [k group][gate][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int ug = blockIdx.x, lg = blockIdx.y, dir = blockIdx.z;
    if (a.dims) { a.n = a.dims[0]; a.npad = a.dims[1]; }
    bool wg_live = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sl = lg * 4 + q;
        wg_live = wg_live || (sl * 16 < a.npad && !(a.slice_T && a.step >= a.slice_T[sl]));
    }
    if (!wg_live) return;                       // (uniform over the workgroup)
    const int slice = lg * 4 + wave;
    const bool wave_live = slice * 16 < a.npad && !(a.slice_T && a.step >= a.slice_T[slice]);

    // W_hh fragments of this (direction, unit group): one coalesced sweep, KGT float4 per thread
    const f32x4 *wf = reinterpret_cast<const f32x4 *>(a.whh_frag) + ((size_t)(dir * KGT + ug) * KGT) * 256;
    f32x4 wreg[KGT];
#pragma unroll
    for (int q = 0; q < KGT; ++q) wreg[q] = wf[q * 256 + tid];

    // epilogue operands (HBM: xproj is streamed) and the slice's h rows, all requested before anything is consumed
    const int unit = ug * 16 + li;
    float xg[4][4], cprev[4];
    bool live[4];
    size_t row[4], sidx[4];
    f32x4 av[KGT];
    if (wave_live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int line = slice * 16 + kq * 4 + r;
            const int Ti = line < a.n ? (a.line_T ? a.line_T[line] : a.T) : 0;
            live[r] = a.step < Ti;
            const int t = dir == 0 ? a.step : Ti - 1 - a.step;
            row[r] = (a.row_off ? (size_t)a.row_off[min(line, a.n - 1)] : (size_t)line * a.T) + t;
            sidx[r] = ((size_t)dir * a.npad + line) * H + unit;
            cprev[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) xg[g][r] = 0.f;
            if (live[r]) {
                const float *xp = a.xproj + row[r] * (8 * H) + (size_t)dir * 4 * H + unit;
#pragma unroll
                for (int g = 0; g < 4; ++g) xg[g][r] = xp[(size_t)g * H];
                cprev[r] = a.c[sidx[r]];
            }
        }
        const float *hrow = a.h_in + ((size_t)dir * a.npad + slice * 16 + li) * H + kq * 4;
#pragma unroll
        for (int q = 0; q < KGT; ++q) av[q] = *reinterpret_cast<const f32x4 *>(hrow + q * 16);
    }
#pragma unroll
    for (int q = 0; q < KGT; ++q) wlds[q * 256 + tid] = wreg[q];
    __syncthreads();
    if (!wave_live) return;

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KGT; ++q) {
        f32x4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = wlds[(q * 4 + g) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][j], bv[g][j], acc[g], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (live[r]) {
            const float gi = sigmoid_f32(acc[0][r] + xg[0][r]);
            const float gf = sigmoid_f32(acc[1][r] + xg[1][r]);
            const float gg = tanhf(acc[2][r] + xg[2][r]);
            const float go = sigmoid_f32(acc[3][r] + xg[3][r]);
            const float cn = gf * cprev[r] + gi * gg;
            const float hn = go * tanhf(cn);
            a.c[sidx[r]] = cn;
            a.h_out[sidx[r]] = hn;
            a.y[row[r] * (2 * H) + (size_t)dir * H + unit] = hn;
        } else {
            a.h_out[sidx[r]] = 0.f;
        }
    }
}

}  // namespace pocr
