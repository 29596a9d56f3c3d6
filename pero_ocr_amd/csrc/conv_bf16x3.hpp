// conv_bf16x3.hpp — 3x3 implicit-GEMM convolution with fp32-level accuracy on the bf16 matrix pipe.
//
// v_mfma_f32_16x16x4_f32 (conv_igemm.hpp) runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA rate, and the conv
// backbone sits at 85-89 % of that 157.3 TFLOP/s peak: the only way up is a different arithmetic.  Every fp32 operand is
// split EXACTLY into three bf16 values (8 significand bits each, by truncation: x = hi + mid + lo, all of x's sign):
//     a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh) + O(2^-24 |a*b|)
// i.e. six v_mfma_f32_16x16x32_bf16 per 32-deep product block instead of eight f32 MFMAs: 6 x 16 cycles instead of
// 8 x 32 per wave -> 2.67x the fp32-MFMA rate.  Products of two bf16 values are exact in fp32, accumulation is fp32 inside
// the MFMA, the three dropped terms are each <= 2^-24 of the product: the same error class as one fp32 rounding.
//
// Layout / tiling as conv_igemm_kernel: M = 16 consecutive pixels of an image row, N = output channels, the 4 waves of
// a workgroup split N, each wave keeps TH*MW row tiles x NS channel tiles of accumulators (same D layout -> same fused
// bias / activation / BatchNorm / max-pool epilogue).  K chunk = 32 input channels:
//   A: the halo tile is split by the stager, once per 32-channel chunk, into three bf16 planes in LDS
//      ([plane][channel octet][pixel][8 bf16]); a lane's A operands (pixel, 8 channels) are three conflict-free ds_read_b128;
//      the tile is single-buffered (one extra barrier per chunk = per 9 steps) to leave LDS for two workgroups per CU;
//   B: weights split on the host, fragment order wsplit[tap][cin/32][cout/16][plane 3][lane][8 bf16]
//      = W[cout = 16 s + (lane & 15)][cin = 32 g + 8 (lane >> 4) + j][tap], streamed through a double-buffered LDS tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "conv_igemm.hpp"

namespace pocr {

// Tuning constants of the loops below (each was a compile-time knob while it was being measured: profiles/r02_conv_bf16x3_bench.txt,
// r03_conv_rowstream.txt, r05_conv_rows.txt; the losing settings and the ablation / trace switches were removed in round 6).
constexpr int kBdirSets = 3;     // register sets of weight fragments in the direct-weights loop: the set of step s + kBdirSets - 1 is requested while step s computes
constexpr int kRowAhead = 2;     // row streaming: A fragments requested this many (row, strip) units before the MFMAs that use them
constexpr int kRowLdaQ = 3;      // row streaming: the next chunk's halo tile is requested after this unit of the chunk's first group (after the weight
                                 // requests of units 0..2: vmcnt counts in order, so a load issued BEFORE them must land before the next group starts)
constexpr int kRowStaDx = 2;     // ... and written to the other LDS buffer after this group (2 = right before the chunk's barrier)
constexpr int kStagePad = 16;    // bytes of padding behind a staged pixel (conv_epilogue_staged): pixel stride = 4 dwords mod 64 banks
// Fixed design choices that used to be switches: issue-order templates (sched_group_barrier) in the main loops (-2 ... -9 % per layer); the
// halo-row streaming loop for the 3x3 / f16x2 / direct-weights layers; double-buffered A and B tiles in the 1x1 (GEMM mode) f16x2 layers
// with LDS weights; A fragments read one tap ahead in the tap-by-tap direct-weights loop (f16x2, MS <= 4); the next chunk's halo tile
// written to the other LDS buffer after tap NTAP / 2; f16x2 kernels that write P2 take the weights as the MFMA's A operand ("TR").

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// exact three-way split of 8 floats into packed bf16 (truncation keeps the sign and makes hi + mid + lo == x bit for bit)
__device__ __forceinline__ void split3_bf16(const f32x4 &p, const f32x4 &q, u32x4 &hi, u32x4 &mid, u32x4 &lo) {
    float x[8] = {p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3]};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned xb = __builtin_bit_cast(unsigned, x[i]);
        h[i] = xb & 0xffff0000u;
        const float r1 = x[i] - __builtin_bit_cast(float, h[i]);
        m[i] = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
        const float r2 = r1 - __builtin_bit_cast(float, m[i]);
        l[i] = __builtin_bit_cast(unsigned, r2) & 0xffff0000u;       // r2 has <= 8 significant bits left: exact
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (h[2 * i] >> 16) | h[2 * i + 1];
        mid[i] = (m[2 * i] >> 16) | m[2 * i + 1];
        lo[i] = (l[2 * i] >> 16) | l[2 * i + 1];
    }
}

template <class TA, class TB>
__device__ __forceinline__ f32x4 mfma16_bf16(const TA a, const TB b, const f32x4 c) {       // operands: any 16-byte vector of eight bf16
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// split of one float4 (4 consecutive channels) into three pairs of packed bf16 (8 bytes per plane)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_quad(const f32x4 p, u32x2 &hi, u32x2 &mid, u32x2 &lo) {
    unsigned h[4], m[4], l[4];
    const float x[4] = {p[0], p[1], p[2], p[3]};        // (bit-casting the vector element expression p[i] itself reads element 0: hipcc 7.2)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned xb = __builtin_bit_cast(unsigned, x[i]);
        h[i] = xb & 0xffff0000u;
        const float r1 = x[i] - __builtin_bit_cast(float, h[i]);
        m[i] = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
        const float r2 = r1 - __builtin_bit_cast(float, m[i]);
        l[i] = __builtin_bit_cast(unsigned, r2) & 0xffff0000u;
    }
    hi = (u32x2){(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
    mid = (u32x2){(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
    lo = (u32x2){(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]};
}

// ---- f16x2: the same idea on the f16 matrix pipe with TWO planes and THREE MFMAs per product block.
// x = h + l * 2^-11 with h = f16(x) (round to nearest, 11 significant bits) and l = f16((x - h) * 2^11): the residual is
// scaled into f16's normal range, so l carries another 11 bits whatever the magnitude of x (x is represented to 2^-22
// relative, as against 2^-24 for fp32 itself).  a * b = ah * bh + 2^-11 (ah * bl + al * bh) + 2^-22 al * bl: the last term
// is dropped (<= 2^-22 |a b|), the two middle terms run in their own accumulator which is scaled by 2^-11 once in the
// epilogue.  Products of two f16 values are exact in fp32 (11 + 11 bits).  Per product the error is ~2^-22 |a b| with
// random sign, which over K terms adds up like sqrt(K) - far below the K / 32 accumulator roundings both schemes share
// (and the K roundings of an fp32 fma chain): measured against float64 the two splits are equally close, 3-5x closer
// than fp32 MFMA (tools/conv_bench_bf16.hip, profiles/r03_conv_f16x2_bench.txt).  Range: |x| must stay below 65504 (f16's
// largest finite value) - activations and weights of a recogniser are O(1); POCR_CONV_SPLIT=3 selects bf16x3 (fp32's range).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <class TA, class TB>
__device__ __forceinline__ f32x4 mfma16_f16(const TA a, const TB b, const f32x4 c) {        // operands: any 16-byte vector of eight f16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
constexpr float kF16x2Scale = 2048.0f;

// x -> (h, l): h = f16(x), l = f16((x - h) * 2^11), round to nearest even.  (Tried: l = f16(fma(h, -2^11, 2^11 x)) on v_fma_mixlo /
// v_fma_mixhi in inline asm, 8 instead of 12 vector instructions per four values, bit-identical in isolation - but the compiler does
// not track the sub-dword write hazard of those instructions inside an asm block: next to stores the results were right, an MFMA that
// consumed them at once read stale low planes (the recurrence lost its l plane); written in C++ the compiler picks v_pk_fma_f32 and
// the same 12 instructions.  No measurable time in any layer either way: dropped.)
__device__ __forceinline__ void split2_quad(const f32x4 p, u32x2 &hi, u32x2 &lo) {
    const f32x2 a = {p[0], p[1]}, b = {p[2], p[3]};
    const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);       // round to nearest even
    const f32x2 ra = (a - __builtin_convertvector(ha, f32x2)) * kF16x2Scale, rb = (b - __builtin_convertvector(hb, f32x2)) * kF16x2Scale;
    const f16x2 la = __builtin_convertvector(ra, f16x2), lb = __builtin_convertvector(rb, f16x2);
    hi = (u32x2){__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
    lo = (u32x2){__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}

// ---- P2: activations kept in HBM already split ("pre-split", f16x2 only).  The two-plane f16 representation of a value
// takes exactly the four bytes of the fp32 it replaces, so a producer's epilogue can split ONCE per output value and every
// consumer's stager becomes a plain 16-byte copy HBM -> LDS: no split arithmetic and no fp32 staging registers in the main
// loop (measured: the in-loop staging is 12 % of conv9), and each halo tile is no longer re-split by every channel-tile
// workgroup that reads it.  Layout of a pixel with C channels (C % 32 == 0), 4 C bytes as before:
//     for each 32-channel chunk g:  [h of channels 32 g .. 32 g + 31 : 32 x f16][l of the same channels : 32 x f16]
// i.e. a chunk is 128 contiguous bytes = the eight 16-byte units (plane, channel octet) the MFMA A operands are made of.
// Offsets and sizes of lines / pixels are those of the fp32 layout, so the geometry tables and pad_fill_kernel do not change.
__device__ __forceinline__ void split2_scalar(float v, _Float16 &h, _Float16 &l) {
    h = (_Float16)v;
    l = (_Float16)((v - (float)h) * kF16x2Scale);
}
__device__ __forceinline__ size_t p2_channel_bytes(int c) { return (size_t)(c >> 5) * 128 + (size_t)(c & 31) * 2; }   // plane h; plane l at + 64

// conv1 in the f16x2 arithmetic of the rest of the stack: K = 27 fits ONE 32-deep product block, so 16 pixels x 16 channels
// are three MFMAs (w_h x_l, w_h x_h, w_l x_h) instead of eight fp32 ones.  The WEIGHTS are the MFMA's A operand (rows =
// channels) and the pixels its B operand, so a lane ends up with four consecutive channels 4 (lane >> 4) + r of ONE pixel
// (lane & 15) - the shape the NHWC / P2 stores want, no transpose.  `patch` holds the normalised input (fp32, [row][col][c],
// PW pixels per row), `base` the element of tap (0, 0) / channel 0 of this lane's pixel, Conv1Slots the lane's
// k slots 8 (lane >> 4) + j.  Shared by conv1_u8_kernel and by conv2's fused prologue
// (conv3x3_bf16x3_kernel FUSE1), which therefore give the same bits.
// k slot 27 carries the BIAS: its input is the constant 1 and its weight the bias (split like a weight), so the MFMAs add it and
// the epilogue does not; slots 28..31 have zero weights.  Patch layout behind the n_patch pixels' values: [one : 1.0][zeros].
// Index of slot k = 8 kq + j of a pixel whose tap (0,0) element is patch[base]: base * sel[j] + off[j] (one v_mad per slot):
// k < 27: sel 1, off = the tap's offset; k = 27: sel 0, off = tail (the constant 1); k > 27: sel 0, off = tail + 1 (a zero).
constexpr int kConv1Tail = 4;                           // floats behind the patch: [0] = 1.0, [1..3] = 0
struct Conv1Slots { int sel[8], off[8]; bool has_one; };
__device__ __forceinline__ void conv1_koff(Conv1Slots &ks, int kq, int PW, int tail) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * kq + j, tap = k / 3, c = k - 3 * tap;
        ks.sel[j] = k < 27 ? 1 : 0;
        ks.off[j] = k < 27 ? ((tap / 3) * PW + tap % 3) * 3 + c : k == 27 ? tail : tail + 1;
    }
    ks.has_one = kq == 3;                               // (k = 27 is slot j = 3 of the lanes with kq = 3)
}
// for a pixel whose output must be ZERO (outside the image: the consumer's padding) pass one = false and a base whose 3 x 3 x 3
// neighbourhood is zeros: its inputs and its bias slot are then zero, and so is what the MFMAs return
__device__ __forceinline__ void conv1_x_frag(const float *patch, int base, const Conv1Slots &ks, int tail, bool one, u32x4 &xh) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int idx = base * ks.sel[j] + ks.off[j];
        if (j == 3) idx = (ks.has_one && !one) ? tail + 1 : idx;
        x[j] = patch[idx];
    }
    // the inputs are the PIXEL BYTES / 256 (and the constant 1): exact in f16, so there is no low plane to make - the 256 / 255 that
    // is missing from the reference's float32(x) / 255 (pytorch_ocr_engine.py:61) sits in the weights (pocr_create)
    const f16x2 p0 = __builtin_convertvector((f32x2){x[0], x[1]}, f16x2), p1 = __builtin_convertvector((f32x2){x[2], x[3]}, f16x2);
    const f16x2 p2 = __builtin_convertvector((f32x2){x[4], x[5]}, f16x2), p3 = __builtin_convertvector((f32x2){x[6], x[7]}, f16x2);
    xh = (u32x4){__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), __builtin_bit_cast(unsigned, p2), __builtin_bit_cast(unsigned, p3)};
}
// ReLU of four conv1 outputs (bias already inside) + the f16x2 range note of non-negative values: their bit patterns order like
// signed integers (-0 = INT_MIN never wins; v_max_f32 turns a NaN into 0 as the compare-and-select before did)
__device__ __forceinline__ f32x4 conv1_relu_note(const f32x4 d, unsigned &m) {
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaxf(d[r], 0.f);
    int mm = (int)m;
    mm = max(max(mm, __builtin_bit_cast(int, v[0])), __builtin_bit_cast(int, v[1]));
    mm = max(max(mm, __builtin_bit_cast(int, v[2])), __builtin_bit_cast(int, v[3]));
    m = (unsigned)mm;
    return v;
}
__device__ __forceinline__ f32x4 conv1_mma_f16x2(u32x4 xh, u32x4 wh, u32x4 wl) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 acc2 = mfma16_f16(wl, xh, z);
    const f32x4 acc = mfma16_f16(wh, xh, z);
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = __builtin_fmaf(acc2[r], 1.0f / kF16x2Scale, acc[r]);
    return d;
}

// One product block of the convolution loops: `act` the pixels' fragment, `w` the weights'.  The two operands of
// v_mfma_f32_16x16x32_f16 have the same fragment layout, so which of them is "A" only decides the layout of the RESULT:
// TR = false: D[pixel 4 kq + r][channel li] (a lane holds four pixels of one channel), TR = true: D[channel 4 kq + r][pixel li]
// (four consecutive channels of one pixel = one 8-byte piece of each P2 plane: no transpose in front of the stores).
constexpr int conv_stage_units(int TH, int TW, int POOLH, int POOLW, int NT) {      // 16-byte units of conv_epilogue_staged's staging area
    return ((TH / POOLH) * (TW / POOLW) * (NT * 4 + kStagePad) + 15) / 16;
}
template <bool TR>
__device__ __forceinline__ f32x4 mfma_conv_f16(u32x4 act, u32x4 w, f32x4 c) {
    if constexpr (TR) return mfma16_f16(w, act, c);
    else return mfma16_f16(act, w, c);
}

// Issue-order template for the scheduler (LDS-weights loop): the G operand reads of a step spread evenly between its TOT MFMAs.
template <int G, int TOT, int... I>
__device__ __forceinline__ void sched_template_lds(std::integer_sequence<int, I...>) {
    ((__builtin_amdgcn_sched_group_barrier(0x008, ((I + 1) * TOT) / G - (I * TOT) / G, 0), __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)), ...);
}

// The same for the direct-weights loop: G operand reads, TOT MFMAs, NV weight loads (mask 0x020 VMEM read) spread over the step.
template <int G, int TOT, int NV, int I>
__device__ __forceinline__ void sched_group_dir() {
    __builtin_amdgcn_sched_group_barrier(0x008, ((I + 1) * TOT) / G - (I * TOT) / G, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if constexpr ((I * NV) / G != ((I + 1) * NV) / G) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
}
template <int G, int TOT, int NV, int... I>
__device__ __forceinline__ void sched_template_dir(std::integer_sequence<int, I...>) {
    (sched_group_dir<G, TOT, NV, I>(), ...);
}

// ---- The epilogue of the kernels whose MFMAs leave the result as [channel][pixel] (mfma_conv_f16<true>): lane (li, kq) holds
// channels 4 kq .. + 3 of pixel li of every 16-pixel tile, i.e. 8-byte pieces of each P2 plane.  Stored from there, a wave
// instruction writes sixteen 32-byte segments of sixteen different lines - and the partial-line writes, not the bytes, were what
// the stores cost (the same bytes as full lines: conv3 -13 %, conv5 -9 %, the others -2..3 %, tools/conv_wino_bench.hip).  So the
// tile's output is put together in LDS (`stage`: the A buffers, free behind the main loop; a pixel = its NT channels in P2 order + 32
// bytes of padding, so that the 8-byte writes of a wave fall on all banks) and leaves as whole 128-byte lines, 16 bytes per lane.
// bias_p / scale_p / shift_p: the channel tile's NT constants (LDS or global); yline: the line's output image; TWO barriers inside.
template <int TH, int MWW, int NS, int WM, int POOLH, int POOLW, int ACT, bool BN, int NT, int TW>
__device__ __forceinline__ void conv_epilogue_staged(const f32x4 (&acc)[TH * MWW][NS], const f32x4 (&acc2)[TH * MWW][NS], const float *bias_p,
                                                     const float *scale_p, const float *shift_p, char *stage, float *yline, int h0, int w0, int Win,
                                                     int Ho, int out_stride, int nt, unsigned &rmax) {
    constexpr int NTHR = 256, TWO = TW / POOLW, NPX = (TH / POOLH) * TWO, UPP = NT / 4, PBP = NT * 4 + kStagePad;
    constexpr unsigned kOut = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int wm = wave % WM, wn = wave / WM;
    (void)lane;
        const int Wo = Win, Wout = Wo / POOLW;
        const unsigned pix_bytes = (unsigned)out_stride * 4u;
        const size_t img_bytes = (size_t)(Ho / POOLH) * Wout * pix_bytes;
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yline, 0, (int)(img_bytes < kOut ? img_bytes : kOut - 1), 0x00020000);
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const int cgl = wn * NS + n;                              // 16-channel group inside the channel tile
            const int cl = cgl * 16 + 4 * kq;                         // this lane's channels cl + r
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(bias_p + cl);
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BN) { sc = *reinterpret_cast<const f32x4 *>(scale_p + cl); sh = *reinterpret_cast<const f32x4 *>(shift_p + cl); }
            // byte of the lane's h piece inside a staged pixel: chunk cgl >> 1, half (cgl & 1) * 32, + 8 kq; the l piece 64 further
            const unsigned piece = (unsigned)(cgl >> 1) * 128u + (unsigned)(cgl & 1) * 32u + (unsigned)kq * 8u;
#pragma unroll
            for (int th = 0; th < TH; th += POOLH) {
#pragma unroll
                for (int mw = 0; mw < MWW; ++mw) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = apply_act(acc[th * MWW + mw][n][r] + acc2[th * MWW + mw][n][r] * (1.0f / kF16x2Scale) + bias[r], ACT);
                        if constexpr (BN) t = t * sc[r] + sh[r];
                        if constexpr (POOLH == 2) {
                            float u2 = apply_act(acc[(th + 1) * MWW + mw][n][r] + acc2[(th + 1) * MWW + mw][n][r] * (1.0f / kF16x2Scale) + bias[r], ACT);
                            if constexpr (BN) u2 = u2 * sc[r] + sh[r];
                            t = fmaxf(t, u2);
                        }
                        if constexpr (POOLW == 2)          // the other pixel of the pair sits in the neighbouring lane
                            t = fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true)));
                        v[r] = t;
                    }
                    if constexpr (ACT == ACT_RELU && !BN) {    // non-negative: bit patterns order like signed integers
                        int mm = (int)rmax;
                        mm = max(max(mm, __builtin_bit_cast(int, v[0])), __builtin_bit_cast(int, v[1]));
                        mm = max(max(mm, __builtin_bit_cast(int, v[2])), __builtin_bit_cast(int, v[3]));
                        rmax = (unsigned)mm;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) range_note(rmax, v[r]);
                    }
                    u32x2 hh, ll;
                    split2_quad(v, hh, ll);
                    // staged pixel (row th / POOLH, column ((wm MWW + mw) 16 + li) / POOLW)
                    const int sp = (th / POOLH) * TWO + ((wm * MWW + mw) * 16 + li) / POOLW;
                    if (POOLW == 1 || (li & 1) == 0) {
                        u32x2 *d = reinterpret_cast<u32x2 *>(stage + (unsigned)sp * PBP + piece);
                        d[0] = hh; d[8] = ll;
                    }
                }
            }
        }
        __syncthreads();
        // 16-byte unit g of the tile = (pixel g / UPP, unit g % UPP): a wave instruction stores 64 consecutive units = whole pixels
        const unsigned tile_off = (unsigned)(h0 / POOLH) * (unsigned)Wout * pix_bytes + (unsigned)(w0 / POOLW) * pix_bytes + (unsigned)nt * (NT * 4u);
        const int rows_ok = Ho / POOLH - h0 / POOLH, cols_ok = Wout - w0 / POOLW;      // staged rows / columns inside the image
#pragma unroll
        for (int it = 0; it < (NPX * UPP + NTHR - 1) / NTHR; ++it) {
            const int g = it * NTHR + tid, sp = g / UPP, unit = g % UPP, row = sp / TWO, col = sp % TWO;
            const bool ok = ((NPX * UPP) % NTHR == 0 || sp < NPX) && row < rows_ok && col < cols_ok;
            const u32x4 val = *reinterpret_cast<const u32x4 *>(stage + (unsigned)sp * PBP + (unsigned)unit * 16u);
            // A branch around the store, NOT an out-of-range offset for the lanes outside the image.  Masked that way the stores are
            // dropped correctly (tools/masked_store_probe.hip) - but without a branch between consecutive stores the scheduler puts the
            // next store's address arithmetic into the data register the previous one just freed:
            //     buffer_store_dwordx4 v[2:5], v22, s[4:7], s11 offen ; v_lshrrev_b32 v2, 5, v1
            // and on gfx950 a store of more than 64 bits WITH an SGPR offset still needs one wait state before a VALU write of its data
            // registers (without the offset: two); LLVM's hazard recogniser pads only the second case.  Under load the store then writes the
            // LATER value of v2: wrong dwords in this kernel's own output, 1.8 % of the stores in tools/store_hazard_probe.hip, every fresh
            // engine in tools/three_in_flight.py (round 5's "rare wrong low planes ... why is not understood").  The branching form never has
            // a VALU write of a data register behind a store; tests/test_host.py::test_no_kernel_has_the_store_data_hazard scans every
            // kernel of the built library for the pair.  Loads masked by the mark - the halo's zero padding - have no data registers to
            // lose and read zeros.  profiles/r06_store_hazard.txt.
#if POCR_EPI_MASKED_STORE        // (tools/masked_store_repro.sh builds this variant into tools/bin to study the effect; never shipped)
            __builtin_amdgcn_raw_buffer_store_b128(val, yrsrc, (int)(ok ? (unsigned)(row * Wout + col) * pix_bytes + (unsigned)unit * 16u : kOut), (int)tile_off, 0);
#else
            if (ok) __builtin_amdgcn_raw_buffer_store_b128(val, yrsrc, (int)((unsigned)(row * Wout + col) * pix_bytes + (unsigned)unit * 16u), (int)tile_off, 0);
#endif
        }
        __syncthreads();                                             // (the staging area is the next tile's A buffer again)
}

// WM waves split the pixel tile (column strips), 4 / WM waves split the output channels; the B tile (weights of one
// (chunk, tap) step for NT channels) is shared through LDS by the WM waves that need it, the A halo tile - already split
// into its three bf16 planes by the stager, once per 32-channel chunk - by all of them.
// BDIR: the weights do not pass through LDS at all - every wave loads the fragments of ITS channel tiles straight from
// L2 into registers, one (chunk, tap) step ahead (fragment order makes that one coalesced 16 B/lane load per tile and
// plane).  LDS then carries only the A planes (double-buffered: one barrier per 32-channel chunk instead of ten), which
// matters because the loop is LDS-bound as much as MFMA-bound: per 16-cycle MFMA a SIMD's share of the LDS pipe is 512 B,
// and operands cost 512 B x (1/NS + 1/MS) to read plus ~100 B to write the shared weight tile.
// KH x KW taps with PADH / PADW rows / columns of zero padding: 3x3 pad 1 (the backbone), AH x 1 pad 0 (the aggregation
// conv), 1x1 pad 0 (GEMM mode: "pixels" are the rows of a [rows][cin] matrix).
// UPCAT: the input is the virtual tensor cat([nearest-upsample-x2(x), x2], channels) of the layout network's decoder
// (conv_igemm.hpp STAGE_UPCAT): 32-channel chunks below cin_up come from x at half resolution, the rest from the skip tensor.
template <int TH, int MW, int NS, int WM, int POOLH, int POOLW, int ACT, bool BN, int MINW = 1, bool BDIR = false,
          int KH = 3, int KW = 3, int PADH = 1, int PADW = 1, bool UPCAT = false, int SPL = 3, bool PRE_IN = false, bool PRE_OUT = false,
          bool FUSE1 = false>
__global__ __launch_bounds__(256, MINW) void conv3x3_bf16x3_kernel(ConvArgs a) {
    // FUSE1 (conv2 of the recogniser, cin = 64): the input tensor is never read - the workgroup computes conv1 (3 -> 64, 3x3,
    // ReLU, from the uint8 crops: conv1_u8.hpp) for the pixels of its own halo tile straight into the two A buffers in LDS.
    // conv1's output (1.5 GB per 256-line launch, written at 2.9 TB/s and read back by conv2) then does not exist; the price
    // is conv1 recomputed on the halo overlap, 168 extra MFMAs per workgroup against 2160.
    static_assert(!FUSE1 || (PRE_IN && BDIR && KH == 3 && KW == 3 && SPL == 2), "FUSE1 rides on the row-streaming P2 loop");
    // PRE_IN: the input is in the P2 (pre-split) layout; PRE_OUT: the epilogue writes that layout (both f16x2 only)
    static_assert(!(PRE_IN || PRE_OUT) || SPL == 2, "the pre-split activation layout is the f16x2 representation");
    static_assert(!(PRE_IN && UPCAT), "the layout network keeps fp32 activations");
    // SPL = planes per operand: 3 = bf16x3 (six MFMAs per 32-deep product block), 2 = f16x2 (three)
    static_assert(SPL == 2 || SPL == 3, "operand split: 3 bf16 planes or 2 f16 planes");
    constexpr bool TR = SPL == 2 && PRE_OUT;      // result layout [channel][pixel] (mfma_conv_f16) and the epilogue written for it
    constexpr int NWAVE = 4, WN = NWAVE / WM, KC = 32, NTAP = KH * KW, WU = SPL * 64, NMF = SPL == 3 ? 6 : 3;
    static_assert(MW % WM == 0, "column strips must divide among the M waves");
    constexpr int TW = 16 * MW, MWW = MW / WM, MS = TH * MWW, NT = NS * WN * 16, NTHR = 256;
    constexpr int HH = TH + KH - 1, HW = TW + KW - 1, NP = HH * HW, NPPAD = (NP + 15) / 16 * 16;
    constexpr int CQ = KC / 4;
    constexpr int PS = 4 * NPPAD;                       // 16-byte units per bf16 plane of the A tile ([octet][pixel])
    constexpr int A_U = SPL * PS;                       // A tile (single-buffered, refilled once per chunk; BDIR: two of them)
    constexpr int B_F4 = BDIR ? 0 : (NT / 16) * WU;     // 16-byte units per B buffer (one (chunk, tap) step)
    constexpr int NP8 = (NP + 7) / 8 * 8;               // PRE_IN: 64 consecutive staging slots = 8 pixels x 8 sixteen-byte units
    constexpr int A_LD = (8 * NP8 + NTHR - 1) / NTHR;   // staging slots per thread and chunk (CQ = 8 quads or 8 pre-split units per pixel)
    constexpr int B_LD = BDIR ? 1 : (B_F4 + NTHR - 1) / NTHR;
    static_assert(POOLH == 1 || TH % 2 == 0, "H-pool needs an even tile height");
    // GEMM mode (one tap per chunk): the tap-by-tap loop below refills its single A buffer between TWO barriers per chunk, which
    // a 9-tap chunk amortises and a 1-tap chunk does not -> own loop, A double-buffered (+5 %: profiles/r03_gemm_pipe.txt; the same
    // loop with the weights straight from L2 - one register set, reloaded channel tile by channel tile - measured the same
    // 210 TFLOP/s on the 53248 x 512 x 2048 projection, so it is not the LDS traffic that holds this tile shape at ~37 % MFMA issue)
    constexpr bool GEMM2 = !BDIR && NTAP == 1 && SPL == 2;
    constexpr int A_BUFS = (BDIR || GEMM2) ? 2 : 1;
    constexpr int F1_PW = HW + 2, F1_N = (HH + 2) * F1_PW * 3;            // FUSE1: conv1's input patch ([row][col][c] floats) behind the A buffers
    constexpr int F1_Z = kConv1Tail + (2 * F1_PW + 3) * 3;            // the patch's tail + a zero 3 x 3 x 3 neighbourhood (conv1_x_frag)
    constexpr int F1_U = FUSE1 ? (F1_N + F1_Z + 3) / 4 : 0;
    __shared__ u32x4 lds[(BDIR ? 2 * A_U : A_BUFS * A_U + 2 * B_F4) + F1_U];      // one scalar type (unsigned) for every access: no type punning      // one scalar type (unsigned) for every access: no type punning
    u32x4 *ldsA = lds;
    u32x4 *ldsB = lds + A_BUFS * A_U;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int wm = wave % WM, wn = wave / WM;

    // block -> tile mapping (as conv_igemm_kernel)
    int nt, ptile;
    {
        const int tn = a.tiles_n;
        const int P = a.tiles ? a.n_ptiles : a.tiles_w * a.tiles_h * a.n;
        if (tn >= 2 && tn <= 8 && (tn & (tn - 1)) == 0) {
            const int G = a.xcd_g > 1 && tn % a.xcd_g == 0 ? a.xcd_g : 1, xg = tn / G;      // (ConvArgs::xcd_g)
            const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, groups = 8 / xg;
            nt = (xcd % xg) * G + k % G;
            ptile = (k / G) * groups + xcd / xg;
            if (ptile >= P) return;
        } else {
            int b = blockIdx.x;
            const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = b & 7, k = b >> 3;
            b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
            nt = b % tn;
            ptile = b / tn;
        }
    }
    int wt, ht, img, Win;
    size_t img_base, out_base;
    if (a.tiles) {
        const PixelTile pt = a.tiles[ptile];
        img = pt.line; ht = pt.ht_wt >> 16; wt = pt.ht_wt & 0xffff;
        Win = a.line_w[img];
        img_base = (size_t)a.in_off[img];
        out_base = (size_t)a.out_off[img];
    } else {
        wt = ptile % a.tiles_w;
        ht = (ptile / a.tiles_w) % a.tiles_h;
        img = ptile / (a.tiles_w * a.tiles_h);
        Win = a.W;
        img_base = (size_t)img * a.H * a.W * a.cin;
        out_base = (size_t)img * (a.Ho / POOLH) * (a.Wo / POOLW) * a.out_stride;
    }
    const int h0 = ht * TH, w0 = wt * TW;

    f32x4 acc[MS][NS], acc2[MS][NS];                    // main term / the five small terms
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int s = 0; s < NS; ++s) { acc[m][s] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[m][s] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int nchunks = a.cin / KC;
    unsigned a_off[A_LD];
    unsigned a_off2[UPCAT ? A_LD : 1];                  // UPCAT: offsets into the skip tensor (a_off: into the half-resolution one)
    bool a_ok[A_LD];
    int a_lds[A_LD];                                    // index (8-byte units) of the quad's slot inside plane 0, -1 = none
#pragma unroll
    for (int r = 0; r < A_LD; ++r) {
        const int e = tid + r * NTHR;
        if constexpr (PRE_IN) {
            // Slot e: eight consecutive lanes take the same unit u = (plane, octet) of eight consecutive halo pixels (their
            // ds_write_b128 fill 128 contiguous bytes of one LDS row: conflict-free), the eight lane groups of a wave the
            // eight units of those pixels (each pixel's 128-byte chunk is read whole by one wave instruction).
            const int l = e & 63, p = (e >> 6) * 8 + (l & 7), u = l >> 3;
            const int hr = p / HW, wc = p % HW;
            const int hi = h0 - PADH + hr, wi = w0 - PADW + wc;
            a_ok[r] = p < NP && hi >= 0 && hi < a.H && wi >= 0 && wi < Win;
            a_off[r] = a_ok[r] ? (unsigned)((hi * Win + wi) * (a.cin >> 2) + u) : 0u;          // 16-byte units; a chunk adds 8
            a_lds[r] = p < NP ? (u >> 2) * PS + (u & 3) * NPPAD + p : -1;                       // 16-byte units
            continue;
        }
        // Slot e of the in-kernel split: as above, eight consecutive lanes take the same channel quad cq of eight consecutive
        // pixels - their 8-byte plane stores land in one contiguous LDS row (with cq fastest the eight quads of a pixel hit
        // rows NPPAD * 16 bytes apart, i.e. the same banks: the 40 % bank-conflict share of the GEMM-mode layers in
        // profiles/r02_pmc_summary.json) - and the wave still reads whole 128-byte pixel chunks.
        const int l6 = e & 63, p = (e >> 6) * 8 + (l6 & 7), cq = l6 >> 3;
        const int hr = p / HW, wc = p % HW;
        const int hi = h0 - PADH + hr, wi = w0 - PADW + wc;
        a_ok[r] = p < NP && hi >= 0 && hi < a.H && wi >= 0 && wi < Win;
        if constexpr (UPCAT) {
            a_off[r] = a_ok[r] ? (unsigned)(((hi >> 1) * (Win >> 1) + (wi >> 1)) * a.cin_up + cq * 4) : 0u;
            a_off2[r] = a_ok[r] ? (unsigned)((hi * Win + wi) * (a.cin - a.cin_up) + cq * 4) : 0u;
        } else {
            a_off[r] = a_ok[r] ? (unsigned)((hi * Win + wi) * a.cin + cq * 4) : 0u;
        }
        a_lds[r] = p < NP ? ((cq >> 1) * NPPAD + p) * 2 + (cq & 1) : -1;
    }
    const float *ximg = UPCAT ? a.x + (size_t)img * (a.H >> 1) * (Win >> 1) * a.cin_up : a.x + img_base;
    const float *ximg2 = UPCAT ? a.x2 + (size_t)img * a.H * Win * (a.cin - a.cin_up) : nullptr;
    const int nch_up = UPCAT ? a.cin_up / KC : 0;
    const f32x4 *wt4 = reinterpret_cast<const f32x4 *>(a.wfrag) + (size_t)nt * (NT / 16) * WU;      // WU x 16 B per cout tile
    const size_t chunk_stride = (size_t)a.cout16 * WU, tap_stride = (size_t)nchunks * chunk_stride;
    f32x4 ra[A_LD], rb[B_LD];
    auto ldA = [&](int chunk) {
#pragma unroll
        for (int r = 0; r < A_LD; ++r) {
            if constexpr (PRE_IN) {
                ra[r] = a_ok[r] ? reinterpret_cast<const f32x4 *>(ximg)[a_off[r] + chunk * 8] : (f32x4){0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const float *src = ximg + chunk * KC + a_off[r];
            if constexpr (UPCAT) { if (chunk >= nch_up) src = ximg2 + (chunk - nch_up) * KC + a_off2[r]; }
            ra[r] = a_ok[r] ? *reinterpret_cast<const f32x4 *>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stA = [&](int abuf = 0) {                      // split into the three bf16 planes on the way into LDS
        u32x2 *base = reinterpret_cast<u32x2 *>(ldsA + abuf * A_U);
#pragma unroll
        for (int r = 0; r < A_LD; ++r)
            if (a_lds[r] >= 0) {
                if constexpr (PRE_IN) {                 // already split by the producer: a 16-byte copy
                    ldsA[abuf * A_U + a_lds[r]] = __builtin_bit_cast(u32x4, ra[r]);
                    continue;
                }
                if constexpr (SPL == 3) {
                    u32x2 hi, mid, lo;
                    split3_quad(ra[r], hi, mid, lo);
                    base[a_lds[r]] = hi;
                    base[a_lds[r] + 2 * PS] = mid;
                    base[a_lds[r] + 4 * PS] = lo;
                } else {
                    u32x2 hi, lo;
                    split2_quad(ra[r], hi, lo);
                    base[a_lds[r]] = hi;
                    base[a_lds[r] + 2 * PS] = lo;
                }
            }
    };
    auto ldB = [&](const f32x4 *tile) {
#pragma unroll
        for (int r = 0; r < B_LD; ++r) {
            const int f = tid + r * NTHR;
            if (B_F4 % NTHR == 0 || f < B_F4) rb[r] = tile[f];
        }
    };
    auto stB = [&](int buf) {
#pragma unroll
        for (int r = 0; r < B_LD; ++r) {
            const int f = tid + r * NTHR;
            if (B_F4 % NTHR == 0 || f < B_F4) ldsB[buf * B_F4 + f] = __builtin_bit_cast(u32x4, rb[r]);
        }
    };
    // f16x2 3x3: the tap-by-tap loops walk the taps column by column (dx outer, dy inner) - the order in which the row-streaming
    // loop adds them up - so that every f16x2 build of a layer gives the same bits whatever loop and tile it uses
    auto tap_w = [](int t) { return (SPL == 2 && KH == 3 && KW == 3) ? (t % 3) * 3 + t / 3 : t; };
    constexpr bool ROWS = BDIR && KH == 3 && KW == 3 && SPL == 2;
    if constexpr (ROWS) {
    // ---- halo-row streaming (3x3, f16x2, weights straight from L2).  The tap-by-tap loop reads every A fragment once per
    // tap - 9 x MS reads of (h, l) per chunk, each issued right before the MFMAs that need it, so a wave sits out the LDS
    // latency MS times per step (profiles/r03_conv_tile_trace.txt: 0.67 us per step against 0.40 us of MFMA issue).  Here a
    // chunk is walked column offset by column offset (dx), and inside one dx halo row by halo row: the fragment of halo row j
    // (pixels j, dx .. dx + 15) is the A operand of output row j - dy for all three dy, so it is read ONCE and used by up to
    // 3 x 3 NS MFMAs: 3 (TH + 2) reads per chunk instead of 9 TH (2.1x fewer for TH = 5), and each read is requested
    // kRowAhead units before its first use (a ring of register pairs).  The three taps (dy, dx) of a column offset are
    // needed together: two sets of 3 taps of weight fragments, the set of the next (chunk, dx) group requested while the
    // current one computes (~1400 cycles of MFMA issue ahead).  Accumulation order per output element: chunk, dx, dy.
    const u32x4 *wq = reinterpret_cast<const u32x4 *>(wt4) + (wn * NS) * WU + lane;
    constexpr int NROW = TH + 2, NU = NROW * MWW, AH = kRowAhead, RING = AH + 1;
    static_assert(NU >= 3 && AH >= 1 && AH <= NU, "row streaming: units per column offset");
    u32x4 bw[2][3][NS][2], ar[RING][2];
    auto ldW = [&](u32x4 (&dst)[NS][2], const u32x4 *tile) {
#pragma unroll
        for (int n = 0; n < NS; ++n) { dst[n][0] = tile[n * WU]; dst[n][1] = tile[n * WU + 64]; }
    };
    auto wtap = [&](int chunk, int tap) {                // (past the last chunk: the last one again - read, never used)
        return wq + (size_t)tap * tap_stride + (size_t)min(chunk, nchunks - 1) * chunk_stride;
    };
    auto rdA = [&](u32x4 (&dst)[2], int abuf, int dx, int unit) {
        const u32x4 *p = ldsA + abuf * A_U + (unit / MWW) * HW + dx + li + kq * NPPAD + (wm * MWW + unit % MWW) * 16;
        dst[0] = p[0]; dst[1] = p[PS];
    };
    if constexpr (!FUSE1) ldA(0);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) ldW(bw[0][dy], wtap(0, dy * 3));
    if constexpr (FUSE1) {
        float *patch = reinterpret_cast<float *>(lds + 2 * A_U);
        const LineDesc ld = a.f1_lines[img];
        const uint8_t *src = a.f1_crops + ld.offset;
        const int src_h = a.f1_src_h > 0 ? a.f1_src_h : a.H;
        // conv1's input patch, as conv1_u8_kernel stages it, one more ring of pixels: the crop's BYTES as floats (conv1_x_frag)
        constexpr int F1_IT = (F1_N + NTHR - 1) / NTHR;
        unsigned char pb[F1_IT];
#pragma unroll
        for (int it = 0; it < F1_IT; ++it) {
            const int e = tid + it * NTHR, c = e % 3, p = e / 3, wc = p % F1_PW, hr = p / F1_PW;
            const int hi = h0 - 2 + hr, wi = w0 - 2 + wc, xc = wi - ld.pad_left;
            const bool pok = e < F1_N && hi >= 0 && hi < src_h && wi >= 0 && wi < Win && xc >= 0 && xc < ld.width;
            // (unconditional loads from a clamped address, then a select: a conditional load is a branch, and the branches put one
            // memory round trip after the other in front of every tile)
            const unsigned char got = *(pok ? src + ((size_t)hi * ld.width + xc) * 3 + c : a.f1_crops);      // (the pool's first byte always exists)
            pb[it] = pok ? got : (unsigned char)0;
        }
        // conv1's weights (A operand) of all four channel tiles (its bias rides in k slot 27: conv1_koff)
        u32x4 xwh[4], xwl[4];
#pragma unroll
        for (int nt1 = 0; nt1 < 4; ++nt1) {
            xwh[nt1] = reinterpret_cast<const u32x4 *>(a.f1_w)[(nt1 * 2 + 0) * 64 + lane];
            xwl[nt1] = reinterpret_cast<const u32x4 *>(a.f1_w)[(nt1 * 2 + 1) * 64 + lane];
        }
#pragma unroll
        for (int it = 0; it < F1_IT; ++it) {
            const int e = tid + it * NTHR;
            if (e < F1_N) patch[e] = (float)pb[it] * (1.0f / 256.0f);     // (0 outside the crop)
        }
        // behind the patch: the constant 1 of the bias slot, then zeros - among them a whole 3 x 3 x 3 neighbourhood for the
        // pixels outside the image (conv2's zero padding: their inputs AND their bias slot are zero, so the MFMAs give 0)
        if (tid < F1_Z) patch[F1_N + tid] = tid == 0 ? 1.f : 0.f;
        Conv1Slots koff8;
        conv1_koff(koff8, kq, F1_PW, F1_N);
        unsigned f1max = 0u;                            // range guard of conv1's activation (never stored in this mode)
        __syncthreads();
        // the waves share the pixel tiles (16 halo pixels each), every wave computes all 64 channels of its tiles
#pragma unroll
        for (int it = 0; it < (NPPAD / 16 + NWAVE - 1) / NWAVE; ++it) {      // (unrolled: the iterations are independent chains gather -> MFMA -> split -> LDS)
            const int mt = wave + it * NWAVE;
            if (mt >= NPPAD / 16) break;
            const int px = mt * 16 + li, pa = min(px, NP - 1);
            const int hi = h0 - 1 + pa / HW, wi = w0 - 1 + pa % HW;
            const bool inside = px < NP && hi >= 0 && hi < a.H && wi >= 0 && wi < Win;      // outside the image: conv2's zero padding
            u32x4 xh;
            conv1_x_frag(patch, inside ? ((pa / HW) * F1_PW + pa % HW) * 3 : F1_N + kConv1Tail, koff8, F1_N, inside, xh);
#pragma unroll
            for (int nt1 = 0; nt1 < 4; ++nt1) {
                const f32x4 v = conv1_relu_note(conv1_mma_f16x2(xh, xwh[nt1], xwl[nt1]), f1max);
                u32x2 hh, ll;
                split2_quad(v, hh, ll);
                // channels 16 nt1 + 4 kq .. + 3: chunk nt1 >> 1, octet 2 (nt1 & 1) + (kq >> 1), half kq & 1
                u32x2 *dst = reinterpret_cast<u32x2 *>(ldsA + (nt1 >> 1) * A_U + (2 * (nt1 & 1) + (kq >> 1)) * NPPAD + px) + (kq & 1);
                dst[0] = hh;
                dst[2 * PS] = ll;
            }
        }
        range_publish(a.f1_range, f1max, lane);
    } else {
        stA(0);
    }
    __syncthreads();
    for (int c0 = 0; c0 < nchunks; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {                    // two chunks = six groups: the weight-set parity is static
            const int chunk = c0 + u;
            if (chunk >= nchunks) break;                 // (uniform)
            const int abuf = chunk & 1;
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
                    if (q < 3) ldW(bw[par ^ 1][q], wtap(nchunk, q * 3 + ndx));
                    if (!FUSE1 && dx == 0 && q == (kRowLdaQ < NU ? kRowLdaQ : NU - 1)) ldA(chunk + 1 < nchunks ? chunk + 1 : chunk);
                    const int j = q / MWW, mw = q % MWW;
                    const u32x4 ah = ar[qq % RING][0], al = ar[qq % RING][1];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int r = j - dy;
                        if (r < 0 || r >= TH) continue;
                        const int m = r * MWW + mw;
                        u32x4 (&bc)[NS][2] = bw[par][dy];
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc2[m][n] = mfma_conv_f16<TR>(al, bc[n][0], acc2[m][n]);
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc[m][n] = mfma_conv_f16<TR>(ah, bc[n][0], acc[m][n]);
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc2[m][n] = mfma_conv_f16<TR>(ah, bc[n][1], acc2[m][n]);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // units stay in source order: reads of unit q + AH, then the MFMAs of unit q
                }
                if (!FUSE1 && dx == kRowStaDx) stA(abuf ^ 1);   // the other A buffer: its last readers passed the barrier of the previous chunk
            }
            __syncthreads();
        }
    }
    } else if constexpr (GEMM2) {
    // ---- GEMM mode, f16x2, weights through LDS: chunk c computes from buffer c & 1 while the tiles of chunk c + 1 (in
    // registers since the middle of chunk c - 1) are split / copied into the other buffer in the middle of chunk c and the
    // loads of chunk c + 2 leave right after: ONE barrier per chunk, a full chunk of MFMAs between a request and its use.
    // The A fragments of row strip m + 1 are read while strip m multiplies.
    ldA(0);
    ldB(wt4);
    stA(0);
    stB(0);
    ldA(nchunks > 1 ? 1 : 0);
    ldB(wt4 + (size_t)(nchunks > 1 ? 1 : 0) * chunk_stride);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        const u32x4 *Ab = ldsA + cur * A_U + li + kq * NPPAD + wm * MWW * 16;
        const u32x4 *Bb = ldsB + cur * B_F4 + (wn * NS) * WU + lane;
        u32x4 bh[NS], bl[NS], af[2][2];
#pragma unroll
        for (int n = 0; n < NS; ++n) { bh[n] = Bb[n * WU]; bl[n] = Bb[n * WU + 64]; }
        af[0][0] = Ab[0]; af[0][1] = Ab[PS];
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            if (m + 1 < MS) {
                const int o = ((m + 1) / MWW) * HW + ((m + 1) % MWW) * 16;
                af[(m + 1) & 1][0] = Ab[o]; af[(m + 1) & 1][1] = Ab[o + PS];
            }
            const u32x4 ah = af[m & 1][0], al = af[m & 1][1];
#pragma unroll
            for (int n = 0; n < NS; ++n) acc2[m][n] = mfma_conv_f16<TR>(al, bh[n], acc2[m][n]);
#pragma unroll
            for (int n = 0; n < NS; ++n) acc[m][n] = mfma_conv_f16<TR>(ah, bh[n], acc[m][n]);
#pragma unroll
            for (int n = 0; n < NS; ++n) acc2[m][n] = mfma_conv_f16<TR>(ah, bl[n], acc2[m][n]);
            __builtin_amdgcn_sched_barrier(0);
            if (m == (MS - 1) / 2) {
                // (unconditional: after the last chunk the other buffer has no reader, and the clamped re-read keeps the
                // number of loads in flight the same in every iteration - counted s_waitcnt instead of a drain)
                stA(cur ^ 1);
                stB(cur ^ 1);
                const int c2 = min(chunk + 2, nchunks - 1);
                ldA(c2);
                ldB(wt4 + (size_t)c2 * chunk_stride);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    } else if constexpr (BDIR) {
    const u32x4 *wq = reinterpret_cast<const u32x4 *>(wt4) + (wn * NS) * WU + lane;
    // three register sets of weight fragments, rotated with the step (statically: 9 taps = 3 x 3; other tap counts unroll
    // three chunks): the set of step s + 2 is requested while step s computes - two steps (~2000 cycles) cover an L2
    // miss, one does not
    constexpr int NSETS = kBdirSets, AHEAD = NSETS - 1;
    constexpr bool APRE = SPL == 2 && MS <= 4;
    u32x4 bw[NSETS][NS][SPL];
    u32x4 apre[APRE ? 2 : 1][APRE ? MS : 1][2];
    auto ldW = [&](u32x4 (&dst)[NS][SPL], const u32x4 *tile) {
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
            for (int p = 0; p < SPL; ++p) dst[n][p] = tile[n * WU + p * 64];
    };
    // step s = chunk * NTAP + tap uses set s % NSETS; the chunk loop is unrolled U-fold so that s % NSETS is static
    constexpr int U = NTAP % NSETS == 0 ? 1 : (2 * NTAP) % NSETS == 0 ? 2 : (3 * NTAP) % NSETS == 0 ? 3 : NSETS;
    auto wstep = [&](int s_) {                           // weights of global step s_ (clamped to the last step: re-read, never used)
        const int sc = min(s_, nchunks * NTAP - 1);
        return wq + (size_t)tap_w(sc % NTAP) * tap_stride + (size_t)(sc / NTAP) * chunk_stride;
    };
    ldA(0);
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) ldW(bw[q], wstep(q));
    stA(0);
    __syncthreads();
    for (int c0 = 0; c0 < nchunks; c0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int chunk = c0 + u;
            if (U > 1 && chunk >= nchunks) break;        // (uniform)
            const bool next_chunk = chunk + 1 < nchunks;
            const int abuf = chunk & 1;
            // no conditionals around the prefetches (the last chunk re-reads itself): the compiler then counts the loads in
            // flight exactly (s_waitcnt vmcnt(n)) instead of draining the queue wherever control flow merges
            const int chunk_n = next_chunk ? chunk + 1 : chunk;
            ldA(chunk_n);
#pragma unroll
            for (int tap = 0; tap < NTAP; ++tap) {
                const int sl = u * NTAP + tap;            // step within the unrolled body: static
                ldW(bw[(sl + AHEAD) % NSETS], wstep(chunk * NTAP + tap + AHEAD));
                const int dy = tap_w(tap) / KW, dx = tap_w(tap) % KW;
                const u32x4 *Ab = ldsA + abuf * A_U + dy * HW + dx + li + kq * NPPAD + wm * MWW * 16;
                u32x4 (&bc)[NS][SPL] = bw[sl % NSETS];
                // f16x2, few row strips (the aggregation conv: 18 MFMAs per step): the A fragments of tap t + 1 are read while tap t
                // multiplies (within a chunk: the other A buffer is only valid behind the chunk's barrier)
                if constexpr (APRE) {
                    if (tap == 0) {
#pragma unroll
                        for (int m = 0; m < MS; ++m) { const int o = (m / MWW) * HW + (m % MWW) * 16; apre[0][m][0] = Ab[o]; apre[0][m][1] = Ab[o + PS]; }
                    }
                    if (tap + 1 < NTAP) {
                        const int dy1 = tap_w(tap + 1) / KW, dx1 = tap_w(tap + 1) % KW;
                        const u32x4 *An = ldsA + abuf * A_U + dy1 * HW + dx1 + li + kq * NPPAD + wm * MWW * 16;
#pragma unroll
                        for (int m = 0; m < MS; ++m) { const int o = (m / MWW) * HW + (m % MWW) * 16; apre[(tap + 1) & 1][m][0] = An[o]; apre[(tap + 1) & 1][m][1] = An[o + PS]; }
                    }
                }
#pragma unroll
                for (int m = 0; m < MS; ++m) {
                    const int o = (m / MWW) * HW + (m % MWW) * 16;
                    if constexpr (SPL == 2) {
                        const u32x4 ah = APRE ? apre[tap & 1][m][0] : Ab[o], al = APRE ? apre[tap & 1][m][1] : Ab[o + PS];
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc2[m][n] = mfma_conv_f16<TR>(al, bc[n][0], acc2[m][n]);
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc[m][n] = mfma_conv_f16<TR>(ah, bc[n][0], acc[m][n]);
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc2[m][n] = mfma_conv_f16<TR>(ah, bc[n][1], acc2[m][n]);
                    } else {
                    const u32x4 ah = Ab[o], am = Ab[o + PS], al = Ab[o + 2 * PS];
#pragma unroll
                    for (int n = 0; n < NS; ++n) acc2[m][n] = mfma16_bf16(al, bc[n][0], acc2[m][n]);
#pragma unroll
                    for (int n = 0; n < NS; ++n) acc[m][n] = mfma16_bf16(ah, bc[n][0], acc[m][n]);
#pragma unroll
                    for (int n = 0; n < NS; ++n) acc2[m][n] = mfma16_bf16(am, bc[n][1], acc2[m][n]);
#pragma unroll
                    for (int n = 0; n < NS; ++n) acc2[m][n] = mfma16_bf16(ah, bc[n][SPL - 1], acc2[m][n]);
#pragma unroll
                    for (int n = 0; n < NS; ++n) acc2[m][n] = mfma16_bf16(am, bc[n][0], acc2[m][n]);
#pragma unroll
                    for (int n = 0; n < NS; ++n) acc2[m][n] = mfma16_bf16(ah, bc[n][SPL - 2], acc2[m][n]);
                    }
                }
                // issue order of one step, as a template for the scheduler: the A reads and the weight loads of the step after
                // next spread between the MFMAs instead of bunched where the source puts them (mask 8 MFMA, 0x100 DS read, 0x20 VMEM read)
                // one A read per group of TOT / G MFMAs; NV weight loads spread over the G groups
                sched_template_dir<MS * SPL, MS * NS * NMF, NS * SPL>(std::make_integer_sequence<int, MS * SPL>{});
                if (tap == (NTAP / 2)) stA(abuf ^ 1);   // the other A buffer: its last readers passed the barrier of the previous chunk
            }
            __syncthreads();
        }
    }
    } else {
    ldA(0);
    ldB(wt4);
    stA();
    stB(0);
    __syncthreads();
    int step = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool next_chunk = chunk + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap, ++step) {
            const bool last = tap == NTAP - 1, more = !last || next_chunk;
            const int bcur = step & 1;
            if (more) ldB(wt4 + (size_t)tap_w(last ? 0 : tap + 1) * tap_stride + (size_t)(last ? chunk + 1 : chunk) * chunk_stride);
            if (tap == 0 && next_chunk) ldA(chunk + 1);
            const int dy = tap_w(tap) / KW, dx = tap_w(tap) % KW;
            const u32x4 *Ab = ldsA + dy * HW + dx + li + kq * NPPAD + wm * MWW * 16;
            const u32x4 *Bb = ldsB + bcur * B_F4 + (wn * NS) * WU + lane;
            u32x4 bh[NS], bm[NS], bl[NS];
#pragma unroll
            for (int n = 0; n < NS; ++n) { bh[n] = Bb[n * WU]; bm[n] = Bb[n * WU + 64]; bl[n] = Bb[n * WU + (SPL - 1) * 64]; }
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                const int o = (m / MWW) * HW + (m % MWW) * 16;
                if constexpr (SPL == 2) {           // bm = the scaled low plane of the weights
                    const u32x4 ah = Ab[o], al = Ab[o + PS];
#pragma unroll
                    for (int n = 0; n < NS; ++n) acc2[m][n] = mfma_conv_f16<TR>(al, bh[n], acc2[m][n]);
#pragma unroll
                    for (int n = 0; n < NS; ++n) acc[m][n] = mfma_conv_f16<TR>(ah, bh[n], acc[m][n]);
#pragma unroll
                    for (int n = 0; n < NS; ++n) acc2[m][n] = mfma_conv_f16<TR>(ah, bm[n], acc2[m][n]);
                    continue;
                }
                const u32x4 ah = Ab[o], am = Ab[o + PS], al = Ab[o + (SPL - 1) * PS];
                // The main term (hi x hi) and the five small terms run in SEPARATE accumulators, added once in the epilogue:
                // the running sum of the main term sees one rounding per 32-deep block (the fp32-MFMA chain: 32), and the
                // small terms round at their own magnitude, 2^-8 of the main one.  Measured against float64: 3x closer than
                // the fp32-MFMA chain.  Terms outermost, so consecutive MFMAs never depend on each other.
#pragma unroll
                for (int n = 0; n < NS; ++n) acc2[m][n] = mfma16_bf16(al, bh[n], acc2[m][n]);
#pragma unroll
                for (int n = 0; n < NS; ++n) acc[m][n] = mfma16_bf16(ah, bh[n], acc[m][n]);
#pragma unroll
                for (int n = 0; n < NS; ++n) acc2[m][n] = mfma16_bf16(am, bm[n], acc2[m][n]);
#pragma unroll
                for (int n = 0; n < NS; ++n) acc2[m][n] = mfma16_bf16(ah, bl[n], acc2[m][n]);
#pragma unroll
                for (int n = 0; n < NS; ++n) acc2[m][n] = mfma16_bf16(am, bh[n], acc2[m][n]);
#pragma unroll
                for (int n = 0; n < NS; ++n) acc2[m][n] = mfma16_bf16(ah, bm[n], acc2[m][n]);
            }
            sched_template_lds<(MS + NS) * SPL, MS * NS * NMF>(std::make_integer_sequence<int, (MS + NS) * SPL>{});
            if (more) stB(bcur ^ 1);
            if (last && next_chunk) {
                __syncthreads();                        // every wave has read the last tap of this chunk's halo tile
                stA();
            }
            __syncthreads();
        }
    }

    }
    if constexpr (TR) {
        // ---- epilogue for the [channel][pixel] result layout: lane (li, kq) holds channels 4 kq .. + 3 of pixel li of every 16-pixel
        // tile.  No transpose: the two 8-byte pieces of a pixel (h plane, l plane: p2_channel_bytes) go out as buffer stores.  (conv_rows.hpp sends the tile through LDS and stores whole
        // lines - conv_epilogue_staged; this kernel keeps the direct stores: it serves conv2 with conv1 in its prologue, whose pooled
        // output is a quarter of its input, and the layers of POCR_CONV_ROWS=0.)
        const int Wo = Win, Wout = Wo / POOLW;
        const unsigned pix_bytes = (unsigned)a.out_stride * 4u;
        const size_t img_bytes = (size_t)(a.Ho / POOLH) * Wout * pix_bytes;
        constexpr unsigned kOut = 0x80000000u;
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + out_base, 0, (int)(img_bytes < kOut ? img_bytes : kOut - 1), 0x00020000);
        unsigned rmax = 0u;
        unsigned lane_off[MWW];
#pragma unroll
        for (int mw = 0; mw < MWW; ++mw) {
            const int col = w0 + (wm * MWW + mw) * 16 + li;
            const bool ok = POOLW == 2 ? ((li & 1) == 0 && col + 1 < Wo) : col < Wo;
            lane_off[mw] = ok ? (unsigned)(col / POOLW) * pix_bytes + (unsigned)kq * 8u : kOut;
        }
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const int cg = nt * (NT / 16) + wn * NS + n;              // 16-channel group; this lane: channels 16 cg + 4 kq + r
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(a.bias + cg * 16 + 4 * kq);
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BN) { sc = *reinterpret_cast<const f32x4 *>(a.bn_scale + cg * 16 + 4 * kq); sh = *reinterpret_cast<const f32x4 *>(a.bn_shift + cg * 16 + 4 * kq); }
            const unsigned chan_off = (unsigned)(cg >> 1) * 128u + (unsigned)(cg & 1) * 32u;      // p2_channel_bytes(16 cg); the lane's 4 kq channels: + 8 kq (lane_off)
#pragma unroll
            for (int th = 0; th < TH; th += POOLH) {
                const int ho = (h0 + th) / POOLH;
                const bool row_ok = h0 + th < a.Ho;
                const unsigned row_off = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)ho * (unsigned)Wout * pix_bytes + chan_off));     // (wave-uniform: said explicitly, or the store becomes a loop over lanes)
#pragma unroll
                for (int mw = 0; mw < MWW; ++mw) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = apply_act(acc[th * MWW + mw][n][r] + acc2[th * MWW + mw][n][r] * (1.0f / kF16x2Scale) + bias[r], ACT);
                        if constexpr (BN) t = t * sc[r] + sh[r];
                        if constexpr (POOLH == 2) {
                            float u = apply_act(acc[(th + 1) * MWW + mw][n][r] + acc2[(th + 1) * MWW + mw][n][r] * (1.0f / kF16x2Scale) + bias[r], ACT);
                            if constexpr (BN) u = u * sc[r] + sh[r];
                            t = fmaxf(t, u);
                        }
                        if constexpr (POOLW == 2)              // the other pixel of the pair sits in the neighbouring lane
                            t = fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true)));
                        v[r] = t;
                    }
                    if constexpr (ACT == ACT_RELU && !BN) {   // non-negative: bit patterns order like signed integers (conv1_relu_note)
                        int mm = (int)rmax;
                        mm = max(max(mm, __builtin_bit_cast(int, v[0])), __builtin_bit_cast(int, v[1]));
                        mm = max(max(mm, __builtin_bit_cast(int, v[2])), __builtin_bit_cast(int, v[3]));
                        rmax = (unsigned)mm;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) range_note(rmax, v[r]);
                    }
                    u32x2 hh, ll;
                    split2_quad(v, hh, ll);
                    if (row_ok && lane_off[mw] != kOut) {          // (a branch, not an out-of-range offset: conv_epilogue_staged)
                        __builtin_amdgcn_raw_buffer_store_b64(hh, yrsrc, (int)lane_off[mw], (int)row_off, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(ll, yrsrc, (int)(lane_off[mw] + 64u), (int)row_off, 0);
                    }
                }
            }
        }
        range_publish(a.range_max, rmax, lane);
        return;
    }
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            if constexpr (SPL == 2) acc[m][n] += acc2[m][n] * (1.0f / kF16x2Scale);
            else acc[m][n] += acc2[m][n];
        }
    // ---- epilogue (identical to conv_igemm_kernel: same D layout)
    const int Wo = Win, Wout = Wo / POOLW;
    unsigned rmax = 0u;                                 // f16x2 range guard (conv_igemm.hpp: range_note)
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int co = (nt * (NT / 16) + wn * NS + n) * 16 + li;
        const float bias = a.bias[co];
        float sc = 1.f, sh = 0.f;
        if constexpr (BN) { sc = a.bn_scale[co]; sh = a.bn_shift[co]; }
        const bool co_ok = co < a.cout_valid;
#pragma unroll
        for (int th = 0; th < TH; th += POOLH) {
#pragma unroll
            for (int mw = 0; mw < MWW; ++mw) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = apply_act(acc[th * MWW + mw][n][r] + bias, ACT);
                    if constexpr (BN) t = t * sc + sh;
                    if constexpr (POOLH == 2) {
                        float u = apply_act(acc[(th + 1) * MWW + mw][n][r] + bias, ACT);
                        if constexpr (BN) u = u * sc + sh;
                        t = fmaxf(t, u);
                    }
                    v[r] = t;
                    if constexpr (SPL == 2) range_note(rmax, t);
                }
                const int ho = (h0 + th) / POOLH;
                const int wbase = w0 + (wm * MWW + mw) * 16 + kq * 4;
                float *yrow = a.y + out_base + ((size_t)ho * Wout) * a.out_stride + co;
                if constexpr (PRE_OUT) {                // P2 layout: every value split here, once (cout % 32 == 0, all valid)
                    char *prow = reinterpret_cast<char *>(a.y + out_base + ((size_t)ho * Wout) * a.out_stride);
                    const size_t pix_bytes = (size_t)a.out_stride * 4;
                    if constexpr (POOLW == 2) {
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr) {
                            const int wc = wbase + 2 * rr;
                            if (h0 + th < a.Ho && wc + 1 < Wo) {
                                _Float16 hh, ll;
                                split2_scalar(fmaxf(v[2 * rr], v[2 * rr + 1]), hh, ll);
                                _Float16 *d = reinterpret_cast<_Float16 *>(prow + (size_t)(wc / 2) * pix_bytes + p2_channel_bytes(co));
                                d[0] = hh; d[32] = ll;
                            }
                        }
                    } else {
                        quad_transpose(v, lane);
                        const int wc = wbase + (li & 3), c4 = co - (li & 3);
                        if (wc < Wo && h0 + th < a.Ho) {
                            u32x2 hh, ll;
                            split2_quad((f32x4){v[0], v[1], v[2], v[3]}, hh, ll);
                            u32x2 *d = reinterpret_cast<u32x2 *>(prow + (size_t)wc * pix_bytes + p2_channel_bytes(c4));
                            d[0] = hh; d[8] = ll;
                        }
                    }
                    continue;
                }
                if constexpr (POOLW == 2) {
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int wc = wbase + 2 * rr;
                        if (co_ok && h0 + th < a.Ho && wc + 1 < Wo) yrow[(size_t)(wc / 2) * a.out_stride] = fmaxf(v[2 * rr], v[2 * rr + 1]);
                    }
                } else {
                    quad_transpose(v, lane);
                    const int wc = wbase + (li & 3);
                    float *dst = yrow - (li & 3) + (size_t)wc * a.out_stride;
                    const int c4 = co - (li & 3);
                    if (wc < Wo && h0 + th < a.Ho) {
                        if (c4 + 3 < a.cout_valid && (a.out_stride & 3) == 0) {
                            *reinterpret_cast<f32x4 *>(dst) = (f32x4){v[0], v[1], v[2], v[3]};
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (c4 + k < a.cout_valid) dst[k] = v[k];
                        }
                    }
                }
            }
        }
    }
    if constexpr (SPL == 2) range_publish(a.range_max, rmax, lane);
}

}  // namespace pocr
