// conv1_u8.hpp — first layer of the backbone fused with the batch assembly and normalisation:
//   zero-pad batch assembly      pero_ocr/ocr_engine/line_ocr_engine.py:121-127
//   uint8 -> float32 / 255.0     pero_ocr/ocr_engine/pytorch_ocr_engine.py:61-62
//   conv 3->64, 3x3 pad 1, ReLU  first VGG block (pero_ocr/ocr_engine/transformer.py:86-110)
// The crops stay uint8 and un-padded in HBM (68 KB per 40x576 line instead of 276 KB as fp32); a workgroup
// stages the (4+2) x (32+2) x 3 halo of its pixel tile into LDS through the 256-entry i/255.0f table
// (bit-exact with torch's true division), zero where the padded row has no crop pixel, and feeds
// v_mfma_f32_16x16x4_f32 straight from that halo: the im2col matrix (K = 27, padded to 32) is never
// built - MFMA operand k = (ky*3 + kx)*3 + c is just a per-lane constant offset into the halo.
// Output-bound: 5.9 MB of fp32 NHWC per 40x576 line.
// F16X2 (the recogniser's default arithmetic): the same contraction as ONE 32-deep f16x2 product block - three
// v_mfma_f32_16x16x32_f16 with the weights as the A operand (conv_bf16x3.hpp: conv1_mma_f16x2) - and in the default mode this
// kernel does not run at all: conv2 computes conv1 for its own halo tile (conv3x3_bf16x3_kernel FUSE1); it then serves
// POCR_NO_FUSE12=1, the non-P2 builds and pocr_debug_read(0).
#pragma once
#include <hip/hip_runtime.h>
#include "conv_igemm.hpp"
#include "conv_bf16x3.hpp"

namespace pocr {

struct Conv1Args {
    const uint8_t *crops;
    const LineDesc *lines;
    const float *lut;            // i / 255.0f
    const float *wfrag;          // [k/16 = 2][cout/16 = 4][lane][4], k = (ky*3+kx)*3 + c, zero for k >= 27
    const float *bias;           // [64]
    float *y;                    // ragged NHWC fp32, line i at out_off[i]
    const PixelTile *tiles;
    const int32_t *line_w;       // padded width of every line
    const int64_t *out_off;
    int32_t H, n_ptiles;
    int32_t src_h;               // rows the crops really have (0: = H); rows [src_h, H) are zero padding (layout network pages)
    const void *w1x2;            // F16X2: [cout/16 = 4][plane h, l][lane] x 8 f16: k = 8 (lane >> 4) + j, cout = 16 s + (lane & 15); k = 27: the bias (its input is the constant 1), zero for k > 27
    unsigned *range_max;         // f16x2 range guard (conv_igemm.hpp: range_publish) or NULL
};

// P2OUT: the output is written in the pre-split f16x2 layout conv2 stages by plain copies (conv_bf16x3.hpp, "P2").
// F16X2: the arithmetic above (the recogniser in its default mode) instead of fp32 MFMA.
template <bool P2OUT = false, bool F16X2 = false>
__global__ __launch_bounds__(256) void conv1_u8_kernel(Conv1Args a) {
    constexpr int TH = 4, TW = 32, HH = TH + 2, HW = TW + 2, NH = HH * HW * 3;
    __shared__ float halo[NH + 4];                    // [row][col][c]; halo[NH] = 0 backs the k >= 27 padding
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    // XCD-aware order as in conv_igemm_kernel (contiguous tile ranges per XCD: neighbouring tiles share halo bytes)
    int b = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = b & 7, k = b >> 3;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const PixelTile pt = a.tiles[b];
    const int img = pt.line, h0 = (pt.ht_wt >> 16) * TH, w0 = (pt.ht_wt & 0xffff) * TW;
    const int Wp = a.line_w[img];
    const LineDesc ld = a.lines[img];
    const uint8_t *src = a.crops + ld.offset;
    const int src_h = a.src_h > 0 ? a.src_h : a.H;
    for (int e = tid; e < NH; e += 256) {
        const int c = e % 3, p = e / 3, wc = p % HW, hr = p / HW;
        const int hi = h0 - 1 + hr, wi = w0 - 1 + wc, xc = wi - ld.pad_left;
        float v = 0.f;
        if (hi >= 0 && hi < src_h && wi >= 0 && wi < Wp && xc >= 0 && xc < ld.width)
            v = F16X2 ? (float)src[((size_t)hi * ld.width + xc) * 3 + c] * (1.0f / 256.0f)      // byte / 256: the 256 / 255 sits in the f16x2 weights (conv_bf16x3.hpp: conv1_x_frag)
                      : a.lut[src[((size_t)hi * ld.width + xc) * 3 + c]];
        halo[e] = v;
    }
    if (tid < 4) halo[NH + tid] = (F16X2 && tid == 0) ? 1.f : 0.f;        // the tail conv1_x_frag expects: the bias slot's constant 1, zeros
    // weights of this wave's 16 output channels and the per-lane halo offsets of its MFMA k slots
    f32x4 wb[2];
    int koff[2][4];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
        wb[kg] = reinterpret_cast<const f32x4 *>(a.wfrag)[(kg * 4 + wave) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 16 * kg + 4 * kq + j;
            const int tap = k / 3, c = k - 3 * tap;
            koff[kg][j] = k < 27 ? ((tap / 3) * HW + tap % 3) * 3 + c : -1;
        }
    }
    u32x4 xwh = {0u, 0u, 0u, 0u}, xwl = {0u, 0u, 0u, 0u};
    Conv1Slots koff8 = {};
    if constexpr (F16X2) {        // (conv_bf16x3.hpp: conv1_mma_f16x2 - weights are the A operand, a lane gets channels 16 wave + 4 kq + r of pixel li)
        xwh = reinterpret_cast<const u32x4 *>(a.w1x2)[(wave * 2 + 0) * 64 + lane];
        xwl = reinterpret_cast<const u32x4 *>(a.w1x2)[(wave * 2 + 1) * 64 + lane];
        // (the bias rides in k slot 27 of the weights: conv_bf16x3.hpp, conv1_koff)
        conv1_koff(koff8, kq, HW, NH);
    }
    __syncthreads();
    const float bias = a.bias[wave * 16 + li];
    float *yimg = a.y + a.out_off[img];
    unsigned rmax = 0u;                               // f16x2 range guard (conv_igemm.hpp)
#pragma unroll
    for (int th = 0; th < TH; ++th) {
#pragma unroll
        for (int mw = 0; mw < 2; ++mw) {
            const int base = (th * HW + mw * 16 + li) * 3;     // halo element of tap (0,0), channel 0 for this lane's pixel
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (F16X2) {
                u32x4 xh;
                conv1_x_frag(halo, base, koff8, NH, true, xh);
                const f32x4 d = conv1_mma_f16x2(xh, xwh, xwl);
                const int ho = h0 + th, wc = w0 + mw * 16 + li;
                if (ho >= a.H || wc >= Wp) continue;
                const f32x4 v = conv1_relu_note(d, rmax);
                if constexpr (P2OUT) {
                    u32x2 hh, ll;
                    split2_quad(v, hh, ll);
                    u32x2 *dd = reinterpret_cast<u32x2 *>(reinterpret_cast<char *>(yimg + ((size_t)ho * Wp + wc) * 64) + p2_channel_bytes(wave * 16 + 4 * kq));
                    dd[0] = hh; dd[8] = ll;
                } else {
                    *reinterpret_cast<f32x4 *>(yimg + ((size_t)ho * Wp + wc) * 64 + wave * 16 + 4 * kq) = v;
                }
                continue;
            } else {
#pragma unroll
            for (int kg = 0; kg < 2; ++kg)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float av = halo[koff[kg][j] >= 0 ? base + koff[kg][j] : NH];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wb[kg][j], acc, 0, 0, 0);
                }
            }
            const int ho = h0 + th;
            if (ho >= a.H) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float t = acc[r] + bias; v[r] = t > 0.f ? t : 0.f; }
            quad_transpose(v, lane);                     // now: pixel 4*kq + (li & 3), channels 16*wave + 4*(li >> 2) + 0..3
            const int wc = w0 + mw * 16 + kq * 4 + (li & 3);
            if constexpr (P2OUT) {
                if (wc < Wp) {
                    u32x2 hh, ll;
                    split2_quad((f32x4){v[0], v[1], v[2], v[3]}, hh, ll);
                    u32x2 *d = reinterpret_cast<u32x2 *>(reinterpret_cast<char *>(yimg + ((size_t)ho * Wp + wc) * 64) +
                                                         p2_channel_bytes(wave * 16 + 4 * (li >> 2)));
                    d[0] = hh; d[8] = ll;
                }
                continue;
            }
            if (wc < Wp)
                *reinterpret_cast<f32x4 *>(yimg + ((size_t)ho * Wp + wc) * 64 + wave * 16 + 4 * (li >> 2)) =
                    (f32x4){v[0], v[1], v[2], v[3]};
        }
    }
    if constexpr (F16X2) range_publish(a.range_max, rmax, lane);
}

}  // namespace pocr
