// conv_igemm.hpp — implicit-GEMM convolution on the gfx950 matrix cores, exact fp32.
//
// Replaces the stock aten::conv2d / relu / leaky_relu / max_pool2d / batch_norm calls the
// reference's opaque TorchScript model executes (pero_ocr/ocr_engine/pytorch_ocr_engine.py:66-69;
// layer topology pero_ocr/ocr_engine/transformer.py:51-72,86-144,351-355).
//
// GEMM view:  D[pixel][cout] = sum_{tap, cin} X[pixel + tap][cin] * W[tap][cin][cout]
//   M = output pixels (16 consecutive columns of one image row per MFMA row-tile),
//   N = output channels, K = taps x input channels.
// MFMA: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate; bitwise an fmaf chain), 32-cycle issue.
//
// Data layout in HBM:  activations NHWC fp32; weights pre-arranged in "fragment order"
//   wfrag[tap][cin/16][cout/16][lane 0..63][j 0..3] = W[tap][cin = 16*g + 4*(lane>>4) + j][cout = 16*s + (lane&15)]
// so that one 16-byte load per lane yields the B operands of four consecutive MFMA k-steps.
//
// LDS:  A tile  [2][KC/4][NPPAD][4]  (input halo tile, channel-quad planes; NPPAD % 16 == 0 makes the
//                                    ds_read_b128 of 16 consecutive pixels x 4 k-lanes conflict-free)
//       B tile  [2][KC/16][NT/16][64][4] (the fragment-order slice for one (chunk, tap) step)
// Pipeline: one barrier per (chunk, tap) step; the global loads for step s+1 are issued before the
// MFMAs of step s and written to the other LDS buffer after them.
//
// Epilogue (fused): + bias, ReLU / LeakyReLU(0.01), optional BatchNorm affine (eval), optional
// max-pool (2,2) or (2,1), store NHWC.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pocr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };
enum { STAGE_F32_NHWC = 0, STAGE_U8_LINES = 1 };

struct LineDesc {            // one text line of a staged chunk (STAGE_U8_LINES)
    int64_t offset;          // byte offset of the crop [H, width, 3] inside the crop pool
    int32_t width;           // crop width in pixels
    int32_t pad_left;        // x position of the crop inside the padded row
};

struct ConvArgs {
    const float *x;          // input NHWC [n][H][W][cin]            (STAGE_F32_NHWC)
    const uint8_t *crops;    // crop pool                             (STAGE_U8_LINES)
    const LineDesc *lines;   //                                       (STAGE_U8_LINES)
    const float *lut;        // 256-entry u8 -> f32 table (i / 255.0f) (STAGE_U8_LINES)
    const float *wfrag;      // fragment-order weights
    const float *bias;       // [cout16*16]
    const float *bn_scale;   // [cout16*16] or null: y = act(conv) * scale + shift
    const float *bn_shift;
    float *y;                // output [n][Hout][Wout][out_stride]
    int32_t n, H, W;         // input image dims
    int32_t Ho, Wo;          // conv output dims BEFORE pooling
    int32_t cin;             // multiple of KC
    int32_t cout16;          // number of 16-wide output-channel groups in wfrag (multiple of NT/16)
    int32_t cout_valid;      // channels actually stored
    int32_t out_stride;      // floats per output pixel
    int32_t tiles_w, tiles_h, tiles_n;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_LEAKY) return v > 0.f ? v : v * 0.01f;
    return v;
}

template <int KH, int KW, int PADH, int PADW, int TH, int MW, int NS, int NWAVE, int KC,
          int POOLH, int POOLW, int ACT, bool BN, int STAGER>
__global__ __launch_bounds__(NWAVE * 64) void conv_igemm_kernel(ConvArgs a) {
    constexpr int TW = 16 * MW;
    constexpr int MS = TH * MW;                 // 16-pixel row-tiles per workgroup (every wave holds all of them)
    constexpr int NT = NS * NWAVE * 16;         // output channels per workgroup
    constexpr int NTHR = NWAVE * 64;
    constexpr int HH = TH + KH - 1, HW = TW + KW - 1;
    constexpr int NP = HH * HW;
    constexpr int NPPAD = (NP + 15) / 16 * 16;
    constexpr int CQ = KC / 4;                  // channel quads per chunk
    constexpr int KG = KC / 16;                 // 16-channel groups per chunk
    constexpr int NTAPS = KH * KW;
    constexpr int A_F4 = CQ * NPPAD;            // float4 per A buffer
    constexpr int B_F4 = KG * (NT / 16) * 64;   // float4 per B buffer
    constexpr int A_LD = (CQ * NP + NTHR - 1) / NTHR;   // float4 loads per thread per chunk
    constexpr int B_LD = (B_F4 + NTHR - 1) / NTHR;
    static_assert(POOLH == 1 || TH % 2 == 0, "H-pool needs an even tile height");
    static_assert(KC % 16 == 0, "KC must be a multiple of 16");

    __shared__ f32x4 lds[2 * A_F4 + 2 * B_F4];
    f32x4 *ldsA = lds;
    f32x4 *ldsB = lds + 2 * A_F4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;

    // tile coordinates: ntile fastest so the workgroups sharing one input tile run together
    int b = blockIdx.x;
    const int nt = b % a.tiles_n; b /= a.tiles_n;
    const int wt = b % a.tiles_w; b /= a.tiles_w;
    const int ht = b % a.tiles_h; b /= a.tiles_h;
    const int img = b;
    const int h0 = ht * TH, w0 = wt * TW;

    f32x4 acc[MS][NS];
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[m][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.cin / KC;
    const int nsteps = nchunks * NTAPS;
    const size_t img_base = (size_t)img * a.H * a.W * a.cin;

    f32x4 ra[A_LD], rb[B_LD];

    auto load_A = [&](int chunk) {
        const int c0 = chunk * KC;
#pragma unroll
        for (int r = 0; r < A_LD; ++r) {
            const int e = tid + r * NTHR;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e < CQ * NP) {
                const int cq = e % CQ, p = e / CQ;
                const int hr = p / HW, wc = p % HW;
                const int hi = h0 - PADH + hr, wi = w0 - PADW + wc;
                if constexpr (STAGER == STAGE_F32_NHWC) {
                    if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W)
                        v = *reinterpret_cast<const f32x4 *>(a.x + img_base + ((size_t)hi * a.W + wi) * a.cin + c0 + cq * 4);
                } else {
                    // conv1: build the im2col row of pixel (hi, wi) from the u8 crop on the fly.
                    // "channel" k = (ky*3 + kx)*3 + c for k < 27, zero above.  u8 -> f32 through the
                    // i/255.0f table (pytorch_ocr_engine.py:61); the zero padding of the batch
                    // assembly (line_ocr_engine.py:121-123) and of the conv itself are both 0.0f.
                    const LineDesc ld = a.lines[img];
                    const uint8_t *src = a.crops + ld.offset;
                    float t4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = cq * 4 + j;
                        float val = 0.f;
                        if (k < 27) {
                            const int tap = k / 3, c = k - tap * 3;
                            const int yy = hi + tap / 3 - 1, xx = wi + tap % 3 - 1;
                            const int xc = xx - ld.pad_left;
                            if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W && xc >= 0 && xc < ld.width)
                                val = a.lut[src[((size_t)yy * ld.width + xc) * 3 + c]];
                        }
                        t4[j] = val;
                    }
                    v = (f32x4){t4[0], t4[1], t4[2], t4[3]};
                }
            }
            ra[r] = v;
        }
    };
    auto store_A = [&](int buf) {
#pragma unroll
        for (int r = 0; r < A_LD; ++r) {
            const int e = tid + r * NTHR;
            if (e < CQ * NP) {
                const int cq = e % CQ, p = e / CQ;
                ldsA[buf * A_F4 + cq * NPPAD + p] = ra[r];
            }
        }
    };
    auto load_B = [&](int chunk, int tap) {
        const int g0 = chunk * KG;
#pragma unroll
        for (int r = 0; r < B_LD; ++r) {
            const int f = tid + r * NTHR;
            if (B_F4 % NTHR == 0 || f < B_F4) {
                const int kg = f / (NT * 4), rr = f % (NT * 4);
                const size_t base = (((size_t)tap * (a.cin / 16) + g0 + kg) * a.cout16 + (size_t)nt * (NT / 16)) * 64;
                rb[r] = reinterpret_cast<const f32x4 *>(a.wfrag)[base + rr];
            }
        }
    };
    auto store_B = [&](int buf) {
#pragma unroll
        for (int r = 0; r < B_LD; ++r) {
            const int f = tid + r * NTHR;
            if (B_F4 % NTHR == 0 || f < B_F4) ldsB[buf * B_F4 + f] = rb[r];
        }
    };

    load_A(0);
    load_B(0, 0);
    store_A(0);
    store_B(0);
    __syncthreads();

    int chunk = 0, tap = 0, abuf = 0;
    for (int s = 0; s < nsteps; ++s) {
        // ---- prefetch step s+1 into registers
        int ntap = tap + 1, nchunk = chunk;
        if (ntap == NTAPS) { ntap = 0; nchunk = chunk + 1; }
        const bool more = (s + 1 < nsteps);
        const bool newA = more && (ntap == 0);
        if (more) load_B(nchunk, ntap);
        if (newA) load_A(nchunk);

        // ---- MFMAs of step s
        const int dy = tap / KW, dx = tap % KW;
        const f32x4 *Ab = ldsA + abuf * A_F4 + dy * HW + dx + li;
        const f32x4 *Bb = ldsB + (s & 1) * B_F4 + (wave * NS) * 64 + lane;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            f32x4 af[MS], bf[NS];
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                const int th = m / MW, mw = m % MW;
                af[m] = Ab[(kg * 4 + kq) * NPPAD + th * HW + mw * 16];
            }
#pragma unroll
            for (int n = 0; n < NS; ++n) bf[n] = Bb[(kg * (NT / 16) + n) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MS; ++m)
#pragma unroll
                    for (int n = 0; n < NS; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bf[n][j], acc[m][n], 0, 0, 0);
        }

        // ---- commit the prefetched tiles to the other LDS buffers
        if (more) store_B((s + 1) & 1);
        if (newA) { store_A(abuf ^ 1); }
        __syncthreads();
        if (newA) abuf ^= 1;
        tap = ntap; chunk = nchunk;
    }

    // ---- epilogue.  D layout: col = lane & 15 (cout), row = (lane >> 4) * 4 + reg (pixel).
    const int Hout = a.Ho / POOLH, Wout = a.Wo / POOLW;
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int co = (nt * (NT / 16) + wave * NS + n) * 16 + li;
        const float bias = a.bias[co];
        float sc = 1.f, sh = 0.f;
        if constexpr (BN) { sc = a.bn_scale[co]; sh = a.bn_shift[co]; }
        const bool co_ok = co < a.cout_valid;
#pragma unroll
        for (int th = 0; th < TH; th += POOLH) {
#pragma unroll
            for (int mw = 0; mw < MW; ++mw) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = apply_act(acc[th * MW + mw][n][r] + bias, ACT);
                    if constexpr (BN) t = t * sc + sh;
                    if constexpr (POOLH == 2) {
                        float u = apply_act(acc[(th + 1) * MW + mw][n][r] + bias, ACT);
                        if constexpr (BN) u = u * sc + sh;
                        t = fmaxf(t, u);
                    }
                    v[r] = t;
                }
                const int ho = (h0 + th) / POOLH;
                const int wbase = w0 + mw * 16 + kq * 4;           // conv-output column of reg 0
                if (!co_ok || h0 + th >= a.Ho) continue;
                float *yrow = a.y + (((size_t)img * Hout + ho) * Wout) * a.out_stride + co;
                if constexpr (POOLW == 2) {
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int wc = wbase + 2 * rr;
                        if (wc + 1 < a.Wo) yrow[(size_t)(wc / 2) * a.out_stride] = fmaxf(v[2 * rr], v[2 * rr + 1]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int wc = wbase + r;
                        if (wc < a.Wo) yrow[(size_t)wc * a.out_stride] = v[r];
                    }
                }
            }
        }
    }
}

}  // namespace pocr
