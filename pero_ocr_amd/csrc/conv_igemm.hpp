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
// Main loop: one barrier per (chunk, tap) step, LDS double-buffered.  PIPE_INTERLEAVED (shipped for the
// 3x3 / aggregation / GEMM layers) unrolls the taps and issues the next step's global loads and LDS
// writes one at a time between MFMAs; PIPE_PLAIN is the straightforward loop (conv1's u8 stager).
//
// Epilogue (fused): + bias, ReLU / LeakyReLU(0.01), optional BatchNorm affine (eval), optional
// max-pool (2,2) or (2,1), store NHWC.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pocr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };
enum { STAGE_F32_NHWC = 0, STAGE_UPCAT = 1 };      // the u8 line stager of the first layer lives in conv1_u8.hpp
// STAGE_UPCAT (U-Net decoder convs of the layout network, parsenet.hpp): the conv input is the VIRTUAL tensor
// cat([nearest-upsample-x2(x), x2], channels) - channels [0, cin_up) come from x at half resolution, pixel (h/2, w/2),
// channels [cin_up, cin) from the skip tensor x2 at full resolution; neither the upsampled nor the concatenated tensor
// is ever materialised (a 16-channel staging chunk lies entirely inside one of the two sources).
// main-loop variants (template parameter PIPE).  ABL3 is an ablation mask used only by tools/conv_bench.hip
// (1 no global loads, 2 no LDS writes, 4 no ds_reads, 8 no barrier, 16 loads waited for at the step end).
enum { PIPE_PLAIN = 0, PIPE_INTERLEAVED = 3, PIPE_DEEP = 4, PIPE_GLDS = 5, PIPE_BREG = 6, PIPE_DEEP3 = 7 };
// PIPE_GLDS = PIPE_INTERLEAVED with the weight tile copied HBM/L2 -> LDS by the load unit itself
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write; the tile is already lane-linear).
// PIPE_BREG: the weights never touch LDS.  In fragment order a wave's B operands of one step are NS 16-byte
// loads per lane of data no other wave needs, so they go straight from L2 into registers, requested two
// steps ahead (three register sets, statically rotated: needs KH*KW % 3 == 0).  Only the input halo tile
// is shared through LDS, which leaves ONE barrier per 16-channel chunk instead of one per tap.

struct LineDesc {            // one text line of a staged chunk (read by conv1_u8_kernel)
    int64_t offset;          // byte offset of the crop [H, width, 3] inside the crop pool
    int32_t width;           // crop width in pixels
    int32_t pad_left;        // x position of the crop inside the padded row
};

struct PixelTile { int32_t line; int32_t ht_wt; };     // ht_wt = (h-tile << 16) | w-tile

// Padding columns.  Left and right of a line's crop the padded row is zero, and far enough from the crop and from
// the row ends (beyond the receptive field) every conv layer's output there is the SAME column vector for every
// line: a function of the weights and the row index only.  Those vectors are computed once per engine (a zero
// line through the conv stack); pixel tiles lying entirely in such a region are left out of the tile tables and
// pad_fill_kernel writes the constant instead - the very values the conv kernels would produce (each output
// pixel is an independent fmaf chain over identical inputs).  Transformer-engine batches are padded to >= 1088
// columns whatever the crop width, so this removes up to half of their conv work.
struct FillSeg { int32_t layer, line, c0, c1; };        // output columns [c0, c1) of `line` in conv layer `layer`

struct FillArgs {
    float *act[9];               // conv outputs (ragged NHWC)
    const float *cvec[9];        // constant column of every layer: [H_out][cout]
    const int64_t *out_off[9];   // element offset of every line in act[l]
    const int32_t *lvl_w[3];     // line widths at the three pooling levels
    int32_t lvl_out[9], h_out[9], cout[9];
    const FillSeg *segs;
    int32_t only_layer;          // -1: every layer but `skip_layer` (-1: none); >= 0: that layer alone
    int32_t skip_layer;
};

__global__ __launch_bounds__(256) void pad_fill_kernel(FillArgs a) {
    const FillSeg sg = a.segs[blockIdx.x];
    const int l = sg.layer, h = blockIdx.y;
    if (h >= a.h_out[l] || l == a.skip_layer || (a.only_layer >= 0 && l != a.only_layer)) return;
    const int W = a.lvl_w[a.lvl_out[l]][sg.line], C4 = a.cout[l] / 4;
    float *dst = a.act[l] + a.out_off[l][sg.line] + ((size_t)h * W + sg.c0) * a.cout[l];
    const float *src = a.cvec[l] + (size_t)h * a.cout[l];
    const int total = (sg.c1 - sg.c0) * C4;
    for (int i = threadIdx.x; i < total; i += 256)
        reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src)[i % C4];
}

struct ConvArgs {
    const float *x;          // input NHWC [n][H][W][cin]
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
    // Ragged batches (lines of one launch padded to DIFFERENT widths, each to its reference chunk's W_pad):
    // when `tiles` is set, pixel tile p works on line tiles[p].line whose input width is line_w[line] and
    // whose input / output images start at element offsets in_off[line] / out_off[line]; n, W, Wo and
    // tiles_w/tiles_h above are then unused and `n_ptiles` is the number of pixel tiles.
    const PixelTile *tiles;
    const int32_t *line_w;
    const int64_t *in_off;
    const int64_t *out_off;
    int32_t n_ptiles;
    // STAGE_UPCAT: x = [n][H/2][W/2][cin_up] (upsampled on the fly), x2 = [n][H][W][cin - cin_up] (skip connection)
    const float *x2;
    int32_t cin_up;
    // channel tiles per XCD (block -> tile mapping of conv3x3_bf16x3_kernel): 0 / 1 = every XCD works on ONE channel tile
    // (its L2 holds 1 / tiles_n of the weights; a pixel tile's halo is fetched by tiles_n XCDs), G > 1 = consecutive
    // workgroups of an XCD take G channel tiles of the SAME pixel tile (the halo is fetched once per G, the XCD's L2 holds G / tiles_n of the weights)
    int32_t xcd_g;
    // conv3x3_bf16x3_kernel FUSE1 (conv2 computing conv1 for its own halo tile, conv_bf16x3.hpp): conv1's inputs, as Conv1Args
    const uint8_t *f1_crops;
    const LineDesc *f1_lines;
    const float *f1_lut;
    const void *f1_w;        // conv1 weights as f16x2 fragments (Conv1Args::w1x2)
    const float *f1_bias;
    int32_t f1_src_h;
    // f16x2 range guard (conv_bf16x3.hpp): [8] words, one per XCD-ish bucket (blockIdx & 7) - the bit pattern of the largest
    // |output| of this launch; NULL = off.  The host decides at collect time whether the launch left f16's range.
    unsigned *range_max;
    unsigned *f1_range;      // the same for conv1's activation when conv2 computes it in its prologue (FUSE1)
    // persistent kernels (conv_rows.hpp): the number of (channel tile, pixel tile) blocks of the layer = conv_grid_blocks(); the launch's
    // grid is smaller and every workgroup walks blocks blockIdx.x, + gridDim.x, ...  0: one block per workgroup (gridDim.x blocks)
    int32_t nblocks;
};

// number of workgroups for a conv launch (must match the block -> tile mapping in the kernel)
inline size_t conv_grid_blocks(const ConvArgs &a) {
    const size_t P = a.tiles ? (size_t)a.n_ptiles : (size_t)a.tiles_w * a.tiles_h * a.n;
    const int tn = a.tiles_n;
    if (tn >= 2 && tn <= 8 && (tn & (tn - 1)) == 0) {
        const int G = a.xcd_g > 1 && tn % a.xcd_g == 0 ? a.xcd_g : 1;
        const size_t groups = 8 * G / tn;              // XCDs that work on the same channel tiles
        return (P + groups - 1) / groups * G * 8;
    }
    return P * tn;
}

// 4x4 transpose inside each quad of lanes: on entry lane (quad base + b) holds v[r] = M[b][r], on exit
// v[c] = M[c][b].  Used by the epilogues: the MFMA D layout gives a lane 4 pixels x 1 channel; after the
// transpose it holds 1 pixel x 4 consecutive channels, i.e. one 16-byte store instead of four 4-byte ones.
__device__ __forceinline__ void quad_transpose(float (&v)[4], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    {   // exchange across bit 0: register pairs (0,1) and (2,3)
        const float s0 = b0 ? v[0] : v[1], s1 = b0 ? v[2] : v[3];
        const float r0 = __shfl_xor(s0, 1, 64), r1 = __shfl_xor(s1, 1, 64);
        if (b0) { v[0] = r0; v[2] = r1; } else { v[1] = r0; v[3] = r1; }
    }
    {   // exchange across bit 1: register pairs (0,2) and (1,3)
        const float s0 = b1 ? v[0] : v[2], s1 = b1 ? v[1] : v[3];
        const float r0 = __shfl_xor(s0, 2, 64), r1 = __shfl_xor(s1, 2, 64);
        if (b1) { v[0] = r0; v[1] = r1; } else { v[2] = r0; v[3] = r1; }
    }
}

// ---- f16x2 range guard.  The two-plane f16 form of a value (conv_bf16x3.hpp) has fp32's precision only while the tensor it
// belongs to stays inside f16's range: h = f16(x) is inf from |x| >= 65520 on, and l = f16((x - h) 2^11) is a subnormal
// (absolute resolution 2^-35) once the whole tensor lies below ~2^-13.  Every kernel that produces an activation an f16x2
// kernel will consume records the largest |value| it wrote (bit patterns of non-negative floats order like integers; inf and
// NaN sort above every finite value); pocr_slot_collect reads the words and re-runs a launch that left the range on the
// bf16x3 kernels (fp32's range) - pocr_hip.hip: range_verdict / fallback.
__device__ __forceinline__ void range_note(unsigned &m, float v) { m = max(m, __builtin_bit_cast(unsigned, v) & 0x7fffffffu); }
__device__ __forceinline__ void range_publish(unsigned *words, unsigned m, int lane) {
    if (!words) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    unsigned *w = words + (blockIdx.x & 7);
    // one L2 read per wave; the atomic only while this wave still raises the maximum (a handful per launch)
    if (lane == 0 && m > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, m);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_LEAKY) return v > 0.f ? v : v * 0.01f;
    return v;
}

template <int KH, int KW, int PADH, int PADW, int TH, int MW, int NS, int NWAVE, int KC,
          int POOLH, int POOLW, int ACT, bool BN, int STAGER, int PIPE = PIPE_PLAIN, int ABL3 = 0, int MINW = 2>
__global__ __launch_bounds__(NWAVE * 64, MINW) void conv_igemm_kernel(ConvArgs a) {   // 2 waves per SIMD: two workgroups per CU cover each other's stalls
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

    // Block -> tile mapping, XCD-aware.  Block b is dispatched to XCD b % 8 and every XCD has its own
    // 4 MiB L2.  (a) When the layer has 2, 4 or 8 output-channel tiles, each XCD works on ONE of them
    // (nt = xcd % tiles_n): the weight stream an XCD re-reads for every pixel tile is 1/tiles_n of the
    // layer's weights and stays L2-resident (conv9: 9.4 MB of weights, 2.4 MB per XCD at NT = 128).
    // With 16 KB weight tiles needed every ~2.6 us by every workgroup, an L2 miss on that stream
    // (2-4 us from HBM/MALL) stalls the whole workgroup at the next barrier.  (b) Otherwise each XCD
    // gets a contiguous range of logical tiles so halo re-reads hit its L2.
    int nt, ptile;
    {
        const int tn = a.tiles_n;
        const int P = a.tiles ? a.n_ptiles : a.tiles_w * a.tiles_h * a.n;
        if (tn >= 2 && tn <= 8 && (tn & (tn - 1)) == 0) {
            const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, groups = 8 / tn;
            nt = xcd % tn;
            ptile = k * groups + xcd / tn;
            if (ptile >= P) return;                 // grid is padded to a multiple of 8 (whole block leaves)
        } else {
            int b = blockIdx.x;
            const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = b & 7, k = b >> 3;
            b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
            nt = b % tn;
            ptile = b / tn;
        }
    }
    int wt, ht, img, Win;            // Win: input (= conv output) width of this line at this layer
    size_t img_base, out_base;       // element offsets of the line's input and output images
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

    f32x4 acc[MS][NS];
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[m][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.cin / KC;
    const int nsteps = nchunks * NTAPS;

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
                if (hi >= 0 && hi < a.H && wi >= 0 && wi < Win)
                    v = *reinterpret_cast<const f32x4 *>(a.x + img_base + ((size_t)hi * Win + wi) * a.cin + c0 + cq * 4);
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

    static_assert(STAGER != STAGE_UPCAT || PIPE == PIPE_INTERLEAVED || PIPE == PIPE_DEEP, "the up-sample/concat stager lives in the interleaved pipelines");
    if constexpr (PIPE == PIPE_PLAIN) {
    // ------------------------------------------------------------------ plain two-stage loop
    // (conv1's u8 stager and the rare aggregation heights use it; KC may be 32 here)
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
        if (newA) store_A(abuf ^ 1);
        __syncthreads();
        if (newA) abuf ^= 1;
        tap = ntap; chunk = nchunk;
    }
    } else if constexpr (PIPE == PIPE_BREG) {
    // ------------------------------------------------------------------ weights in registers, one barrier per chunk
    static_assert(KG == 1, "written for KC == 16");
    static_assert(NTAPS % 3 == 0, "three weight register sets rotate statically over the unrolled taps");
    constexpr int NMFMA = 4 * MS * NS;
    constexpr int NSLOT = (NMFMA % 16 == 0 && NS + A_LD <= 8) ? 16 : (NMFMA % 12 == 0 && NS + A_LD <= 6) ? 12
                        : (NMFMA % 10 == 0 && NS + A_LD <= 5) ? 10 : 8;
    constexpr int STRIDE = NMFMA / NSLOT;
    static_assert(NMFMA % NSLOT == 0 && NS + A_LD <= NSLOT / 2 && A_LD <= NSLOT / 2, "slot plan does not fit");
    unsigned a_off[A_LD];
    bool a_ok[A_LD];
    int a_lds[A_LD];
#pragma unroll
    for (int r = 0; r < A_LD; ++r) {
        const int e = tid + r * NTHR;
        const int cq = e % CQ, p = e / CQ;
        const int hr = p / HW, wc = p % HW;
        const int hi = h0 - PADH + hr, wi = w0 - PADW + wc;
        a_ok[r] = e < CQ * NP && hi >= 0 && hi < a.H && wi >= 0 && wi < Win;
        a_off[r] = a_ok[r] ? (unsigned)((hi * Win + wi) * a.cin + cq * 4) : 0u;
        a_lds[r] = e < CQ * NP ? cq * NPPAD + p : -1;
    }
    const float *ximg = a.x + img_base;
    // this lane's column of fragments: tile(tap, chunk)[n * 64] is the B operand of fragment n
    const f32x4 *wf4 = reinterpret_cast<const f32x4 *>(a.wfrag) + ((size_t)nt * (NT / 16) + wave * NS) * 64 + lane;
    const size_t tap_stride = (size_t)(a.cin / 16) * a.cout16 * 64;
    const size_t chunk_stride = (size_t)a.cout16 * 64;
    auto ldA = [&](int r, int chunk_) {
        ra[r] = a_ok[r] ? *reinterpret_cast<const f32x4 *>(ximg + chunk_ * KC + a_off[r]) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto stA = [&](int r, int buf) { if (a_lds[r] >= 0) ldsA[buf * A_F4 + a_lds[r]] = ra[r]; };
    f32x4 bq[3][NS];
#pragma unroll
    for (int r = 0; r < A_LD; ++r) ldA(r, 0);
#pragma unroll
    for (int n = 0; n < NS; ++n) bq[0][n] = wf4[n * 64];
    if (nsteps > 1) {
#pragma unroll
        for (int n = 0; n < NS; ++n) bq[1][n] = (wf4 + tap_stride)[n * 64];
    }
#pragma unroll
    for (int r = 0; r < A_LD; ++r) stA(r, 0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int abuf = chunk & 1;
        const bool next_chunk = chunk + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            const bool last = tap == NTAPS - 1;
            constexpr int dummy = 0; (void)dummy;
            const int cur = tap % 3, fill = (tap + 2) % 3;
            const int tap2 = (tap + 2) % NTAPS, adv2 = (tap + 2) / NTAPS;
            const bool more2 = chunk + adv2 < nchunks;
            const f32x4 *tile2 = wf4 + (size_t)tap2 * tap_stride + (size_t)(chunk + adv2) * chunk_stride;
            const bool ldA_now = tap == 0 && next_chunk;
            const bool stA_now = last && next_chunk;
            const int dy = tap / KW, dx = tap % KW;
            const f32x4 *Ab = ldsA + abuf * A_F4 + dy * HW + dx + li + kq * NPPAD;
            f32x4 af[MS];
#pragma unroll
            for (int m = 0; m < MS; ++m) af[m] = Ab[(m / MW) * HW + (m % MW) * 16];
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                const int j = q / (MS * NS), m = (q / NS) % MS, n = q % NS;
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bq[cur][n][j], acc[m][n], 0, 0, 0);
                if ((q + 1) % STRIDE == 0) {
                    const int slot = (q + 1) / STRIDE - 1;
                    __builtin_amdgcn_sched_barrier(0);
                    if (slot < NS) { if (more2) bq[fill][slot] = tile2[slot * 64]; }
                    else if (slot < NS + A_LD) { if (ldA_now) ldA(slot - NS, chunk + 1); }
                    const int sslot = slot - (NSLOT - A_LD);
                    if (sslot >= 0) { if (stA_now) stA(sslot, abuf ^ 1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (last) __syncthreads();      // the next halo tile is published; nobody reads this one any more
        }
    }
    } else {
    static_assert(PIPE == PIPE_INTERLEAVED || PIPE == PIPE_DEEP || PIPE == PIPE_GLDS || PIPE == PIPE_DEEP3, "unknown pipeline id");
    constexpr bool GLDS = PIPE == PIPE_GLDS;
    constexpr bool DEEP = PIPE == PIPE_DEEP;     // weight tile s+2 is requested right after tile s+1 has been written to LDS
    // DEEP3: two staging register sets; tile s+3 is requested right after tile s+1 has been written, so a weight
    // load has two full steps to arrive.  The sets alternate with the step parity; the taps are unrolled, so the set
    // of a tap is static and an odd tap count is squared up by swapping the sets once per chunk.
    constexpr bool DEEP3 = PIPE == PIPE_DEEP3;
    static_assert(!DEEP3 || (NTAPS % 2 == 1 && NTAPS >= 3), "DEEP3 is written for odd tap counts (3x3 convs)");
    // ------------------------------------------------------------------ interleaved two-stage pipeline
    // Same LDS double buffering as the plain loop, but (a) the tap loop is fully unrolled inside a runtime
    // chunk loop, so tap offsets are immediates and the per-step scalar bookkeeping disappears,
    // and (b) the global loads of the next step and the LDS writes that publish them are issued
    // one at a time in the shadow of the MFMA stream (a 32-cycle MFMA hides ~5 issue slots)
    // instead of as bursts before the first / after the last MFMA of a step.
    static_assert(KG == 1, "the interleaved pipeline is written for KC == 16");
    constexpr int NMFMA = 4 * MS * NS;
    // issue slots per step: the largest of 16, 12, 10, 8 that divides the MFMA count and leaves
    // separate halves for the loads (first half) and the LDS writes (second half)
    constexpr int NSLOT = (NMFMA % 16 == 0 && B_LD + A_LD <= 8) ? 16 : (NMFMA % 12 == 0 && B_LD + A_LD <= 6) ? 12
                        : (NMFMA % 10 == 0 && B_LD + A_LD <= 5) ? 10 : 8;
    constexpr int STRIDE = NMFMA / NSLOT;                   // MFMAs between slots
    static_assert(NMFMA % NSLOT == 0 && B_LD + A_LD <= NSLOT / 2, "slot plan does not fit");
    // per-thread constant parts of the staging addresses
    constexpr bool UPCAT = STAGER == STAGE_UPCAT;
    unsigned a_off[A_LD];
    unsigned a_off2[UPCAT ? A_LD : 1];          // UPCAT: offsets into the skip tensor (a_off: into the half-resolution one)
    bool a_ok[A_LD];
    int a_lds[A_LD];
    const int cin_skip = a.cin - a.cin_up, nch_up = a.cin_up / KC;
#pragma unroll
    for (int r = 0; r < A_LD; ++r) {
        const int e = tid + r * NTHR;
        const int cq = e % CQ, p = e / CQ;
        const int hr = p / HW, wc = p % HW;
        const int hi = h0 - PADH + hr, wi = w0 - PADW + wc;
        a_ok[r] = e < CQ * NP && hi >= 0 && hi < a.H && wi >= 0 && wi < Win;
        if constexpr (UPCAT) {
            a_off[r] = a_ok[r] ? (unsigned)(((hi >> 1) * (Win >> 1) + (wi >> 1)) * a.cin_up + cq * 4) : 0u;
            a_off2[r] = a_ok[r] ? (unsigned)((hi * Win + wi) * cin_skip + cq * 4) : 0u;
        } else {
            a_off[r] = a_ok[r] ? (unsigned)((hi * Win + wi) * a.cin + cq * 4) : 0u;
        }
        a_lds[r] = e < CQ * NP ? cq * NPPAD + p : -1;
    }
    const float *ximg = UPCAT ? a.x + (size_t)img * (a.H >> 1) * (Win >> 1) * a.cin_up : a.x + img_base;
    const float *ximg2 = UPCAT ? a.x2 + (size_t)img * a.H * Win * cin_skip : nullptr;
    const f32x4 *wf4 = reinterpret_cast<const f32x4 *>(a.wfrag) + (size_t)nt * (NT / 16) * 64;
    const size_t tap_stride = (size_t)(a.cin / 16) * a.cout16 * 64;     // f32x4 per tap
    const size_t chunk_stride = (size_t)a.cout16 * 64;                   // f32x4 per 16-channel group
    auto ldA = [&](int r, int chunk_) {
        if constexpr (UPCAT) {
            const float *src = chunk_ < nch_up ? ximg + chunk_ * KC + a_off[r] : ximg2 + (chunk_ - nch_up) * KC + a_off2[r];
            ra[r] = a_ok[r] ? *reinterpret_cast<const f32x4 *>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
        } else {
            ra[r] = a_ok[r] ? *reinterpret_cast<const f32x4 *>(ximg + chunk_ * KC + a_off[r]) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stA = [&](int r, int buf) { if (a_lds[r] >= 0) ldsA[buf * A_F4 + a_lds[r]] = ra[r]; };
    auto ldB = [&](int r, const f32x4 *tile) {
        const int f = tid + r * NTHR;
        if (B_F4 % NTHR == 0 || f < B_F4) rb[r] = tile[f];
    };
    auto stB = [&](int r, int buf) {
        const int f = tid + r * NTHR;
        if (B_F4 % NTHR == 0 || f < B_F4) ldsB[buf * B_F4 + f] = rb[r];
    };
    f32x4 rb2[DEEP3 ? B_LD : 1];                   // second staging set (DEEP3)
    auto ldB2 = [&](int r, const f32x4 *tile) {
        const int f = tid + r * NTHR;
        if (B_F4 % NTHR == 0 || f < B_F4) rb2[r] = tile[f];
    };
    auto stB2 = [&](int r, int buf) {
        const int f = tid + r * NTHR;
        if (B_F4 % NTHR == 0 || f < B_F4) ldsB[buf * B_F4 + f] = rb2[r];
    };
    // direct-to-LDS copy of piece r of a weight tile: every lane supplies its own global address, the LDS
    // destination is (wave-uniform base) + lane * 16 B, which is exactly the tile's linear layout
    auto dmaB = [&](int r, const f32x4 *tile, int buf) {
        static_assert(!GLDS || B_F4 % NTHR == 0, "GLDS needs whole-wave pieces");
        const int f0 = __builtin_amdgcn_readfirstlane(wave * 64 + r * NTHR);
        typedef const __attribute__((address_space(1))) void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        __builtin_amdgcn_global_load_lds((gptr_t)(tile + f0 + lane), (lptr_t)(ldsB + buf * B_F4 + f0), 16, 0, 0);
    };
#pragma unroll
    for (int r = 0; r < A_LD; ++r) ldA(r, 0);
#pragma unroll
    for (int r = 0; r < B_LD; ++r) ldB(r, wf4);
#pragma unroll
    for (int r = 0; r < A_LD; ++r) stA(r, 0);
#pragma unroll
    for (int r = 0; r < B_LD; ++r) stB(r, 0);
    if constexpr (DEEP || DEEP3) {
        if (nsteps > 1) {
            const f32x4 *t1 = NTAPS > 1 ? wf4 + tap_stride : wf4 + chunk_stride;
#pragma unroll
            for (int r = 0; r < B_LD; ++r) ldB(r, t1);
        }
    }
    if constexpr (DEEP3) {
        if (nsteps > 2) {
#pragma unroll
            for (int r = 0; r < B_LD; ++r) ldB2(r, wf4 + 2 * tap_stride);
        }
    }
    __syncthreads();

    f32x4 abl_a3[MS], abl_b3[NS];
    if constexpr (ABL3 & 4) {
#pragma unroll
        for (int m = 0; m < MS; ++m) abl_a3[m] = ldsA[li + kq * NPPAD + (m / MW) * HW + (m % MW) * 16];
#pragma unroll
        for (int n = 0; n < NS; ++n) abl_b3[n] = ldsB[(wave * NS + n) * 64 + lane];
    }
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int abuf = chunk & 1;
        const int par = (NTAPS & 1) ? (chunk & 1) : 0;      // parity of the global step index at tap 0
        const bool next_chunk = chunk + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            const bool last = tap == NTAPS - 1;
            const bool more = !last || next_chunk;          // is there a step after this one?
            const int bcur = (tap & 1) ^ par, bnext = bcur ^ 1;
            const f32x4 *tile = wf4 + (size_t)(last ? 0 : tap + 1) * tap_stride + (size_t)(last ? chunk + 1 : chunk) * chunk_stride;
            // DEEP: the tile two steps ahead (requested in this step's late slots, right after the writes of tile s+1)
            const int tap2 = (tap + 2) % NTAPS, adv2 = (tap + 2) / NTAPS;
            const bool more2 = chunk + adv2 < nchunks;
            const f32x4 *tile2 = wf4 + (size_t)tap2 * tap_stride + (size_t)(chunk + adv2) * chunk_stride;
            const int tap3 = (tap + 3) % NTAPS, adv3 = (tap + 3) / NTAPS;
            const bool more3 = chunk + adv3 < nchunks;
            const f32x4 *tile3 = wf4 + (size_t)tap3 * tap_stride + (size_t)(chunk + adv3) * chunk_stride;
            const bool ldA_now = (NTAPS == 1 ? next_chunk : (tap == 0 && next_chunk));   // request the next halo tile early
            const bool stA_now = last && next_chunk;

            const int dy = tap / KW, dx = tap % KW;
            const f32x4 *Ab = ldsA + abuf * A_F4 + dy * HW + dx + li + kq * NPPAD;
            const f32x4 *Bb = ldsB + bcur * B_F4 + (wave * NS) * 64 + lane;
            f32x4 af[MS], bf[NS];
            if constexpr (ABL3 & 4) {     // ablation (tools/conv_bench only): operands from fixed LDS addresses, hoistable
#pragma unroll
                for (int m = 0; m < MS; ++m) af[m] = abl_a3[m];
#pragma unroll
                for (int n = 0; n < NS; ++n) bf[n] = abl_b3[n];
            } else {
#pragma unroll
            for (int m = 0; m < MS; ++m) af[m] = Ab[(m / MW) * HW + (m % MW) * 16];
#pragma unroll
            for (int n = 0; n < NS; ++n) bf[n] = Bb[n * 64];
            }
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                const int j = q / (MS * NS), m = (q / NS) % MS, n = q % NS;
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bf[n][j], acc[m][n], 0, 0, 0);
                if ((q + 1) % STRIDE == 0) {
                    const int slot = (q + 1) / STRIDE - 1;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!(ABL3 & 1)) {
                        if (slot < B_LD) {
                            if constexpr (GLDS) { if (more) dmaB(slot, tile, bnext); }
                            else if constexpr (!DEEP && !DEEP3) { if (more) ldB(slot, tile); }
                        }
                        else if (slot < B_LD + A_LD) { if (ldA_now) ldA(slot - B_LD, chunk + 1); }
                    }
                    const int sslot = slot - (NSLOT - B_LD - A_LD);
                    if constexpr (!(ABL3 & 2)) {
                        if (sslot >= 0 && sslot < B_LD) {
                            if constexpr (DEEP3) {
                                if ((tap & 1) == 0) { if (more) stB(sslot, bnext); if (more3) ldB(sslot, tile3); }
                                else { if (more) stB2(sslot, bnext); if (more3) ldB2(sslot, tile3); }
                            } else {
                            if constexpr (!GLDS) { if (more) stB(sslot, bnext); }
                            if constexpr (DEEP) { if (more2) ldB(sslot, tile2); }
                            }
                        }
                        else if (sslot >= B_LD) { if (stA_now) stA(sslot - B_LD, abuf ^ 1); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (ABL3 & 16) {      // ablation: loads kept alive, consumed (waited for) only at the very end of the step
#pragma unroll
                for (int r = 0; r < B_LD; ++r) asm volatile("" ::"v"(rb[r]));
#pragma unroll
                for (int r = 0; r < A_LD; ++r) asm volatile("" ::"v"(ra[r]));
            }
            if constexpr (!(ABL3 & 8)) __syncthreads();
        }
        if constexpr (DEEP3) {          // odd tap count: the set that comes next is rb2 -> make it rb
#pragma unroll
            for (int r = 0; r < B_LD; ++r) { const f32x4 t = rb[r]; rb[r] = rb2[r]; rb2[r] = t; }
        }
    }
    }

    // ---- epilogue.  D layout: col = lane & 15 (cout), row = (lane >> 4) * 4 + reg (pixel).
    const int Wo = Win;              // 3x3 pad 1, (k x 1) valid and 1x1 convs keep the width
    const int Wout = Wo / POOLW;
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
                float *yrow = a.y + out_base + ((size_t)ho * Wout) * a.out_stride + co;
                if constexpr (POOLW == 2) {
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int wc = wbase + 2 * rr;
                        if (co_ok && h0 + th < a.Ho && wc + 1 < Wo) yrow[(size_t)(wc / 2) * a.out_stride] = fmaxf(v[2 * rr], v[2 * rr + 1]);
                    }
                } else {
                    // every lane of the wave takes part in the transpose; rows/channels out of range are masked at the store
                    quad_transpose(v, lane);         // pixel wbase + (li & 3), channels co - (li & 3) + 0..3
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
}

}  // namespace pocr
