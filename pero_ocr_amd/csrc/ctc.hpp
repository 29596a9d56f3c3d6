// ctc.hpp — greedy CTC decode on the GPU.  Replaces greedy_decode_ctc (3-D branch),
// pero_ocr/ocr_engine/pytorch_ocr_engine.py:18-28:
//   best = argmax over classes per frame (torch.argmax: first index wins ties, NaN is maximal),
//   drop a frame equal to its predecessor (virtual frame before t=0 is blank), drop blanks (C-1),
//   keep the remaining class ids in order.
// Kernel 1: one wavefront per frame, wave-shuffle (value, index) arg-max reduction.
// Kernel 2: one wavefront per line, ballot + popcount stream compaction of the kept frames.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pocr {

// "a beats b" under torch.argmax semantics
__device__ __forceinline__ bool argmax_better(float av, int ai, float bv, int bi) {
    const bool an = av != av, bn = bv != bv;
    if (an || bn) return (an && !bn) || (an && bn && ai < bi);
    return av > bv || (av == bv && ai < bi);
}

// logits [frames][C] (frames = n*T), out [frames].  nonfinite (or NULL): set to 1 when a frame's winner is NaN / +-inf - the
// reference would decode such logits silently (torch.argmax: NaN is maximal); the engine prints a warning, because with the
// f16x2 arithmetic an activation beyond f16's range (65504) is the one thing that can produce them (conv_bf16x3.hpp).
__global__ __launch_bounds__(256) void frame_argmax_kernel(const float *logits, int32_t *out, int frames, int C, int32_t *nonfinite = nullptr) {
    const int lane = threadIdx.x & 63;
    const int frame = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= frames) return;
    const float *row = logits + (size_t)frame * C;
    float bv = 0.f;
    int bi = 0x7fffffff;          // "no candidate yet": loses to everything
    bool have = false;
    for (int c = lane; c < C; c += 64) {
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
    if (lane == 0) {
        out[frame] = bi;
        if (nonfinite && !(fabsf(bv) <= 3.4028234e38f)) *nonfinite = 1;
    }
}

// best [rows] -> labels [n][stride] (compacted, -1 padded), len [n].  Line i owns rows row_off[i] ..
// row_off[i] + line_T[i] (NULL arrays: every line has T frames, line i starts at row i * T).
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int32_t *best, int32_t *labels, int32_t *len,
                                                          int T_uniform, int blank, const int32_t *line_T,
                                                          const int32_t *row_off, int stride) {
    const int line = blockIdx.x, lane = threadIdx.x;
    const int T = line_T ? line_T[line] : T_uniform;
    const int32_t *b = best + (row_off ? (size_t)row_off[line] : (size_t)line * T_uniform);
    int32_t *out = labels + (size_t)line * stride;
    int count = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        bool keep = false;
        int cur = blank;
        if (t < T) {
            cur = b[t];
            const int prev = t == 0 ? blank : b[t - 1];
            keep = (cur != prev) && (cur != blank);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) out[count + __popcll(m & ((1ull << lane) - 1ull))] = cur;
        count += __popcll(m);
    }
    for (int t = count + lane; t < stride; t += 64) out[t] = -1;      // deterministic tail
    if (lane == 0) len[line] = count;
}

}  // namespace pocr
