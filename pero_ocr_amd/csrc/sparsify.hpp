// sparsify.hpp — logit sparsification on the GPU.  Replaces, per line,
//   line_probs = softmax(line_logits, axis=1); line_logits[line_probs < 0.0001] = 0;
//   line_logits = scipy.sparse.csc_matrix(line_logits)
// (pero_ocr/ocr_engine/line_ocr_engine.py:168-171, softmax = pero_ocr/ocr_engine/softmax.py:4-46)
// so that only the CSC triplets cross PCIe instead of the dense [T, C] float matrix, and the host
// does no per-element work.  An entry is kept iff NOT (p < threshold) and its logit is not exactly 0
// (csc_matrix drops explicit zeros).  Rows may be restricted to [row_begin, row_end) per line
// (tight_crop_logits, line_ocr_engine.py:146-150); stored row indices are relative to row_begin.
//
//   sparse_count_kernel   one workgroup per line: per-row max / sum(exp) (wave per row), then every
//                         thread counts the kept entries of its own columns (no atomics)
//   sparse_scan_kernel    exclusive scan of the per-line totals -> line_off[n+1]
//   sparse_fill_kernel    per line: column scan -> indptr, then each thread streams its columns
//                         top to bottom so rows come out sorted, as CSC requires
//   line_confidence_kernel  transcription confidence of a line from the SAME kept set, i.e. what the reference's
//                         caller computes from the sparse matrix: PageParser.compute_line_confidence + get_prob
//                         (pero_ocr/document_ocr/page_parser.py:485-496, 437-450) on TextLine.get_dense_logits
//                         (pero_ocr/core/layout.py:65-68: dropped entries count as -80)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pocr {

constexpr int SP_MAXT = 1024;        // frames per line supported (T <= 960 for 3840-px chunks)
constexpr int SP_COLS = 8;           // columns per thread -> C <= 2048

__device__ __forceinline__ bool sp_keep(float x, float rmax, float rsum, float thr) {
    const float p = expf(x - rmax) / rsum;
    return !(p < thr) && x != 0.0f;
}

// logits [n][T][C]; rowstat [n][T][2]; colcount [n][C]; line_nnz [n]
__global__ __launch_bounds__(256) void sparse_count_kernel(const float *logits, const int32_t *row_begin,
                                                           const int32_t *row_end, float *rowstat, int32_t *colcount,
                                                           int32_t *line_nnz, int T_uniform, int C, float thr,
                                                           const int32_t *line_T, const int32_t *row_off) {
    __shared__ float smax[SP_MAXT], ssum[SP_MAXT];
    __shared__ int wsum[4];
    const int line = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = line_T ? line_T[line] : T_uniform;
    const size_t row0 = row_off ? (size_t)row_off[line] : (size_t)line * T_uniform;     // first row of this line
    const int r0 = row_begin ? row_begin[line] : 0, r1 = row_end ? row_end[line] : T;
    const float *x = logits + row0 * C;
    for (int t = r0 + wave; t < r1; t += 4) {
        const float *row = x + (size_t)t * C;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += expf(row[c] - m);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) {
            smax[t] = m; ssum[t] = s;
            rowstat[(row0 + t) * 2] = m;
            rowstat[(row0 + t) * 2 + 1] = s;
        }
    }
    __syncthreads();
    int cnt[SP_COLS];
#pragma unroll
    for (int k = 0; k < SP_COLS; ++k) cnt[k] = 0;
    for (int t = r0; t < r1; ++t) {
        const float *row = x + (size_t)t * C;
        const float m = smax[t], s = ssum[t];
#pragma unroll
        for (int k = 0; k < SP_COLS; ++k) {
            const int c = tid + 256 * k;
            if (c < C && sp_keep(row[c], m, s, thr)) ++cnt[k];
        }
    }
    int tot = 0;
#pragma unroll
    for (int k = 0; k < SP_COLS; ++k) {
        const int c = tid + 256 * k;
        if (c < C) colcount[(size_t)line * C + c] = cnt[k];
        tot += cnt[k];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
    if (lane == 0) wsum[wave] = tot;
    __syncthreads();
    if (tid == 0) line_nnz[line] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void sparse_scan_kernel(const int32_t *line_nnz, int64_t *line_off, int n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t acc = 0;
        for (int i = 0; i < n; ++i) { line_off[i] = acc; acc += line_nnz[i]; }
        line_off[n] = acc;
    }
}

// indptr [n][C+1] (per line, starting at 0); data / indices: global CSC pools addressed by line_off
__global__ __launch_bounds__(256) void sparse_fill_kernel(const float *logits, const int32_t *row_begin,
                                                          const int32_t *row_end, const float *rowstat,
                                                          const int32_t *colcount, const int64_t *line_off,
                                                          int32_t *indptr, float *data, int32_t *indices, int T_uniform, int C,
                                                          float thr, int64_t capacity, const int32_t *line_T,
                                                          const int32_t *row_off) {
    __shared__ int part[256];
    const int line = blockIdx.x, tid = threadIdx.x;
    const int T = line_T ? line_T[line] : T_uniform;
    const size_t row0 = row_off ? (size_t)row_off[line] : (size_t)line * T_uniform;
    const int r0 = row_begin ? row_begin[line] : 0, r1 = row_end ? row_end[line] : T;
    const float *x = logits + row0 * C;
    const int32_t *cc = colcount + (size_t)line * C;
    // exclusive scan over columns: thread i owns the contiguous column block [i*per, (i+1)*per)
    const int per = (C + 255) / 256;
    int local = 0;
    for (int c = tid * per; c < min(C, (tid + 1) * per); ++c) local += cc[c];
    part[tid] = local;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - local;
    int32_t *ip = indptr + (size_t)line * (C + 1);
    for (int c = tid * per; c < min(C, (tid + 1) * per); ++c) { ip[c] = run; run += cc[c]; }
    if (tid == 255) ip[C] = part[255];
    __syncthreads();                                    // indptr of this line is complete (global, same block)
    // fill: the thread that owns column c walks the rows in order
    const int64_t base = line_off[line];
    for (int k = 0; k < SP_COLS; ++k) {
        const int c = tid + 256 * k;
        if (c >= C) break;
        int64_t pos = base + ip[c];
        for (int t = r0; t < r1; ++t) {
            const float v = x[(size_t)t * C + c];
            const float m = rowstat[(row0 + t) * 2], s = rowstat[(row0 + t) * 2 + 1];
            if (sp_keep(v, m, s, thr)) {
                if (pos < capacity) { data[pos] = v; indices[pos] = t - r0; }
                ++pos;
            }
        }
    }
}

// conf [n].  Per frame: D = kept ? logit : fill; winner = first arg-max of D, its probability = 1 / sum exp(D - Dmax).
// Per line: frames are grouped into runs of equal winner, a run is worth its highest probability, the line its worst run.
__global__ __launch_bounds__(256) void line_confidence_kernel(const float *logits, const int32_t *row_begin,
                                                              const int32_t *row_end, const float *rowstat, float *conf,
                                                              int T_uniform, int C, float thr, float fill,
                                                              const int32_t *line_T, const int32_t *row_off) {
    __shared__ int bid[SP_MAXT];
    __shared__ float bp[SP_MAXT];
    const int line = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = line_T ? line_T[line] : T_uniform;
    const size_t row0 = row_off ? (size_t)row_off[line] : (size_t)line * T_uniform;
    const int r0 = row_begin ? row_begin[line] : 0, r1 = row_end ? row_end[line] : T;
    const float *x = logits + row0 * C;
    for (int t = r0 + wave; t < r1; t += 4) {
        const float *row = x + (size_t)t * C;
        const float m = rowstat[(row0 + t) * 2], s = rowstat[(row0 + t) * 2 + 1];
        float kv = -INFINITY, ks = 0.f;
        int ki = 0x7fffffff, di = 0x7fffffff, nd = 0;
        for (int c = lane; c < C; c += 64) {
            const float v = row[c];
            if (sp_keep(v, m, s, thr)) {
                ks += expf(v - m);
                if (v > kv || (v == kv && c < ki)) { kv = v; ki = c; }
            } else {
                ++nd;
                di = min(di, c);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            ks += __shfl_xor(ks, off, 64);
            nd += __shfl_xor(nd, off, 64);
            di = min(di, __shfl_xor(di, off, 64));
            const float ov = __shfl_xor(kv, off, 64);
            const int oi = __shfl_xor(ki, off, 64);
            if (ov > kv || (ov == kv && oi < ki)) { kv = ov; ki = oi; }
        }
        if (lane == 0) {
            float dmax = kv;
            int did = ki;
            if (nd > 0 && (ki == 0x7fffffff || fill > kv || (fill == kv && di < ki))) { dmax = fill; did = di; }
            const float sum = (ki == 0x7fffffff ? 0.f : ks * expf(m - dmax)) + (float)nd * expf(fill - dmax);
            bid[t] = did;
            bp[t] = 1.0f / sum;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float worst = 1.f, run_best = 1.f;
        int run_id = -1;
        for (int t = r0; t < r1; ++t) {
            if (bid[t] != run_id) {
                worst = fminf(worst, run_best);
                run_id = bid[t];
                run_best = bp[t];
            } else {
                run_best = fmaxf(run_best, bp[t]);
            }
        }
        conf[line] = fminf(worst, run_best);
    }
}

}  // namespace pocr
