// sparsify.hpp — logit sparsification on the GPU.  Replaces, per line,
//   line_probs = softmax(line_logits, axis=1); line_logits[line_probs < 0.0001] = 0;
//   line_logits = scipy.sparse.csc_matrix(line_logits)
// (pero_ocr/ocr_engine/line_ocr_engine.py:168-171, softmax = pero_ocr/ocr_engine/softmax.py:4-46)
// so that only the CSC triplets cross PCIe instead of the dense [T, C] float matrix, and the host
// does no per-element work.  An entry is kept iff NOT (p < threshold) and its logit is not exactly 0
// (csc_matrix drops explicit zeros).  Rows may be restricted to [row_begin, row_end) per line
// (tight_crop_logits, line_ocr_engine.py:146-150); stored row indices are relative to row_begin.
//
//   sparse_block_kernel   one workgroup per (line, block of 64 frames): per-row max / sum(exp) (wave per row), the
//                         per-row winner and probability of the confidence (below), then every thread counts the kept
//                         entries of its own columns inside the block (no atomics)
//   sparse_line_kernel    one workgroup per line: per column an exclusive prefix over the line's blocks (in place) and the
//                         column totals, their exclusive scan -> indptr, the line's total, and the confidence's serial pass
//   sparse_scan_kernel    exclusive scan of the per-line totals -> line_off[n+1]
//   sparse_fill_kernel    per (line, block): each thread streams its columns top to bottom from the block's offset, so rows
//                         come out sorted, as CSC requires
//   confidence            transcription confidence of a line from the SAME kept set, i.e. what the reference's
//                         caller computes from the sparse matrix: PageParser.compute_line_confidence + get_prob
//                         (pero_ocr/document_ocr/page_parser.py:485-496, 437-450) on TextLine.get_dense_logits
//                         (pero_ocr/core/layout.py:65-68: dropped entries count as -80)
// Rounds 1-3 ran ONE workgroup per line (a thread walked all T frames of its column): fine for 256 lines of 144 frames, a
// 960-step dependent chain per thread for a page of 47 long lines - 2.9 ms per page, 11 ms when the page is one launch
// (profiles/r04_sparse_blocks.txt).  Same per-row arithmetic, same kept set, same order: bit-identical triplets and confidences.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pocr {

constexpr int SP_RB = 64;            // frames per block
constexpr int SP_COLS = 8;           // columns per thread -> C <= 2048
constexpr int SP_CHUNK = 1024;       // frames per pass of the confidence's serial walk

__device__ __forceinline__ bool sp_keep(float x, float rmax, float rsum, float thr) {
    const float p = expf(x - rmax) / rsum;
    return !(p < thr) && x != 0.0f;
}

inline int sp_blocks(int t_max) { return t_max > 0 ? (t_max + SP_RB - 1) / SP_RB : 1; }

// logits: line i = rows row_off[i] .. (uniform: i * T_uniform ..); rowstat [rows][2]; blkcount [n][nblk][C];
// bid / bp [rows] (winner id and probability per frame) or NULL: no confidence wanted
__global__ __launch_bounds__(256) void sparse_block_kernel(const float *logits, const int32_t *row_begin, const int32_t *row_end,
                                                           float *rowstat, int32_t *blkcount, int32_t *bid, float *bp,
                                                           int T_uniform, int C, float thr, float fill,
                                                           const int32_t *line_T, const int32_t *row_off) {
    __shared__ float smax[SP_RB], ssum[SP_RB];
    const int line = blockIdx.x, rb = blockIdx.y, nblk = gridDim.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = line_T ? line_T[line] : T_uniform;
    const size_t row0 = row_off ? (size_t)row_off[line] : (size_t)line * T_uniform;     // first row of this line
    const int r0 = row_begin ? row_begin[line] : 0, r1 = row_end ? row_end[line] : T;
    const int t0 = max(r0, rb * SP_RB), t1 = min(r1, (rb + 1) * SP_RB);
    const float *x = logits + row0 * C;
    for (int t = t0 + wave; t < t1; t += 4) {
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
            smax[t - rb * SP_RB] = m; ssum[t - rb * SP_RB] = s;
            rowstat[(row0 + t) * 2] = m;
            rowstat[(row0 + t) * 2 + 1] = s;
        }
        if (bid) {
            // Per frame: D = kept ? logit : fill; winner = first arg-max of D, its probability = 1 / sum exp(D - Dmax).
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
                bid[row0 + t] = did;
                bp[row0 + t] = 1.0f / sum;
            }
        }
    }
    __syncthreads();
    int cnt[SP_COLS];
#pragma unroll
    for (int k = 0; k < SP_COLS; ++k) cnt[k] = 0;
    for (int t = t0; t < t1; ++t) {
        const float *row = x + (size_t)t * C;
        const float m = smax[t - rb * SP_RB], s = ssum[t - rb * SP_RB];
#pragma unroll
        for (int k = 0; k < SP_COLS; ++k) {
            const int c = tid + 256 * k;
            if (c < C && sp_keep(row[c], m, s, thr)) ++cnt[k];
        }
    }
    int32_t *bc = blkcount + ((size_t)line * nblk + rb) * C;
#pragma unroll
    for (int k = 0; k < SP_COLS; ++k) {
        const int c = tid + 256 * k;
        if (c < C) bc[c] = cnt[k];          // (an empty block writes zeros: the line kernel reads every block)
    }
}

// blkcount [n][nblk][C]: counts in, exclusive prefix over the line's blocks out; indptr [n][C+1] (per line, starting at 0);
// line_nnz [n]; conf [n] or NULL (then bid / bp are not read).  Per line: frames are grouped into runs of equal winner, a run
// is worth its highest probability, the line its worst run.
__global__ __launch_bounds__(256) void sparse_line_kernel(int32_t *blkcount, int nblk, int32_t *indptr, int32_t *line_nnz,
                                                          const int32_t *bid, const float *bp, float *conf,
                                                          const int32_t *row_begin, const int32_t *row_end, int T_uniform, int C,
                                                          const int32_t *line_T, const int32_t *row_off) {
    __shared__ int scol[256 * SP_COLS];
    __shared__ int part[256];
    __shared__ int sb[SP_CHUNK];
    __shared__ float sp[SP_CHUNK];
    const int line = blockIdx.x, tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < SP_COLS; ++k) {
        const int c = tid + 256 * k;
        if (c < C) {
            int run = 0;
            int32_t *bc = blkcount + (size_t)line * nblk * C + c;
            for (int b = 0; b < nblk; ++b) { const int v = bc[(size_t)b * C]; bc[(size_t)b * C] = run; run += v; }
            scol[c] = run;
        }
    }
    __syncthreads();
    // exclusive scan over columns: thread i owns the contiguous column block [i*per, (i+1)*per)
    const int per = (C + 255) / 256;
    int local = 0;
    for (int c = tid * per; c < min(C, (tid + 1) * per); ++c) local += scol[c];
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
    for (int c = tid * per; c < min(C, (tid + 1) * per); ++c) { ip[c] = run; run += scol[c]; }
    if (tid == 255) { ip[C] = part[255]; line_nnz[line] = part[255]; }
    if (!conf) return;
    const int T = line_T ? line_T[line] : T_uniform;
    const size_t row0 = row_off ? (size_t)row_off[line] : (size_t)line * T_uniform;
    const int r0 = row_begin ? row_begin[line] : 0, r1 = row_end ? row_end[line] : T;
    float worst = 1.f, run_best = 1.f;       // (thread 0's)
    int run_id = -1;
    for (int base = r0; base < r1; base += SP_CHUNK) {
        const int cnt = min(SP_CHUNK, r1 - base);
        __syncthreads();
        for (int i = tid; i < cnt; i += 256) { sb[i] = bid[row0 + base + i]; sp[i] = bp[row0 + base + i]; }
        __syncthreads();
        if (tid == 0)
            for (int i = 0; i < cnt; ++i) {
                if (sb[i] != run_id) {
                    worst = fminf(worst, run_best);
                    run_id = sb[i];
                    run_best = sp[i];
                } else {
                    run_best = fmaxf(run_best, sp[i]);
                }
            }
    }
    if (tid == 0) conf[line] = fminf(worst, run_best);
}

__global__ void sparse_scan_kernel(const int32_t *line_nnz, int64_t *line_off, int n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t acc = 0;
        for (int i = 0; i < n; ++i) { line_off[i] = acc; acc += line_nnz[i]; }
        line_off[n] = acc;
    }
}

// data / indices: global CSC pools addressed by line_off + indptr + the block's offset inside its column
__global__ __launch_bounds__(256) void sparse_fill_kernel(const float *logits, const int32_t *row_begin,
                                                          const int32_t *row_end, const float *rowstat,
                                                          const int32_t *blkoff, const int64_t *line_off,
                                                          const int32_t *indptr, float *data, int32_t *indices, int T_uniform, int C,
                                                          float thr, int64_t capacity, const int32_t *line_T,
                                                          const int32_t *row_off) {
    const int line = blockIdx.x, rb = blockIdx.y, nblk = gridDim.y, tid = threadIdx.x;
    const int T = line_T ? line_T[line] : T_uniform;
    const size_t row0 = row_off ? (size_t)row_off[line] : (size_t)line * T_uniform;
    const int r0 = row_begin ? row_begin[line] : 0, r1 = row_end ? row_end[line] : T;
    const int t0 = max(r0, rb * SP_RB), t1 = min(r1, (rb + 1) * SP_RB);
    if (t0 >= t1) return;
    const float *x = logits + row0 * C;
    const int32_t *ip = indptr + (size_t)line * (C + 1);
    const int32_t *bo = blkoff + ((size_t)line * nblk + rb) * C;
    // fill: the thread that owns column c walks the block's rows in order
    const int64_t base = line_off[line];
    for (int k = 0; k < SP_COLS; ++k) {
        const int c = tid + 256 * k;
        if (c >= C) break;
        int64_t pos = base + ip[c] + bo[c];
        for (int t = t0; t < t1; ++t) {
            const float v = x[(size_t)t * C + c];
            const float m = rowstat[(row0 + t) * 2], s = rowstat[(row0 + t) * 2 + 1];
            if (sp_keep(v, m, s, thr)) {
                if (pos < capacity) { data[pos] = v; indices[pos] = t - r0; }
                ++pos;
            }
        }
    }
}

// The four launches.  t_max: the longest line's frames (uniform batches: T_uniform); conf == NULL: no confidences.
inline void sparsify_launch(hipStream_t st, const float *logits, const int32_t *r0, const int32_t *r1, float *rowstat,
                            int32_t *blk, int32_t *line_nnz, int64_t *line_off, int32_t *indptr, float *data, int32_t *indices,
                            int32_t *bid, float *bp, float *conf, int n, int t_max, int T_uniform, int C, float thr, float fill,
                            int64_t capacity, const int32_t *line_T, const int32_t *row_off) {
    const int nblk = sp_blocks(t_max);
    hipLaunchKernelGGL(sparse_block_kernel, dim3(n, nblk), dim3(256), 0, st, logits, r0, r1, rowstat, blk, conf ? bid : nullptr, bp,
                       T_uniform, C, thr, fill, line_T, row_off);
    hipLaunchKernelGGL(sparse_line_kernel, dim3(n), dim3(256), 0, st, blk, nblk, indptr, line_nnz, bid, bp, conf, r0, r1, T_uniform, C,
                       line_T, row_off);
    hipLaunchKernelGGL(sparse_scan_kernel, dim3(1), dim3(64), 0, st, line_nnz, line_off, n);
    hipLaunchKernelGGL(sparse_fill_kernel, dim3(n, nblk), dim3(256), 0, st, logits, r0, r1, rowstat, blk, line_off, indptr, data, indices,
                       T_uniform, C, thr, capacity, line_T, row_off);
}

}  // namespace pocr
