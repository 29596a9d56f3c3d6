/* pocr.h — C ABI of the MI355X-native text-line recogniser (libpocr_hip.so).
 *
 * This is the drop-in boundary for pero-ocr's batched line-recognition hot path.
 * The reference has no native code; its Python engine does all device work inside
 *
 *   PytorchEngineLineOCR.run_ocr        pero_ocr/ocr_engine/pytorch_ocr_engine.py:59-74
 *   greedy_decode_ctc                   pero_ocr/ocr_engine/pytorch_ocr_engine.py:13-34
 *   BaseEngineLineOCR.process_lines     pero_ocr/ocr_engine/line_ocr_engine.py:121-129 (batch assembly + run_ocr)
 *
 * Each entry point below names the reference lines it replaces.  Plain pointers and
 * sizes only; the caller owns every host buffer, the library owns all device memory.
 * One engine = one GPU + one HIP stream; an engine is NOT thread-safe; every call
 * blocks until its outputs are in the caller's host buffers.
 * Every function returning int returns 0 on success, non-zero on failure with the
 * message available from pocr_last_error() (thread-local).
 */
#ifndef POCR_H
#define POCR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POCR_ABI_VERSION 13
#define POCR_NUM_SLOTS 4

typedef struct pocr_engine pocr_engine;

/* Network geometry ("vgg_blstm_ctc", see pero_ocr_amd/netspec.py).  Replaces what
 * torch.jit.load() reads from the TorchScript file (pytorch_ocr_engine.py:52-57). */
typedef struct pocr_config {
    int32_t abi_version;   /* POCR_ABI_VERSION */
    int32_t height;        /* line_px_height, multiple of 8 (line_ocr_engine.py:22) */
    int32_t num_classes;   /* C, blank = C-1 (pytorch_ocr_engine.py:12) */
    int32_t conv_out;      /* E: aggregation conv output channels, multiple of 16 */
    int32_t lstm_hidden;   /* multiple of 16 (POCR_ARCH_BLSTM) */
    int32_t lstm_layers;   /* >= 1          (POCR_ARCH_BLSTM) */
    int32_t arch;          /* POCR_ARCH_BLSTM or POCR_ARCH_SA */
    int32_t sa_layers;     /* POCR_ARCH_SA: encoder layers      (transformer.py:366-376, JSON encoder_layers) */
    int32_t sa_heads;      /* POCR_ARCH_SA: heads, conv_out/heads in {32, 64, 128} */
    int32_t sa_ff;         /* POCR_ARCH_SA: feed-forward width, multiple of 16 */
    int32_t dec_layers;    /* POCR_ARCH_S2S: decoder layers (transformer.py:13-47, JSON decoder_layers) */
    int32_t embed_num;     /* 0: no embeddings layer; > 0: style embeddings, table [embed_num + 1][2 * conv_out] after the
                              head in the weight blob (the engine JSON's embed_num, line_ocr_engine.py:32-35) */
} pocr_config;

/* Sequence model after the conv backbone: BiLSTM stack, or the self-attention encoder
 * (LineSelfAttentionEncoder, pero_ocr/ocr_engine/transformer.py:366-385 = BASELINE config 4). */
/* POCR_ARCH_S2S: conv backbone + self-attention encoder + autoregressive transformer decoder = the
 * reference's TransformerOCR (transformer.py:388-508) as driven by TransformerEngineLineOCR
 * (pero_ocr/ocr_engine/transformer_ocr_engine.py); num_classes then = symbols + boundary + ignore,
 * the sa_* fields describe encoder and decoder alike.  Only the pocr_s2s_* calls run such an engine. */
enum { POCR_ARCH_BLSTM = 0, POCR_ARCH_SA = 1, POCR_ARCH_S2S = 2 };

/* Style-embedding models (replaces `ids_embedding = LongTensor([embed_id] * N); model(batch_data, ids_embedding)`,
 * pytorch_ocr_engine.py:64-66): every line of every later launch is recognised with row `embed_id` of the embeddings
 * table, 0 <= embed_id <= embed_num; embed_num itself is the "mean" embedding (get_mean_embed_id, :49-50).
 * An engine whose config has embed_num > 0 refuses to launch until an id is set (the reference's TorchScript call
 * fails without the second argument); one without an embeddings layer refuses the call. */
int pocr_set_embed_id(pocr_engine *e, int32_t embed_id);

/* Number of float32 values pocr_create() expects (tensor order = netspec.tensor_table). */
size_t pocr_num_weight_floats(const pocr_config *cfg);

/* Build an engine on HIP device `device_id`: uploads and re-lays-out the weights
 * (MFMA fragment order), creates the stream.  Replaces _load_exported_model
 * (pytorch_ocr_engine.py:52-57).  Fails (non-zero) when no gfx950 device is usable:
 * there is NO CPU fallback. */
int pocr_create(const pocr_config *cfg, const float *weights, size_t n_floats,
                int device_id, pocr_engine **out);
void pocr_destroy(pocr_engine *e);

const char *pocr_last_error(void);
int pocr_abi_version(void);
/* Arithmetic of the convolution / GEMM kernels of this process (diagnostic; `bench.py` prices its roofline by it).  The
 * reference computes them in float32 (`aten::conv2d` behind `model(x)`, pytorch_ocr_engine.py:66,69).  2 = fp32 operands
 * as two f16 planes, three f16 MFMAs per 32-deep block (default); 3 = three bf16 planes, six MFMAs (POCR_CONV_SPLIT=3);
 * 0 = fp32 MFMA (POCR_CONV_FP32=1).  All three accumulate in fp32 and stay within the fp32 reference's own rounding noise. */
int pocr_conv_split(void);
/* f16x2 range guard (ABI 11).  The default arithmetic represents an fp32 operand as two f16 planes: fp32's precision, f16's
 * range.  Every kernel that produces such an operand records the largest |value| it wrote; a launch in which one reached
 * 65504 (or was not finite), or in which a whole activation tensor lay below 2^-13, is re-run - same lines, same requests,
 * transparently, at collect time - on the bf16x3 kernels (fp32's range) of a second engine.  Replaces
 * plain fp32 of pero_ocr/ocr_engine/pytorch_ocr_engine.py:61-69.  Returns the number of launches re-run so far.
 * The sequence-to-sequence engine does the same at pocr_s2s_decode (encoder and decoding loop again on the second engine, ABI 12);
 * POCR_CONV_SPLIT=3 selects bf16x3 for everything.
 * The second engine is built by a thread that pocr_create starts (ABI 13; ~0.2 s of host work laying the weights out again, off
 * every caller's path; POCR_FALLBACK_EAGER=0: built by the first launch that needs it, which then waits ~0.1 s for it).
 * Memory: a second copy of the weights (~90 MB for the recogniser of BASELINE.json) from creation on, and, for every slot the
 * second engine has served, activation buffers of that launch's size (17.6 MB per line at W_pad 576: 4.5 GB for a 256-line
 * launch) - those only once a launch of that slot has left the range.  What a re-run costs a caller: tools/fallback_cost.py. */
int64_t pocr_range_fallbacks(pocr_engine *e);
/* 1: the fall-back engine is there; 0: its builder is still running (wait != 0: block until it has finished); -1: this engine has
 * none (bf16x3 / fp32 arithmetic, POCR_FALLBACK_EAGER=0 before the first fall-back, or the build failed: it is then repeated by
 * the first launch that needs it, where the error is reported). */
int pocr_fallback_ready(pocr_engine *e, int32_t wait);
/* Resident BiLSTM recurrence (ABI 12).  One launch per layer hands the hidden state from step to step between co-resident
 * workgroups (torch.nn.LSTM of the reference's model, pero_ocr/ocr_engine/pytorch_ocr_engine.py:66-69); every wait is bounded.  A launch
 * in which a hand-off timed out (not all of a cluster's workgroups became resident: other tenants on the chip) is repeated - same
 * lines, same requests, at collect time - with one launch per step; the next 4 (8, 16 ... 256 after further timeouts) launches
 * then use the step kernels before the resident path is tried again.  Returns the number of launches repeated so far (this engine
 * and its range-guard fall-back engine).  POCR_LSTM_RESIDENT=0: step kernels only; POCR_LSTM_SPIN_LIMIT=<polls> at creation: the
 * bound of a wait (tests force the timeout path with 1). */
int64_t pocr_lstm_timeouts(pocr_engine *e);
/* Number of visible HIP devices (0 when none / no driver). */
int pocr_device_count(void);

/* ---- one padded batch: replaces run_ocr(batch_data) (pytorch_ocr_engine.py:59-74) ----
 * batch_nhwc : uint8 [n, height, w_pad, 3] (the array line_ocr_engine.py:121-127 builds);
 *              w_pad >= 4; T = (w_pad / 2) / 2 (two floor-mode 2x pools; = w_pad / 4 for the
 *              reference's widths, which are multiples of 32).
 * logits_ntc : float32 [n, T, C] or NULL  (= logits.permute(0,2,1), :72)
 * frame_argmax_nt : int32 [n, T] or NULL  (torch.argmax over C, first index wins, :19)
 * labels_nt  : int32 [n, T]  greedy-CTC-collapsed label ids, row i valid in [0, label_len_n[i])
 * label_len_n: int32 [n]
 */
int pocr_run_batch(pocr_engine *e, const uint8_t *batch_nhwc, int32_t n, int32_t w_pad,
                   float *logits_ntc, int32_t *frame_argmax_nt,
                   int32_t *labels_nt, int32_t *label_len_n);

/* ---- ragged chunk: fuses the zero-pad batch assembly (line_ocr_engine.py:121-127) into
 * the first kernel's HBM->LDS staging.  crops = the chunk's line crops packed back to
 * back, crop i = uint8 [height, widths[i], 3] starting at byte crop_offsets[i].
 * Every line is placed at x = pad_left (32 = line_padding_px, line_ocr_engine.py:54)
 * inside a zero row of w_pad pixels; columns beyond w_pad are dropped (:125-127).
 * pocr_stage_lines uploads the chunk and leaves it resident in HBM;
 * pocr_run_staged runs the network on the resident chunk (may be called repeatedly). */
int pocr_stage_lines(pocr_engine *e, const uint8_t *crops, const int64_t *crop_offsets,
                     const int32_t *widths, int32_t n, int32_t w_pad, int32_t pad_left);
int pocr_run_staged(pocr_engine *e, float *logits_ntc, int32_t *frame_argmax_nt,
                    int32_t *labels_nt, int32_t *label_len_n);

/* ---- stand-alone greedy CTC decode on the GPU: replaces greedy_decode_ctc(scores_probs, chars)
 * (pytorch_ocr_engine.py:13-34, 3-D branch) for caller-supplied scores.  logits_ntc: float32 [n, T, C]
 * (the reference takes [N, C, T]; transpose on the caller's side), blank = C-1.  No engine needed. */
int pocr_ctc_greedy(int device_id, const float *logits_ntc, int32_t n, int32_t T, int32_t C,
                    int32_t *frame_argmax_nt, int32_t *labels_nt, int32_t *label_len_n);

/* Transcription confidence of every line of the last COLLECTED sparse launch of `slot`, computed on the device from
 * the same kept set: what PageParser.compute_line_confidence / get_prob (pero_ocr/document_ocr/page_parser.py:485-496,
 * 437-450) return for the line's sparse logits read back through TextLine.get_dense_logits (core/layout.py:65-68,
 * dropped entries = -80): per frame the winner's probability, per run of equal winners its maximum, per line the
 * minimum over runs.  confidence_n: float32 [n]. */
int pocr_slot_confidence(pocr_engine *e, int32_t slot, float *confidence_n);

/* ---- stand-alone sparsification on the GPU of caller-supplied logits: replaces, for every line,
 *   probs = softmax(logits, axis=1); logits[probs < threshold] = 0; scipy.sparse.csc_matrix(logits)
 * (line_ocr_engine.py:168-171; the reference's own known-answer test: test/test_document_ocr/test_layout.py:10-26).
 * logits_ntc float32 [n, T, C].  Outputs: line_off int64 [n+1], indptr int32 [n][C+1], and the kept values / row
 * indices of all lines back to back in data / indices (capacity entries each; n*T*C always suffices).  No engine needed. */
int pocr_sparsify(int device_id, const float *logits_ntc, int32_t n, int32_t T, int32_t C, float threshold,
                  float *data, int32_t *indices, int64_t capacity, int32_t *indptr, int64_t *line_off);

/* ---- pipelined chunks (no reference counterpart: the reference runs its chunks strictly one
 * after the other, line_ocr_engine.py:80-129).  An engine has POCR_NUM_SLOTS independent slots, each
 * with its own HIP stream and activation buffers.  stage -> launch -> collect per slot; launch returns
 * as soon as the work is enqueued, so chunk k+1 can be staged and launched on the other slot while
 * chunk k is still running its latency-bound LSTM tail / copying results back.  The caller's input
 * buffers are free as soon as pocr_slot_stage_lines returns (they are copied to pinned memory);
 * the output buffers are written by pocr_slot_collect.  Outputs must be requested at launch.
 * pocr_stage_lines / pocr_run_staged / pocr_run_batch above are the blocking forms on slot 0. */
int pocr_num_slots(void);
int pocr_slot_stage_lines(pocr_engine *e, int32_t slot, const uint8_t *crops, const int64_t *crop_offsets,
                          const int32_t *widths, int32_t n, int32_t w_pad, int32_t pad_left);
/* Ragged staging: like pocr_slot_stage_lines, but every line i is padded to its OWN width w_pads[i]
 * (the W_pad of the reference chunk it belongs to, line_ocr_engine.py:81-82,121).  A line's padded width
 * is part of the numerical contract; which lines share a device launch is not: lines of several
 * reference chunks can be staged together and run as one launch sequence.  With ragged staging
 *   T_i = (w_pads[i] / 2) / 2,  rows = sum of T_i,  T_max = max T_i
 *   logits_ntc      : float32 [rows][C], line i = rows [sum_{j<i} T_j, +T_i)
 *   frame_argmax_nt : int32 [rows], same row order
 *   labels_nt       : int32 [n][T_max] (-1 padded), label_len_n : int32 [n]
 *   sparse indptr / line_off as documented below, row indices relative to the line. */
int pocr_slot_stage_ragged(pocr_engine *e, int32_t slot, const uint8_t *crops, const int64_t *crop_offsets,
                           const int32_t *widths, const int32_t *w_pads, int32_t n, int32_t pad_left);
int pocr_slot_launch(pocr_engine *e, int32_t slot, int32_t want_logits, int32_t want_argmax);
int pocr_slot_collect(pocr_engine *e, int32_t slot, float *logits_ntc, int32_t *frame_argmax_nt,
                      int32_t *labels_nt, int32_t *label_len_n);

/* Error recovery (no reference counterpart): drains both streams of `slot` and clears its staged / in-flight state, so
 * that an engine stays usable after a failure or an interrupt between launch and collect.  Results of the abandoned
 * launch are lost. */
int pocr_slot_reset(pocr_engine *e, int32_t slot);

/* ---- multi-GPU exchange (the reference is single-device).  Pages shard by the reference's own
 * chunks (line_ocr_engine.py:79-90: independent forwards), one process + one engine per GPU, and the only exchange
 * is ONE RCCL all-gather of the decoded label ids per page stream over xGMI.  librccl.so is dlopen()ed by the first
 * of these calls; a single-GPU process never loads it.
 *   pocr_comm_unique_id : rank 0 creates the 128-byte rendezvous id (ncclGetUniqueId) and hands it to the other ranks
 *                         through any out-of-band channel (the Python host uses a TCP socket, sharding.py).
 *   pocr_comm_init      : every rank, same id (ncclCommInitRank; blocks until all ranks have joined).
 *   pocr_allgather_labels: send = int32 [count] host buffer of THIS rank (fixed stride: every rank passes the same
 *                         count - all ranks derive it from the same chunk plan, so no size exchange is needed);
 *                         recv = int32 [world * count], rank r's block at r * count.  Blocking.
 *   pocr_comm_allreduce_max: in-place max over ranks of one double (also a barrier; bench.py's max-over-ranks time).
 *   pocr_comm_info      : what the communicator itself reports - *count = ncclCommCount, *rank = ncclCommUserRank (both 0
 *                         when the engine has no communicator).  A run that claims N ranks over RCCL prints this number.
 *   pocr_comm_destroy   : optional, pocr_destroy() does it too. */
#define POCR_UNIQUE_ID_BYTES 128
int pocr_comm_unique_id(uint8_t *id128);
int pocr_comm_init(pocr_engine *e, const uint8_t *id128, int32_t rank, int32_t world);
int pocr_comm_destroy(pocr_engine *e);
int pocr_allgather_labels(pocr_engine *e, const int32_t *send, int64_t count, int32_t *recv);
int pocr_comm_allreduce_max(pocr_engine *e, double *value);
int pocr_comm_info(pocr_engine *e, int32_t *count, int32_t *rank);
/* hipDeviceSynchronize() on the engine's device (fences of the measurement harness). */
int pocr_device_synchronize(pocr_engine *e);

/* ---- sparse logits: replaces `softmax -> logits[p < 1e-4] = 0 -> scipy.sparse.csc_matrix` per line
 * (line_ocr_engine.py:168-171, softmax.py:4-46) with device kernels, so only CSC triplets cross PCIe.
 * pocr_slot_launch_sparse = pocr_slot_launch(no dense logits) + sparsification with `threshold`
 * (the reference uses 0.0001).  row_begin/row_end (both int32 [n] or both NULL) restrict line i to the
 * frames [row_begin[i], row_end[i]) - the tight_crop_logits slice of line_ocr_engine.py:146-150; stored
 * row indices are relative to row_begin[i].
 * pocr_slot_sparse_nnz waits for the launch and returns the total number of kept entries, so the
 * caller can size `data` / `indices`.  pocr_slot_collect_sparse then fills:
 *   line_off [n+1] int64 : entries of line i are data/indices[line_off[i] .. line_off[i+1])
 *   indptr   [n][C+1]    : CSC column pointers of line i (starting at 0)
 *   data, indices        : values and row (frame) indices, column-major within a line, rows ascending
 * i.e. scipy.sparse.csc_matrix((data[a:b], indices[a:b], indptr[i]), shape=(rows_i, C)). */
int pocr_slot_launch_sparse(pocr_engine *e, int32_t slot, const int32_t *row_begin, const int32_t *row_end,
                            float threshold, int32_t want_argmax);
int pocr_slot_sparse_nnz(pocr_engine *e, int32_t slot, int64_t *total_nnz);
int pocr_slot_collect_sparse(pocr_engine *e, int32_t slot, float *data, int32_t *indices, int32_t *indptr,
                             int64_t *line_off, int32_t *frame_argmax_nt, int32_t *labels_nt, int32_t *label_len_n);

/* ---- sequence-to-sequence recognition (POCR_ARCH_S2S): replaces TransformerEngineLineOCR.run_ocr /
 * transcribe_batch (pero_ocr/ocr_engine/transformer_ocr_engine.py:32-89) for the batches that
 * BaseEngineLineOCR.process_lines builds in its "transformer" branches (line_ocr_engine.py:84-85,95-127).
 *
 * pocr_s2s_stage: like pocr_slot_stage_ragged with a left offset per line.  Line i is placed at
 *   x = pad_lefts[i] inside a zero row of w_pads[i] pixels - i.e. line_padding_px plus, for batches
 *   narrower than 1088 px, the centring offset of transformer_ocr_engine.py:36-40 (w_pads[i] is then 1088).
 * pocr_s2s_launch: enqueues the encoder (conv backbone, self-attention encoder, key/value projection of
 *   its output for every decoder layer) and returns.  batch_first [n_batches + 1]: lines
 *   [batch_first[b], batch_first[b+1]) form reference batch b (all with the same w_pad).  The batch is
 *   the unit of the decoding loop: it runs until every line of the batch has produced the boundary
 *   symbol or the step count exceeds w_pad / 4 (:74-80); batches of one launch are decoded side by side.
 * pocr_s2s_decode: runs the greedy decoding loop (blocking) and reports
 *   steps [n_batches] : decoding steps of each batch = rows of its logits (:81 torch.stack(logits))
 *   *s_max            : max over steps[]
 * pocr_s2s_collect: tokens [n][s_max] int32 = arg-max sample of every step (-1 beyond the line's batch);
 *   the reference feeds back / keeps the samples of steps 0 .. steps[b]-2 (partial_transcripts[1:], :82-84).
 *   logits [n][s_max][C] float32 or NULL (requested at pocr_s2s_decode), rows >= steps[b] are zero. */
int pocr_s2s_stage(pocr_engine *e, int32_t slot, const uint8_t *crops, const int64_t *crop_offsets,
                   const int32_t *widths, const int32_t *w_pads, const int32_t *pad_lefts, int32_t n);
int pocr_s2s_launch(pocr_engine *e, int32_t slot, const int32_t *batch_first, int32_t n_batches);
int pocr_s2s_decode(pocr_engine *e, int32_t slot, int32_t want_logits, int32_t *steps, int32_t *s_max);
int pocr_s2s_collect(pocr_engine *e, int32_t slot, int32_t *tokens, float *logits);
/* Sparse decoder logits: the softmax / p < threshold -> 0 / csc_matrix step of line_ocr_engine.py:168-171 on the
 * device, after pocr_s2s_decode (the logits stay resident; they need not have been requested).  Line i keeps its
 * rows [0, row_end[i]) - the reference cuts a line's logits to len(transcription) rows before sparsifying
 * (:183, :134-142) - so the caller first derives the lengths from the tokens.  Outputs as for
 * pocr_slot_collect_sparse: line_off [n+1], indptr [n][C+1], data / indices [total_nnz]. */
int pocr_s2s_sparse(pocr_engine *e, int32_t slot, const int32_t *row_end, float threshold, int64_t *total_nnz);
int pocr_s2s_collect_sparse(pocr_engine *e, int32_t slot, float *data, int32_t *indices, int32_t *indptr, int64_t *line_off);

/* ---- line cropper remap: replaces cv2.remap(img, map_x, map_y, INTER_LINEAR, BORDER_CONSTANT)
 * in EngineLineCropper.fast_remap (pero_ocr/core/crop_engine.py:146-163) for all lines of a page in one call.
 * page_hwc uint8 [H][W][C] (C <= 4); line i: coords + coord_off[i] = float32 [line_height][widths[i]][2] (x, y) source
 * position of every crop pixel (what get_crop_inputs returns, :54-99); its crop uint8 [line_height][widths[i]][C] is
 * written at crops + crop_off[i].  OpenCV's 8-bit fixed-point bilinear arithmetic (see csrc/crop.hpp).  No engine needed. */
int pocr_crop_lines(int device_id, const uint8_t *page_hwc, int32_t H, int32_t W, int32_t C, const float *coords,
                    const int64_t *coord_off, const int32_t *widths, int32_t n, int32_t line_height, uint8_t *crops,
                    const int64_t *crop_off);

/* The same with the sampling grid generated on the device from each line's 1-D curves - the [line_height x width]
 * float64 tail of get_crop_inputs (crop_engine.py:90-98) leaves the host, which keeps only the per-column work
 * (spline of the baseline, normals).  Line i: curves = float64 [4][widths[i]] (base_x, base_y, normal_x, normal_y),
 * packed back to back in line order; rows + i*line_height = float64 [line_height] offsets along the normal
 * (np.linspace(-up, down, line_height)); rot + 4*i = the 2x2 rotation R (row-major).  grid_out (or NULL):
 * float32 [line_height][widths[i]][2] grids back to back, for tests. */
int pocr_crop_curves(int device_id, const uint8_t *page_hwc, int32_t H, int32_t W, int32_t C, const double *curves,
                     const double *rows, const double *rot, const int32_t *widths, int32_t n, int32_t line_height,
                     uint8_t *crops, const int64_t *crop_off, float *grid_out);

/* ---- resident line cropper: the whole of EngineLineCropper.crop (pero_ocr/core/crop_engine.py:16-30) for all lines of a
 * page with the page in HBM and the per-column mathematics of get_crop_inputs (:73-89: walk the interpolated baseline at
 * unit steps, arc length, resample at the target resolution, normals) on the device in float64, bit-identical to the
 * reference's numpy / scipy sequence.  The host keeps the per-LINE scalars (:54-72): integer baseline, rotation R, and
 * the interpolant f - scipy.interpolate.interp1d(kind="cubic") as its B-spline (knots t, coefficients c), or np.poly1d. */
typedef struct pocr_cropper pocr_cropper;
typedef struct pocr_crop_spec {
    double x_min, x_max;   /* np.arange(x_min, x_max): the rotated baseline's extent (:73) */
    double lo, hi;         /* interp1d's domain: outside it the reference raises and crop() returns its fallback (mode 0 only) */
    double zoom;           /* target_height / (up + down) (:76) */
    double above, below;   /* scaled line heights: rows = np.linspace(-above, below, line_height) (:90) */
    double rot[4];         /* R row-major (:57) */
    int32_t mode;          /* 0: cubic B-spline; 1: polynomial, np.poly1d coefficient order (highest power first) */
    int32_t n_coef;        /* mode 0: number of B-spline coefficients (knots: n_coef + 4); mode 1: degree + 1 */
    int32_t coef_off, knot_off;   /* first coefficient / knot of this line in the arrays handed to pocr_cropper_measure */
    int32_t n_x;           /* len(np.arange(x_min, x_max)) */
    int32_t pad_;
} pocr_crop_spec;
int pocr_cropper_create(int device_id, pocr_cropper **out);
void pocr_cropper_destroy(pocr_cropper *c);
/* Starts the upload of the page (uint8 [H][W][C], C <= 4) on a helper thread and returns at once: the caller computes its
 * per-line splines meanwhile.  page_hwc must stay valid until pocr_cropper_wait_page / pocr_cropper_crop returns.
 * The page stays resident for any number of measure / crop calls. */
int pocr_cropper_set_page(pocr_cropper *c, const uint8_t *page_hwc, int32_t H, int32_t W, int32_t C);
int pocr_cropper_wait_page(pocr_cropper *c);
/* Arc length of every line -> widths[i] = int(arc[-1] * zoom) (the crop's width, :76); status[i] != 0: the evaluation left
 * the interpolant's domain (the reference raises -> fallback crop).  width 0 = empty grid (also the fallback). */
int pocr_cropper_measure(pocr_cropper *c, const pocr_crop_spec *specs, int32_t n, const double *knots, int64_t n_knots,
                         const double *coefs, int64_t n_coefs, int32_t *widths, int32_t *status);
/* Crops of the lines measured last: line i uint8 [line_height][widths[i]][C] at crop_off[i] - copied to `crops`, or, when
 * crops is NULL, left in the cropper's pinned host buffer (pocr_cropper_pinned_crops, valid until the next crop call).
 * grid_out (or NULL): the float32 [line_height][width][2] sampling grids back to back (lines with status 0), for tests.
 * status [n]: final per-line status. */
int pocr_cropper_crop(pocr_cropper *c, int32_t line_height, const int64_t *crop_off, uint8_t *crops, float *grid_out, int32_t *status);
const uint8_t *pocr_cropper_pinned_crops(pocr_cropper *c);
/* Crops that STAY in HBM between the cropper and the recogniser (reference: LineCropper.process_page fills line.crop,
 * page_parser.py:384-393, and PageOCR.process_page hands the same arrays to process_lines, :418-430 - on the host both).
 * pocr_cropper_crop_resident = pocr_cropper_crop without the copy back: the device buffer that holds the crops
 * (line i uint8 [line_height][widths[i]][3] at crop_off[i]) is detached from the cropper and returned as a handle, so the
 * next page can be cropped while the recogniser still reads this one.  pocr_slot_stage_resident stages lines straight
 * from such buffers (one handle per line: a launch may mix lines of several pages; same device as the engine; nothing is
 * copied - the handles must stay alive until the launch has been collected).  pocr_crops_read copies bytes back on
 * demand (callers that want the numpy crop after all); pocr_crops_release returns the buffer to the cropper's pool. */
typedef struct pocr_crops pocr_crops;
int pocr_cropper_crop_resident(pocr_cropper *c, int32_t line_height, const int64_t *crop_off, int32_t *status, pocr_crops **out);
void pocr_crops_release(pocr_crops *k);
int64_t pocr_crops_bytes(const pocr_crops *k);
int pocr_crops_read(const pocr_crops *k, int64_t offset, int64_t nbytes, uint8_t *out);
int pocr_slot_stage_resident(pocr_engine *e, int32_t slot, const pocr_crops *const *crops_of_line, const int64_t *crop_offsets,
                             const int32_t *widths, const int32_t *w_pads, int32_t n, int32_t pad_left);
/* Test hook: the float64 per-column curves of the last crop call - line i (status 0 at measure time): [4][widths[i]]
 * (base_x, base_y, normal_x, normal_y), lines back to back - for bit-comparison with the numpy / scipy sequence. */
int pocr_cropper_read_curves(pocr_cropper *c, double *out, int64_t cap);
float pocr_cropper_last_ms(pocr_cropper *c);

/* Host-side helper (no GPU): find_best_overlap of the transformer branch (line_ocr_engine.py:196-211, edit distance
 * pero_ocr/sequence_alignment.py:4-13) on two symbol-id sequences: the overlap length i in 1..min(na, nb) whose
 * suffix-of-a / prefix-of-b pair has the lowest character error rate (first such i; 0 when none is below 1).
 * Long lines are recognised in overlapping parts and every seam costs one such search: O(n^3), seconds in the
 * reference's Python, milliseconds here. */
int32_t pocr_best_overlap(const int32_t *a, int32_t na, const int32_t *b, int32_t nb);

/* ---- layout network: replaces Net.__init__ (torch.jit.load of the ParseNet model,
 * pero_ocr/layout_engines/torch_parsenet.py:8-20) and TorchParseNet.get_maps (:37-58) - area down-sampling by an
 * integer factor (cv2.resize INTER_AREA, :42), zero canvas padded to multiples of 64 (:44-47), uint8 * (1/255.) (:50),
 * the network, crop to the un-padded size (:56).  Network = this build's "parsenet_unet64" (pero_ocr_amd/parsenet_spec.py;
 * the reference's model is an opaque download); weights = float32 blob in parsenet_spec.tensor_table() order.
 *   img_hwc : uint8 [H][W][3] page;  downsample >= 1 (1 = no resize)
 *   out_hw5 : float32 [h][w][5] with (h, w) = pocr_parsenet_out_shape(H, W, downsample) = cvRound(H / ds), cvRound(W / ds);
 *             channels: 0, 1 line heights above / below the baseline, 2 baseline, 3 line end, 4 region border.
 * The host-side adaptive resolution loop (get_maps_with_optimal_resolution, :60-103) stays in the host language
 * (pero_ocr_amd/layout_engines/torch_parsenet.py).  Blocking; one handle = one GPU + one stream. */
typedef struct pocr_parsenet pocr_parsenet;
size_t pocr_parsenet_num_weight_floats(void);
int pocr_parsenet_create(const float *weights, size_t n_floats, int device_id, pocr_parsenet **out);
void pocr_parsenet_destroy(pocr_parsenet *p);
int pocr_parsenet_out_shape(int32_t h, int32_t w, int32_t downsample, int32_t *out_h, int32_t *out_w);
int pocr_parsenet_get_maps(pocr_parsenet *p, const uint8_t *img_hwc, int32_t H, int32_t W, int32_t downsample, float *out_hw5);
/* The same for a FRACTIONAL down-sampling factor (get_maps_with_optimal_resolution remembers one after the first page,
 * torch_parsenet.py:60-93): the area resample runs on the device from separable tap tables the caller builds - output row o
 * = sum over a < taps_y of wy[o][a] * source row min(y0[o] + a, H - 1), columns alike; float64, rows first, products rounded
 * before they are added, then rint - exactly the host restatement `resize_area` (parity of cv2.resize(INTER_AREA) itself:
 * unpinned, no OpenCV in the build image).  out_hw5: float32 [out_h][out_w][5]. */
int pocr_parsenet_get_maps_area(pocr_parsenet *p, const uint8_t *img_hwc, int32_t H, int32_t W, int32_t out_h, int32_t out_w,
                                const double *wy, const int32_t *y0, int32_t taps_y, const double *wx, const int32_t *x0, int32_t taps_x,
                                float *out_hw5);
/* GPU time in ms of the last get_maps between the end of the upload and the end of the last kernel (HIP events). */
int pocr_parsenet_last_ms(pocr_parsenet *p, float *ms);
/* f16x2 range guard of the layout network (ABI 12): a page on which a conv layer's activation reached 65504 (or was not finite) or
 * lay below 2^-13 as a whole is run again - transparently, inside the same get_maps call - on the bf16x3 kernels (fp32's range)
 * of a second network created on first use; torch_parsenet.py:49-53 computes in plain fp32.  Returns the pages re-run so far. */
int64_t pocr_parsenet_range_fallbacks(pocr_parsenet *p);

/* ---- measurement / test taps (not part of the reference surface) ----
 * Per-stage GPU time of the last pocr_run_* call in milliseconds, measured with HIP
 * events on the engine's stream.  Stage ids: POCR_STAGE_*.  Returns the number of
 * stages written (<= cap). */
enum {
    POCR_STAGE_CONV1 = 0,   /* stage+normalise+conv1 (u8 -> f32, K0 fused with conv1) */
    POCR_STAGE_CONV2, POCR_STAGE_CONV3, POCR_STAGE_CONV4, POCR_STAGE_CONV5,
    POCR_STAGE_CONV6, POCR_STAGE_CONV7, POCR_STAGE_CONV8, POCR_STAGE_CONV9,
    POCR_STAGE_AGG,         /* aggregation conv */
    POCR_STAGE_LSTM,        /* sequence model: BiLSTM layers (projections + recurrence) or the self-attention encoder */
    POCR_STAGE_HEAD,        /* projection to C classes */
    POCR_STAGE_CTC,         /* argmax + collapse */
    POCR_STAGE_TOTAL,       /* first kernel start -> last kernel end */
    POCR_NUM_STAGES
};
int pocr_last_stage_ms(pocr_engine *e, float *ms, int32_t cap);
int pocr_slot_stage_ms(pocr_engine *e, int32_t slot, float *ms, int32_t cap);
/* Enable/disable per-stage event recording (default off: events cost a little). */
int pocr_set_profiling(pocr_engine *e, int32_t enabled);

/* Copy an intermediate activation of the last run to the host (tests only).
 * what: 0..8 = output of conv1..conv9 (NHWC, after activation/pool/BN),
 *       9 = aggregation features [n, T, E];
 *       BLSTM: 10+l = BiLSTM layer l output [n, T, 2*hidden];
 *       SA / S2S: 10 = LayerNorm + positional encoding [n, T, E], 11+l = encoder layer l output [n, T, E].
 * In the default (f16x2) mode conv1 runs inside conv2's prologue and its activation is not kept: what = 0 computes it on
 * demand from the crops still staged in the slot (same arithmetic, same bits); conv activations kept in the two-plane f16
 * layout are converted to the fp32 values they stand for.
 * Writes min(cap, size) floats, stores the full size in *n_floats. */
int pocr_debug_read(pocr_engine *e, int32_t what, float *out, size_t cap, size_t *n_floats);

#ifdef __cplusplus
}
#endif
#endif /* POCR_H */
