#!/usr/bin/env python3
"""bench.py — throughput of the batched text-line recognition hot path on MI355X.

Metric (BASELINE.json): text-line crops/s, CTC-decoded, at 40x512.

Workloads (--workload, default c2):
  c2  BASELINE configs[1]: 256 synthetic 40x512 line crops = ONE reference chunk (batch_size 274 -> 480*274//512 =
      256 lines, W_pad 576, T 144), VGG+BiLSTM+CTC engine, C = 232.  One "step" = one pass of the hot path over that
      chunk.  Two timed regions in one run:
        value        crops already resident in HBM -> staging/normalise -> conv backbone -> BiLSTM -> head -> greedy
                     CTC -> label ids on the host -> strings                                  (the contract's `value`)
        end_to_end   the same, but every step packs its 256 crops from host memory and uploads them (H2D) first:
                     what PytorchEngineLineOCR.process_lines does per launch
      With N > 1 GPUs every rank runs its own chunk (weak scaling; chunks are independent units) and the decoded
      labels are all-gathered over RCCL (one fixed-stride ncclAllGather per step through the C ABI).
  c3  BASELINE configs[2]: a 2048-line page stream, widths uniform in 128..1024, the reference's default batch_size 8
      (300+ chunks), chunk-sharded over the ranks (LPT), one all-gather of the labels per pass - STRONG scaling.
      One "step" = the whole stream; inputs come from host memory every step (that is the path).
  c4  BASELINE configs[3]: 256 x 40x768, self-attention encoder instead of the BiLSTM (W_pad 832, T 208).
  c5  BASELINE configs[4]: end to end on a synthetic 4k x 3k page - layout network (ParseNet contract, downsample 4) ->
      layout post-processing STUB (baselines = the page generator's ground truth; cnn_layout_engine is out of scope) ->
      line cropper -> line OCR (c2's engine, default batch_size 8) -> strings.  One "step" = one page; pages/s.
      With N GPUs every rank processes its own pages (pages are independent jobs: no collective in the data path).

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5]
With --gpus N > 1 and no WORLD_SIZE in the environment bench.py starts the N ranks itself
(python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...); under torchrun it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*.  Fewer than N visible GPUs is an error, never a silent 1-GPU run.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0      # same guide: dense bf16 MFMA peak (v_mfma_f32_32x32x16 / 16x16x32_bf16)
# conv2..conv9, the aggregation conv and the GEMM-mode layers run on the 16-bit matrix pipe with fp32-level accuracy
# (csrc/conv_bf16x3.hpp); the library says which operand split this process uses (pocr_conv_split):
#   2 (default)          f16x2: every fp32 operand = two f16 planes (h + l / 2048), THREE v_mfma_f32_16x16x32_f16 per 32-deep
#                        product block -> ceiling for ALGORITHMIC fp32 FLOPs = 2500 / 3 = 833.3 TFLOP/s
#   3 (POCR_CONV_SPLIT=3) bf16x3: three bf16 planes (exact), SIX v_mfma_f32_16x16x32_bf16 per block -> 2500 / 6 = 416.7
#   0 (POCR_CONV_FP32=1)  fp32 MFMA kernels -> 157.3
MFMA_PER_BLOCK = {2: 3.0, 3: 6.0, 0: 1.0}
SPLIT_NAME = {2: "f16x2", 3: "bf16x3", 0: "fp32 MFMA"}


def conv_peak_tflops(split):
    return BF16_MFMA_PEAK_TFLOPS / MFMA_PER_BLOCK[split] if split else F32_MFMA_PEAK_TFLOPS
HEIGHT = 40
WORKLOADS = {
    # fixture = tests/golden/<name>: weight seed / kwargs / calibrated head bias (and, for c3, the page stream itself)
    "c2": dict(fixture="c2", n_lines=256, width=512, batch_size=274, crop_seed=305),
    "c3": dict(fixture="c3", batch_size=8),
    "c4": dict(fixture="c4", n_lines=256, width=768, batch_size=410, crop_seed=501),
    "c5": dict(fixture="c2", batch_size=8, page_h=3072, page_w=4096),
}


class Dev:
    type = "cuda"

    def __init__(self, index):
        self.index = index


def fixture_model(name):
    """(meta, spec, weights) of a golden fixture: the seeded weights plus the calibrated tensors it stores."""
    from pero_ocr_amd import netspec
    with open(os.path.join(REPO, "tests", "golden", f"{name}.json"), encoding="utf8") as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(REPO, "tests", "golden", f"{name}.npz"))
    spec = netspec.NetSpec.from_json(meta["spec"])
    weights = netspec.generate_weights(spec, meta["weight_seed"], **meta.get("weight_kwargs", {}))
    for k in arrays.files:
        if k.startswith("override_"):
            weights[k[len("override_"):]] = arrays[k]
    return meta, spec, weights


def conv_flops_per_line(w_pad, height=HEIGHT, conv_out=512):
    """Algorithmic FLOPs (2*MAC) of each conv-backbone kernel for ONE line (SURVEY.md section 8d:
    12,193,280 MAC per padded input column at H=40 -> 14.047 GFLOP at W_pad 576)."""
    from pero_ocr_amd.netspec import CONV_PLAN
    h, w, out = height, w_pad, {}
    for i, (cin, cout, _a, (ph, pw)) in enumerate(CONV_PLAN, start=1):
        out[f"conv{i}"] = 2.0 * h * w * cout * cin * 9
        h, w = h // ph, w // pw
    out["agg"] = 2.0 * w * conv_out * 512 * h
    return out


def physical_cores():
    try:
        pairs = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                elif not ln.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        return len(pairs) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def kernel_trace_summary(workload, fl, n_lines, peak):
    """From the committed rocprofv3 kernel trace of this bench (profiles/rNN_bench_<workload>_kernel_stats.txt, newest round):
    the kernel with the longest average duration and the share of the sequence stage in the summed kernel time - the `roofline`
    object names conv9, the largest kernel by FLOPs; under overlapping launches other kernels can last longer per call."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r[0-9][0-9]_bench_{workload}_kernel_stats.txt")))
    if not files:
        return None
    rows = []
    for ln in open(files[-1]):
        m = re.match(r"\s*(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+(.*)", ln)
        if m:
            rows.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), m.group(5).strip()))
    if not rows:
        return None
    total = sum(r[1] for r in rows)
    longest = max((r for r in rows if "pocr::" in r[4]), key=lambda r: r[2])
    # algorithmic FLOPs of the conv kernels by their template signature (fused conv1+2: the ...true> at the end; conv9: BN = true)
    sig = {"conv1+2": "10, 1, 1, 1, 2, 2, 1, false, 2, true, 3, 3, 1, 1, false, 2, true, true, true>",
           "conv9": "5, 1, 2, 1, 1, 1, 2, true, 2, true, 3, 3, 1, 1, false, 2, true, true, false>",
           "conv9 ": "conv3x3_rows_kernel<5, 1, 2, 1, 1, 1, 2, true, 2, false>"}        # (round 5: conv3 .. conv9 on csrc/conv_rows.hpp)
    name = next((k.strip() for k, v in sig.items() if v in longest[4]), None)
    flops = None
    if name == "conv9":
        flops = fl["conv9"] * n_lines
    elif name == "conv1+2":
        flops = (fl["conv1"] + fl["conv2"]) * n_lines
    seq = sum(r[1] for r in rows if "lstm_" in r[4])
    out = {"source": os.path.relpath(files[-1], REPO),
           "longest_kernel_in_trace": {"kernel": longest[4][:140], "layer": name, "avg_ms": round(longest[2] / 1e3, 4), "calls": longest[0],
                                       "share_of_kernel_time": round(longest[1] / total, 4),
                                       "frac": round(flops / (longest[2] * 1e-6) / 1e12 / peak, 4) if flops else None},
           "sequence_stage_share_of_kernel_time": round(seq / total, 4)}
    return out


def cpu_baseline_measure(workload):
    """Runs in the CHILD process `cpu_baseline` starts (threads bound to cores before the OpenMP runtime exists).
    The oracle (PyTorch-CPU restatement of the reference path, parity-pinned against the imported reference) timed on this box's
    host cores, ONE protocol throughout: a pass = the step's chunk as forward passes of 64 lines, every one padded to the chunk's
    W_pad (the reference's per-line arithmetic; oneDNN threads a 64-line batch better than the whole chunk).  Per thread count of
    the sweep: one discarded 64-line warm-up AT THAT COUNT, then one timed pass; at the best count three more timed passes, whose
    median is the baseline - the same thing measured four times, so the two figures must agree (`agreement`).  Last, and only
    reported: all lines of the chunk in ONE forward pass, what the reference does with this batch_size."""
    try:                                         # BEFORE the OpenMP runtime starts: it binds this thread to the first place
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    import torch
    from oracle import engine_oracle, model_oracle
    from pero_ocr_amd import synth
    torch.set_num_interop_threads(1)
    wl = WORKLOADS[workload]
    _meta, spec, weights = fixture_model(wl["fixture"])
    width, batch_size = wl["width"], wl["batch_size"]
    crops = synth.make_crops(wl["crop_seed"], [width] * wl["n_lines"], spec.height)
    net = model_oracle.OracleNet(spec, weights)
    chars = [""] * spec.num_classes
    max_width = -(-width // 32) * 32

    def one_pass(ids):
        batch = engine_oracle.assemble_batch(crops, ids, spec.height, max_width, 480 * batch_size)
        nct = model_oracle.forward_logits(net, batch)
        _best, labels = engine_oracle.greedy_ctc(nct)
        return [engine_oracle.labels_to_text(l, chars) for l in labels]

    n = len(crops)
    piece = min(64, n)
    parts = [list(range(k, min(k + piece, n))) for k in range(0, n, piece)]

    def timed_pass(pp):
        t0 = time.perf_counter()
        for p in pp:
            one_pass(p)
        return sum(len(p) for p in pp) / (time.perf_counter() - t0)

    phys, logical = physical_cores(), os.cpu_count() or 1
    cands = sorted({c for c in (8, 16, 32, 64, phys) if c <= usable}) or [usable]
    sweep = {}
    t_sweep = time.perf_counter()
    for c in cands:                              # ascending; stop once more threads clearly lose (oversubscribed small convs)
        if sweep and time.perf_counter() - t_sweep > 25.0:
            break
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        one_pass(parts[0])                       # warm-up at THIS thread count (thread team, oneDNN primitives, scratchpads)
        if not sweep:                            # bounded sample: a pass of at most ~4 s at the first count's rate
            parts = parts[:max(1, min(len(parts), int(4.0 / max(time.perf_counter() - t0, 1e-3))))]
            n = sum(len(p) for p in parts)
        sweep[c] = timed_pass(parts)
        if sweep[c] < 0.8 * max(sweep.values()):
            break
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    one_pass(parts[0])
    by64 = [timed_pass(parts) for _ in range(3)]
    med_64 = float(np.median(by64))
    agreement = med_64 / sweep[best]
    # the whole chunk in one forward pass: bounded to ~12 s (one warm-up + two timed passes when the chunk takes < 4 s)
    t0 = time.perf_counter()
    one_pass(list(range(n)))
    t_whole = time.perf_counter() - t0
    whole = [n / t_whole] if t_whole > 4.0 else [timed_pass([list(range(n))]) for _ in range(2)]
    out = {"value": round(med_64, 2), "unit": "lines/s", "cores": int(best), "threads": int(best), "physical_cores": int(phys),
           "logical_cpus": int(logical), "usable_cpus": int(usable), "kind": "port",
           "sample": f"{n} of the step's {len(crops)} 40x{width} crops as {len(parts)} forward passes of {piece} lines, every pass padded to W_pad "
                     f"{max_width + 64} like the step's chunk: median of 3 timed passes at {best} threads - the best count of a sweep that "
                     f"times the SAME pass once per count after a warm-up at that count (sweep figure at {best}: {sweep[best]:.1f} lines/s, "
                     f"ratio {agreement:.2f}); threads bound to cores (OMP_PROC_BIND=close, OMP_PLACES=cores), one inter-op thread; "
                     f"the same {n} lines in ONE forward pass: {float(np.median(whole)):.1f} lines/s; torch {torch.__version__} CPU fp32",
           "protocol": "pass = chunk as forward passes of 64 lines; sweep: warm-up + 1 timed pass per thread count; baseline: 3 more timed passes at the best count",
           "by_64_lines_per_s": [round(r, 2) for r in by64],
           "thread_sweep_lines_per_s": {str(k): round(v, 2) for k, v in sweep.items()},
           "agreement_with_sweep": round(agreement, 3),
           "whole_chunk_lines_per_s": [round(r, 2) for r in whole],
           "omp_env": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_NUM_THREADS")}}
    if not 0.85 <= agreement <= 1.15:
        # same pass, same thread count, measured seconds apart: a gap this large is the box (another tenant, clock), not the protocol
        one_pass(parts[0])
        again = timed_pass(parts)
        out["agreement_note"] = (f"timed passes and the sweep's pass at {best} threads differ by more than 15 %; the pass timed once more: "
                                 f"{again:.1f} lines/s - the host's rate itself moves between seconds (shared box)")
    return out


def cpu_baseline(workload):
    """The CPU baseline of the bench line: `cpu_baseline_measure` in a child process whose OpenMP threads are bound to cores
    (OMP_PROC_BIND / OMP_PLACES are read when the runtime starts, i.e. before `import torch` - hence a child).  Unbound, the
    sweep and the timed passes of round 5 disagreed by 2.1x on the 2-socket host: threads and pages migrated between the
    sockets when the thread count changed."""
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--workload", workload],
                       env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        raise RuntimeError(f"cpu baseline child failed (rc {p.returncode}): {p.stderr[-400:]}")
    return json.loads(lines[-1])


def default_contract_region(engine, eng, crops, n_lines, width, wl, spec, weights, chars, tmp, local_rank):
    """The reference's DEFAULT call - process_lines(crops) returns sparse logits (line_ocr_engine.py:57,168-171) - on a stream of
    8 x 256 lines @40x512 in one call (8 reference chunks of 256 lines, pipelined inside process_lines): host crops ->
    strings + scipy CSC logits + logit_coords on the host.  Twice: with the fixture's seeded weights as they are (flat posteriors:
    most classes stay above p >= 1e-4, the "sparse" logits are nearly dense) and with the CTC head scaled x8 (peaked posteriors,
    the regime of a trained recogniser; the same scaling `--workload c5` uses)."""
    from pero_ocr_amd import netspec
    from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR
    reps = 8
    big = [crops[i % n_lines] for i in range(reps * n_lines)]
    out = {"what": f"engine.process_lines(list of {reps * n_lines} crops @40x{width}) with the reference's default arguments "
                   "(sparse_logits=True): strings, scipy.sparse.csc_matrix logits and logit_coords on the host; inputs are host "
                   "numpy crops (PCIe-inclusive)", "unit": "lines/s"}

    def timed(e_):
        e_.process_lines(big)                                 # warm-up: same call (pinned buffers and the speculative copy size settle)
        e_.model.device_synchronize()
        dts = []
        for _ in range(3):                                    # median of three calls
            t0 = time.perf_counter()
            tr, lg, _lc = e_.process_lines(big)
            e_.model.device_synchronize()
            dts.append(time.perf_counter() - t0)
        dt = sorted(dts)[1]
        nnz = sum(m.nnz for m in lg) / float(sum(m.shape[0] for m in lg))
        assert len(tr) == len(big) and tr[:n_lines] == tr[n_lines:2 * n_lines]
        # The same calls as a STREAM: call k + 1 is begun (planned, its first launches enqueued) before call k is ended - the two
        # halves of process_lines that document_ocr.page_stream uses - so the engine's launch pipeline is not drained between
        # calls.  A lone call pays the pipeline's fill and tail once (one launch's latency minus one period, ~8 ms of a 79 ms
        # call: profiles/r04_launch_timeline.txt); a caller with further pages does not.  Same plan, same launches, same results.
        n_stream = 6

        def stream(k):
            """-> the last call's result; every call's strings are checked as it ends (results are not kept: 12 k live csc_matrix
            objects make every allocation's garbage-collection pass longer, which is the caller's business, not the engine's)"""
            ticket, last, ok = e_.process_lines_begin(big), None, True
            for _ in range(k - 1):
                nxt = e_.process_lines_begin(big)
                last = e_.process_lines_end(ticket)
                ok = ok and last[0] == tr
                ticket = nxt
            last = e_.process_lines_end(ticket)
            e_.model.device_synchronize()
            return last, ok and last[0] == tr
        stream(2)                                             # (warm: the speculative read-back size settles on this launch pattern)
        # A collected heap at the start of the region: the calls above left ~10^5 tracked objects (csc_matrix instances, their
        # dicts and arrays) in the young generations, and the full collection they trigger fell INTO the stream of whichever
        # engine was measured second - 24.9 k instead of 27.8 k lines/s, either engine (profiles/r06_default_call_stream.txt)
        import gc
        gc.collect()
        t0 = time.perf_counter()
        last, ok = stream(n_stream)
        dt_stream = (time.perf_counter() - t0) / n_stream
        assert ok and all(a.nnz == b.nnz for a, b in zip(last[1], lg))
        del last
        # ... and the SAME call without logits (process_lines(crops, no_logits=True), one call at a time): what the sparse logits cost a
        # call is the difference to this figure - not the difference to `value`, whose step loop never drains its pipeline
        e_.process_lines(big, no_logits=True)
        e_.model.device_synchronize()
        dts_nl = []
        for _ in range(3):
            t0 = time.perf_counter()
            e_.process_lines(big, no_logits=True)
            e_.model.device_synchronize()
            dts_nl.append(time.perf_counter() - t0)
        dt_nl = sorted(dts_nl)[1]
        e_.process_lines(big)                                 # (leave the read-back size as a sparse call left it)
        return {"value": round(len(big) / dt, 1), "ms_per_256_lines": round(1e3 * dt / reps, 3), "nnz_per_frame": round(nnz, 1),
                "calls_ms": [round(1e3 * d, 1) for d in dts],
                "same_call_no_logits": {"value": round(len(big) / dt_nl, 1), "calls_ms": [round(1e3 * d, 1) for d in dts_nl],
                                        "sparse_logits_cost_ms_per_call": round(1e3 * (dt - dt_nl), 2)},
                "streamed": {"value": round(len(big) / dt_stream, 1), "ms_per_256_lines": round(1e3 * dt_stream / reps, 3), "calls": n_stream,
                             "what": "the same calls with call k + 1 begun before call k is ended (process_lines_begin / _end)"}}

    out["seeded_weights"] = timed(engine)
    w8 = dict(weights)
    w8["head.weight"] = w8["head.weight"] * np.float32(8.0)
    w8["head.bias"] = w8["head.bias"] * np.float32(8.0)
    netspec.save_blob(os.path.join(tmp.name, "weights_peaked.pocrw"), spec, w8)
    with open(os.path.join(tmp.name, "ocr_peaked.json"), "w", encoding="utf8") as f:
        json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "weights_peaked.pocrw",
                   "characters": chars[:-1], "net_name": "bench"}, f)
    peaked = PytorchEngineLineOCR(os.path.join(tmp.name, "ocr_peaked.json"), Dev(local_rank), batch_size=wl["batch_size"])
    peaked.model.fallback_ready(wait=True)
    out["head_x8"] = timed(peaked)
    out["value"] = out["head_x8"]["value"]
    del peaked
    return out


def run_extra_workloads(timeout_s=170.0):
    """BASELINE configs 3, 4, 5 measured by this same script in child processes (one engine each), condensed into objects of
    the ONE JSON line the default run prints."""
    jobs = {"c3": ["--workload", "c3", "--steps", "2", "--warmup", "1"],
            "c4": ["--workload", "c4", "--steps", "5", "--warmup", "2"],
            # (64 pages = 16 recogniser calls of four pages: with 16 pages the stream's fill and tail - one call's latency - were a quarter of
            # the region, and the figure moved by 10 % between runs of the same build: 67.6 / 70.5 here against 72.9-75.6 for 20 pages)
            "c5": ["--workload", "c5", "--steps", "64", "--warmup", "8"]}
    only = os.environ.get("POCR_BENCH_EXTRAS")          # e.g. "c3,c5" or "" (none): which child workloads to run
    if only is not None:
        jobs = {k: v for k, v in jobs.items() if k in only.split(",")}
    out = {}
    for name, argv in jobs.items():
        t0 = time.perf_counter()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv + ["--no-cpu-baseline", "--no-extras"],
                               capture_output=True, text=True, timeout=timeout_s)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
            r = json.loads(line)
            keep = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "scaling") if k in r}
            keep["config"] = r["config"]["workload"]
            for k in ("roofline", "conv_backbone", "encoder", "page_at_a_time", "lines_per_s", "lines_per_page",
                      "sparse_logits_nnz_per_frame", "end_to_end"):
                if k in r:
                    keep[k] = r[k] if not isinstance(r[k], dict) else {kk: vv for kk, vv in r[k].items()
                                                                         if not isinstance(vv, (str, dict)) or kk in ("kernel", "stage_ms_per_page")}
            keep["wall_s"] = round(time.perf_counter() - t0, 1)
            out[name] = keep
        except Exception as exc:           # noqa: BLE001 - the main line must not be lost to an extra
            out[name] = {"value": None, "error": f"{type(exc).__name__}: {str(exc)[:300]}"}
    return out


def shape_rccl_fields(result, rccl_fields, collective):
    """A figure whose exchange ran over the gloo FALLBACK must not be readable as the RCCL result: `value` becomes null,
    the measured figure moves to `value_gloo_fallback` (compute per rank is real, the collective is not the product's)."""
    out = dict(rccl_fields)
    if collective.startswith("gloo FALLBACK"):
        out["value_gloo_fallback"] = result["value"]
        out["value"] = None
        out["rccl_ranks"] = 0
    return out


class c_stdout_to_stderr:
    """librccl prints a version banner through C stdio on stdout; the contract is ONE JSON line there.  File descriptor 1 points
    at stderr while the communicator is set up, and C stdio is flushed before it is restored."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def rank_launch_plan(n, argv, port=None, environ=None):
    """What `--gpus N` without a launcher starts: the command line and the environment of the N ranks (one process per
    GPU through torch.distributed.run, which sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* per rank; a rank binds GPU
    LOCAL_RANK).  Rendezvous on 127.0.0.1 (the container's hostname may not resolve) at a free port; the dmabuf IPC
    mode RCCL needs on this host driver is kept (or set) in the children's environment."""
    import socket
    environ = os.environ if environ is None else environ
    if port is None:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(environ, HSA_ENABLE_IPC_MODE_LEGACY=environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return cmd, env


def spawn_ranks(n, argv):
    """--gpus N without a launcher: start the N ranks ourselves, one process per GPU."""
    from pero_ocr_amd import _native
    have = _native.device_count()
    if have < n and os.environ.get("POCR_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible")
    cmd, env = rank_launch_plan(n, argv)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true",
                    help="default c2 run on one GPU: skip the extra objects (c2 with the reference's default sparse-logits call, c3, c4, c5)")
    ap.add_argument("--head-temperature", type=float, default=8.0,
                    help="c5: scale of the CTC head of the synthetic weights (8: ~8 classes per frame above p = 1e-4; 1: 219 of 232)")
    ap.add_argument("--pages-per-batch", type=int, default=4, help="c5: pages whose lines share one process_lines call")
    ap.add_argument("--front-workers", type=int, default=int(os.environ.get("POCR_BENCH_FRONTS", "2")),
                    help="c5: (layout network, cropper) pairs, one helper thread each, working on consecutive pages of the stream")
    args = ap.parse_args()
    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline_measure(args.workload)), flush=True)
        return
    if args.steps is None:
        args.steps = 5 if args.workload == "c3" else 20
    if args.warmup is None:
        args.warmup = 1 if args.workload == "c3" else 3 if args.workload != "c5" else 2

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("POCR_BENCH_SHARE_GPU") == "1":
        local_rank = 0          # test hook: N ranks on ONE GPU (RCCL refuses duplicate devices -> exercises the gloo fallback and
                                # the N-rank control flow on a single-GPU box; the numbers mean nothing)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a run of a different size")

    from pero_ocr_amd import _native, netspec, sharding, synth
    from pero_ocr_amd.ocr_engine.line_ocr_engine import Chunk, Launch
    from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR, labels_to_strings
    if _native.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: GPU {local_rank} not visible ({_native.device_count()} device(s))")

    wl = WORKLOADS[args.workload]
    meta, spec, weights = fixture_model(wl["fixture"])
    chars = meta["characters"]
    if args.workload == "c5" and args.head_temperature != 1.0:
        # seeded random weights give nearly flat posteriors: 219 of 232 classes per frame stay above the reference's
        # p >= 1e-4 sparsification threshold (line_ocr_engine.py:168-171) and the "sparse" logits of a page are 59 MB.
        # A trained recogniser is peaked; scaling the head restores that regime (nnz per frame is reported).
        weights = dict(weights)
        weights["head.weight"] = weights["head.weight"] * np.float32(args.head_temperature)
        weights["head.bias"] = weights["head.bias"] * np.float32(args.head_temperature)
    tmp = tempfile.TemporaryDirectory(prefix="pocr_bench_")
    netspec.save_blob(os.path.join(tmp.name, "weights.pocrw"), spec, weights)
    with open(os.path.join(tmp.name, "ocr.json"), "w", encoding="utf8") as f:
        json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "weights.pocrw",
                   "characters": chars[:-1], "net_name": "bench"}, f)
    engine = PytorchEngineLineOCR(os.path.join(tmp.name, "ocr.json"), Dev(local_rank), batch_size=wl["batch_size"])
    eng = engine.model
    eng.fallback_ready(wait=True)      # the range guard's bf16x3 engine is built on a thread behind pocr_create: not inside a timed region
    # Launches in flight in the c2 / c4 step loop.  The product's process_lines keeps THREE in flight (pipeline_depth: a ragged
    # stream's launches complete in pairs with two and the host assembles results in between); this loop runs one uniform chunk
    # per step with nothing for the host to assemble, and two is its measured optimum (8.98 against 9.19 ms per step with three,
    # profiles/r04_launch_timeline.txt) - `extra.c2_sparse`, `extra.c3` and `extra.c5` go through process_lines at the product's depth.
    n_slots = min(eng.num_slots, 2)
    if os.environ.get("POCR_BENCH_SLOTS"):   # experiments
        n_slots = max(1, min(eng.num_slots, int(os.environ["POCR_BENCH_SLOTS"])))

    # the exchange step: RCCL through the C ABI (POCR_FORCE_DIST=1 exercises it with a single rank)
    transport = None
    collective = "none"
    rccl_fields = {}
    if world > 1 or os.environ.get("POCR_FORCE_DIST") == "1":
        collective = "rccl (pocr_allgather_labels, C ABI)"
        if world == 1:
            with c_stdout_to_stderr():
                transport = sharding.init_rccl_from_env(eng, rank, world)
                transport.barrier()                   # (the banner comes with the first collective)
        else:
            # N ranks: RCCL through the C ABI is the product's exchange.  The bench must still print its line if that
            # cannot be set up on the box it lands on, so the ranks first agree (over a gloo group used for nothing else)
            # whether every one of them got its communicator; otherwise all of them carry the exchange over gloo and the
            # JSON line says so.
            import threading
            box = {}

            def _init():
                try:
                    box["t"] = sharding.init_rccl_from_env(eng, rank, world)
                except BaseException as exc:          # noqa: BLE001
                    box["err"] = f"{type(exc).__name__}: {exc}"
            th = threading.Thread(target=_init, daemon=True)
            with c_stdout_to_stderr():
                th.start()
                th.join(timeout=float(os.environ.get("POCR_RCCL_INIT_TIMEOUT", "180")))
                if "t" in box:
                    try:
                        box["t"].barrier()            # (the banner comes with the first collective)
                    except BaseException as exc:      # noqa: BLE001
                        box["err"] = f"{type(exc).__name__}: {exc}"
                        del box["t"]
            with c_stdout_to_stderr():                 # (gloo announces its peers on stdout)
                import torch.distributed as dist       # only now: PyTorch brings its own librccl / HSA runtime into the process
                dist.init_process_group("gloo", rank=rank, world_size=world)
                import torch
                ok = torch.tensor([1 if "t" in box else 0], dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                transport = box["t"]
            else:
                why = box.get("err", "timeout" if th.is_alive() else "another rank failed")
                if os.environ.get("POCR_BENCH_REQUIRE_RCCL") == "1":
                    raise SystemExit(f"[bench rank {rank}] POCR_BENCH_REQUIRE_RCCL=1 and the RCCL communicator is not available "
                                     f"on every rank (this rank: {why}): refusing the gloo fallback")
                collective = f"gloo FALLBACK (RCCL communicator not available on every rank; this rank: {why})"
                print(f"[bench rank {rank}] {collective}", file=sys.stderr)
                transport = sharding.TorchDistTransport()

    # self-certification of the exchange: the number of ranks the RCCL communicator ITSELF reports (ncclCommCount through
    # pocr_comm_info), 0 when the exchange does not run over RCCL (single GPU without a communicator, or the gloo fallback)
    rccl_fields = {"rccl_ranks": eng.comm_info()[0] if isinstance(transport, sharding.RcclTransport) else 0}
    if isinstance(transport, sharding.RcclTransport) and rccl_fields["rccl_ranks"] != world:
        raise SystemExit(f"[bench rank {rank}] the RCCL communicator reports {rccl_fields['rccl_ranks']} ranks, the launcher {world}")

    def fence():
        eng.device_synchronize()
        if transport is not None:
            transport.barrier()
        eng.device_synchronize()

    stage_sum = {}
    extra = {}

    if args.workload == "c5":
        # ------------------------------------------------------------------ c5: page -> layout maps -> crops -> text
        from pero_ocr_amd import parsenet_spec
        from pero_ocr_amd.document_ocr.page_ocr import LineCropper, PageOCR
        from pero_ocr_amd.layout_engines import torch_parsenet

        class Line:
            def __init__(self, i, baseline, heights):
                self.id, self.baseline, self.heights = f"r0-l{i}", np.array(baseline), heights
                self.crop = self.transcription = self.logits = self.characters = self.logit_coords = None
                self.transcription_confidence = None

        class Layout:
            def __init__(self, lines):
                self.lines = lines

            def lines_iterator(self):
                return iter(self.lines)

        ph, pw = wl["page_h"], wl["page_w"]
        n_pages = 4                                  # distinct pages, cycled
        pages = [synth.make_page(900 + 10 * rank + k, ph, pw) for k in range(n_pages)]
        boxes = [synth.page_line_boxes(900 + 10 * rank + k, ph, pw) for k in range(n_pages)]
        pn_path = os.path.join(tmp.name, "parsenet.pocrp")
        torch_parsenet.save_blob(pn_path, parsenet_spec.generate_weights(20261001))
        parsenet = torch_parsenet.TorchParseNet(pn_path, Dev(local_rank), downsample=4, adaptive_downsample=False)
        cropper = LineCropper({"LINE_HEIGHT": str(spec.height), "INTERP": "2", "LINE_SCALE": "1.0",
                               "RESIDENT_CROPS": "no" if os.environ.get("POCR_BENCH_HOST_CROPS") == "1" else "yes"}, device_id=local_rank)
        page_ocr = PageOCR({"OCR_JSON": os.path.join(tmp.name, "ocr.json")}, Dev(local_rank))
        stage = {"layout_net": 0.0, "crop": 0.0, "ocr": 0.0}
        lines_done = 0

        def one_page(k):
            nonlocal lines_done
            t_a = time.perf_counter()
            maps, ds = parsenet.get_maps_with_optimal_resolution(pages[k])
            assert maps.shape == (ph // 4, pw // 4, 5) and ds == 4
            # layout post-processing stub: the generator's line boxes as baselines (3 points, 30 px above / 10 px below)
            layout = Layout([Line(i, [[x0, y0 + 30], [x0 + wd // 2, y0 + 30], [x0 + wd, y0 + 30]], [30, 10])
                             for i, (x0, y0, wd) in enumerate(boxes[k])])
            t_b = time.perf_counter()
            cropper.process_page(pages[k], layout)
            t_c = time.perf_counter()
            page_ocr.process_page(pages[k], layout)
            t_d = time.perf_counter()
            stage["layout_net"] += t_b - t_a; stage["crop"] += t_c - t_b; stage["ocr"] += t_d - t_c
            lines_done += len(layout.lines)
            return [ln.transcription for ln in layout.lines]

        from pero_ocr_amd.document_ocr.page_stream import PageStream

        def layout_front(img):
            k = page_index[id(img)]
            maps, ds = parsenet.get_maps_with_optimal_resolution(img)
            assert maps.shape == (ph // 4, pw // 4, 5) and ds == 4
            return Layout([Line(i, [[x0, y0 + 30], [x0 + wd // 2, y0 + 30], [x0 + wd, y0 + 30]], [30, 10])
                           for i, (x0, y0, wd) in enumerate(boxes[k])])

        page_index = {id(pg): k for k, pg in enumerate(pages)}
        ppb = args.pages_per_batch
        # further front pairs (layout network + cropper instances of their own: a pair owns device buffers), one worker thread each
        n_fronts = max(1, args.front_workers)
        extra_fronts = []
        for _w in range(1, n_fronts):
            pn_w = torch_parsenet.TorchParseNet(pn_path, Dev(local_rank), downsample=4, adaptive_downsample=False)
            cr_w = LineCropper({"LINE_HEIGHT": str(spec.height), "INTERP": "2", "LINE_SCALE": "1.0",
                                "RESIDENT_CROPS": "no" if os.environ.get("POCR_BENCH_HOST_CROPS") == "1" else "yes"}, device_id=local_rank)

            def front_w(img, pn_w=pn_w):
                k = page_index[id(img)]
                maps, ds = pn_w.get_maps_with_optimal_resolution(img)
                assert maps.shape == (ph // 4, pw // 4, 5) and ds == 4
                return Layout([Line(i, [[x0, y0 + 30], [x0 + wd // 2, y0 + 30], [x0 + wd, y0 + 30]], [30, 10])
                               for i, (x0, y0, wd) in enumerate(boxes[k])])
            extra_fronts.append((front_w, cr_w))
        stream = PageStream(layout_front, cropper, page_ocr, pages_per_batch=ppb, extra_fronts=extra_fronts)

        def run_stream(n):
            nonlocal lines_done
            texts = None
            for _img, layout in stream.process(pages[i % n_pages] for i in range(n)):
                lines_done += len(layout.lines)
                texts = [ln.transcription for ln in layout.lines]
                nnz_frames[0] += sum(ln.logits.nnz for ln in layout.lines)
                nnz_frames[1] += sum(ln.logits.shape[0] for ln in layout.lines)
            return texts

        nnz_frames = [0, 0]

        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # the engine prints the reference's "Line too long" warnings
            # (1) one page at a time, stage after stage: what a single PageParser.process_page call costs (latency)
            for i in range(max(1, args.warmup)):
                one_page(i % n_pages)
            for k in stage:
                stage[k] = 0.0
            fence()
            t0 = time.perf_counter()
            n_lat = min(args.steps, 8)
            for i in range(n_lat):
                one_page(i % n_pages)
            fence()
            lat = (time.perf_counter() - t0) / n_lat
            extra["page_at_a_time"] = {"pages_per_s": round(1.0 / lat, 2), "ms_per_page": round(1e3 * lat, 3),
                                        "stage_ms_per_page": {k: round(1e3 * v / n_lat, 3) for k, v in stage.items()}}
            # (2) the page stream (the figure `value` reports): front of the next pages on a helper thread, the
            # recogniser fed with the lines of `ppb` pages per call
            run_stream(max(ppb, args.warmup))
            lines_done = 0
            for k in stream.stats:
                stream.stats[k] = 0
            fence()
            t0 = time.perf_counter()
            texts = run_stream(args.steps)
            fence()
            elapsed = time.perf_counter() - t0
            # where the consumer thread's time went: waiting for the front (layout network + cropper of the NEXT pages, helper
            # thread) or inside the recogniser's calls
            extra["stream_consumer_ms_per_page"] = {"waiting_for_front": round(1e3 * stream.stats["wait_front_s"] / args.steps, 3),
                                                    "recogniser_calls": round(1e3 * stream.stats["ocr_s"] / args.steps, 3),
                                                    "batches_in_flight_overlap": bool(stream.overlap_batches), "front_workers": n_fronts}
        assert all(isinstance(t, str) for t in texts)
        # the stream's result is CHECKED: the last batch of `ppb` pages again, sequentially (same lines in one process_lines call ->
        # same reference chunk plan -> the deterministic GPU path must reproduce every string and coordinate); and against one
        # page at a time (another chunk plan: a line's padded width, hence its last bits, may differ - reported, bounded)
        with contextlib.redirect_stdout(sys.stderr):
            first = ((args.steps - 1) // ppb) * ppb
            idx = [i % n_pages for i in range(first, args.steps)]
            streamed = [lay for _img, lay in stream.process(pages[k] for k in idx)]
            again = []
            for k in idx:
                lay = layout_front(pages[k]); cropper.process_page(pages[k], lay); again.append(lay)
            page_ocr.process_pages(again)
            for a_, b_ in zip(streamed, again):
                assert [l.transcription for l in a_.lines] == [l.transcription for l in b_.lines], "c5: the stream and the sequential batch disagree"
                assert [l.logit_coords for l in a_.lines] == [l.logit_coords for l in b_.lines]
            assert texts == [l.transcription for l in streamed[-1].lines]
            single = one_page(idx[-1])
            # the stream with ONE page per batch has one page's chunk plan: it must reproduce the page-at-a-time strings exactly
            lone = [lay for _img, lay in PageStream(layout_front, cropper, page_ocr, pages_per_batch=1).process(iter([pages[idx[-1]]]))]
            assert [l.transcription for l in lone[0].lines] == single, "c5: a one-page stream and PageOCR.process_page disagree"
            # with `ppb` pages per batch a line may sit in a chunk of another padded width than in its own page's plan (the plan is a
            # function of ALL widths handed to process_lines, exactly as in the reference): its logits then differ in the last bits
            # and a near-tie frame may decode differently - counted and reported, not an error
            differing = sum(x != y for x, y in zip(single, texts))
        extra["checked"] = {"stream_equals_sequential_batch": True, "one_page_stream_equals_process_page": True,
                            "lines_differing_between_batched_stream_and_page_at_a_time": differing, "lines": len(texts)}
        lines_per_step = 1 * world                  # unit of this workload: pages
        scaling = "weak"
        extra["unit_note"] = "value is PAGES/s for this workload"
        extra["lines_per_page"] = lines_done / args.steps
        extra["lines_per_s"] = round(lines_done * world / elapsed, 1)
        extra["layout_net_gpu_ms"] = round(parsenet.net.last_ms(), 3)
        extra["sparse_logits_nnz_per_frame"] = round(nnz_frames[0] / max(1, nnz_frames[1]), 2)
        extra["head_temperature"] = args.head_temperature
        workload_txt = (f"c5: stream of {ph}x{pw} synthetic pages, one per step per GPU: layout network (parsenet_unet64, downsample 4 -> {ph // 4}x{pw // 4}) -> "
                        f"layout post-processing stub (ground-truth baselines of the {len(boxes[0])} pasted lines) -> resident GPU line cropper (crops stay in HBM) -> "
                        f"VGG+BiLSTM+CTC line OCR (default batch_size 8, sparse logits + confidences) -> strings; inputs: host page per step; "
                        f"layout + crop of the next pages on {n_fronts} helper thread(s) (a layout-network + cropper instance each), the recogniser gets the lines of {ppb} pages per process_lines call"
                        + (", the next call's launches enqueued before a call's last ones are collected" if stream.overlap_batches else ""))
        w_pad = None
    elif args.workload == "c3":
        # ------------------------------------------------------------------ c3: sharded page stream (strong scaling)
        widths = meta["widths"]
        lines = synth.make_crops(meta["crop_seed"], widths, spec.height, meta.get("crop_indices"))
        n_total = len(lines)
        sh = sharding.ShardedLineOCR(sharding.engine_recogniser(engine), engine.characters, engine.max_input_horizontal_pixels,
                                     transport=transport if transport is not None else sharding.LocalTransport())
        texts = None
        for _ in range(args.warmup):
            texts = sh.process_lines(lines, no_logits=True)[0]
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            texts = sh.process_lines(lines, no_logits=True)[0]
        fence()
        elapsed = time.perf_counter() - t0
        assert texts == meta["transcriptions"], "c3: transcriptions differ from the reference fixture"
        lines_per_step = n_total
        scaling = "strong"
        workload_txt = (f"c3: {n_total}-line page stream, widths 128..1024 (seeded), reference plan at default batch_size 8 "
                        f"({len(meta['plan'])} chunks), whole chunks dealt to {world} rank(s) (LPT), inputs: host->device every "
                        "step, " + ("one RCCL all-gather of the labels per pass" if rccl_fields["rccl_ranks"] >= 1 else
                                    ("one all-gather of the labels per pass over the gloo carrier (NOT RCCL)" if transport is not None else
                                     "LocalTransport: a world of one, NO collective")) +
                        "; transcriptions checked against the reference fixture")
        w_pad = None
    else:
        # ------------------------------------------------------------------ c2 / c4: one reference chunk per step
        n_lines, width = wl["n_lines"], wl["width"]
        crops = synth.make_crops(wl["crop_seed"] + 1000 * rank, [width] * n_lines, spec.height)
        w_pad = width + 64
        T = (w_pad // 2) // 2
        pool = np.concatenate([c.reshape(-1) for c in crops])
        offsets = np.arange(n_lines, dtype=np.int64) * (spec.height * width * 3)
        for sl in range(n_slots):                   # inputs resident in HBM (one staged copy per pipeline slot)
            eng.slot_stage_lines(sl, pool, offsets, np.full(n_lines, width, np.int32), w_pad, 32)
        eng.slot_launch(0, want_logits=False)       # also waits for the uploads
        eng.slot_collect(0)
        line_ids = np.arange(n_lines, dtype=np.int32) + rank * n_lines
        m_of = [n_lines] * world

        def to_strings(labels, lens):
            if transport is not None:               # ONE fixed-stride all-gather: rows [line id, length, labels...]
                rows = np.full((n_lines, T + 2), -1, dtype=np.int32)
                rows[:, 0], rows[:, 1], rows[:, 2:2 + labels.shape[1]] = line_ids, lens, labels
                got = sharding.allgather_rows(transport, rows, n_lines, m_of)
                labels, lens = got[:, 2:], got[:, 1]
            return labels_to_strings(labels, lens, chars)

        # Steps are software-pipelined over the engine's two slots: step i is enqueued (conv backbone -> sequence
        # model -> head -> CTC -> async D2H) before step i-1 is collected and decoded to strings, so the latency-bound
        # tail and the host work of one step overlap the MFMA-bound convs of the next.  Every step's work,
        # including its string decode and all-gather, completes inside the timed region.
        def finish(slot):
            _lg, _am, labels, lens = eng.slot_collect(slot)
            for k, v in eng.slot_stage_ms(slot).items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
            return to_strings(labels, lens)

        def run_resident(k_steps):
            for i in range(k_steps):
                eng.slot_launch(i % n_slots, want_logits=False)
                if i > 0:
                    finish((i - 1) % n_slots)
            return finish((k_steps - 1) % n_slots)

        chunk = Chunk(list(range(n_lines)), -(-width // 32) * 32, w_pad)

        def run_end_to_end(k_steps):
            """Per step: pack the crops from host memory, H2D, launch, collect, strings (process_lines' launch loop)."""
            pending, texts = None, None
            for i in range(k_steps):
                handle = engine._submit_launch(crops, Launch([chunk]), False, i % n_slots)
                if pending is not None:
                    t_, _l = engine._collect_launch(pending)[:2]
                    texts = t_
                pending = handle
            return engine._collect_launch(pending)[0]

        if args.warmup:
            run_resident(args.warmup)
        eng.set_profiling(True)
        stage_sum.clear()
        fence()
        t0 = time.perf_counter()
        texts = run_resident(args.steps)
        fence()
        elapsed = time.perf_counter() - t0
        assert len(texts) == n_lines * world
        ms = {k: v / args.steps for k, v in stage_sum.items()}
        eng.set_profiling(False)
        # second timed region: host crops -> strings per step (PCIe-inclusive); N = 1 only reports it
        run_end_to_end(max(1, args.warmup))
        fence()
        t1 = time.perf_counter()
        texts_e2e = run_end_to_end(args.steps)
        fence()
        e2e = time.perf_counter() - t1
        assert texts_e2e == texts[rank * n_lines:(rank + 1) * n_lines] if transport is None else len(texts_e2e) == n_lines
        if transport is not None:
            e2e = transport.allreduce_max(e2e)
        extra["end_to_end"] = {"value": round(n_lines * world * args.steps / e2e, 1), "unit": "lines/s",
                               "ms_per_step": round(1e3 * e2e / args.steps, 3),
                               "what": "every step packs its crops from host memory (numpy), H2D, launch, collect, strings - "
                                       "PytorchEngineLineOCR's launch loop; no all-gather in this region"}
        # third region (not part of any rate): the same launch ALONE - one chunk at a time, nothing else on the GPU - for the
        # per-stage times the kernels have without the other chunk's sequence stage next to them
        ms_alone = {}
        eng.set_profiling(True)
        for _ in range(3):
            eng.slot_launch(0, want_logits=False)
            eng.slot_collect(0)
            for k, v in eng.slot_stage_ms(0).items():
                ms_alone[k] = ms_alone.get(k, 0.0) + v / 3.0
        eng.set_profiling(False)
        if args.workload == "c2" and world == 1 and not args.no_extras:
            extra["c2_sparse"] = default_contract_region(engine, eng, crops, n_lines, width, wl, spec, weights, chars, tmp, local_rank)
        lines_per_step = n_lines * world
        scaling = "weak"
        seq = "BiLSTM(2x256)" if spec.arch == netspec.ARCH else f"self-attention encoder ({spec.sa_layers}x{spec.sa_heads} heads, ff {spec.sa_ff})"
        workload_txt = (f"{args.workload}: {n_lines} lines @40x{width} per GPU, one reference chunk (batch_size {wl['batch_size']}, "
                        f"W_pad {w_pad}, T {T}), VGG+{seq}+CTC, C={spec.num_classes}, seeded synthetic weights; "
                        "value: inputs resident in HBM, strings on the host; end_to_end: inputs host->device every step")

    if transport is not None:
        elapsed = transport.allreduce_max(elapsed)

    if rank == 0:
        split = _native.conv_split()
        peak = conv_peak_tflops(split)
        dtype_txt = {2: "f32 (convs / GEMMs: fp32 operands as two f16 planes h + l/2048, three f16 MFMAs per 32-deep block, fp32 accumulate)",
                     3: "f32 (convs / GEMMs: fp32 operands as exact sums of three bf16, six bf16 MFMAs per 32-deep block, fp32 accumulate)",
                     0: "f32"}[split]
        result = {
            "metric": {"c4": "text-line crops/s (CTC-decoded) at 40x768", "c5": "pages/s end to end (4k x 3k page: layout net + crop + line OCR)"}.get(
                args.workload, "text-line crops/s (CTC-decoded) at 40x512"),
            "value": round(lines_per_step * args.steps / elapsed, 2 if args.workload == "c5" else 1),
            "unit": "pages/s" if args.workload == "c5" else "lines/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtype_txt, "data": "synthetic",
            "config": {"workload": workload_txt, "lines_per_step": lines_per_step,
                       "parallelism": f"chunk-sharded x{world}, one RCCL all-gather of labels per step (C ABI)"
                                      if transport is not None else "single GPU, no collective",
                       "collective": collective, "conv_arithmetic": SPLIT_NAME[split],
                       "pipelining": (f"{n_slots} launches in flight per GPU (separate HIP streams) in this step loop - its measured optimum for one uniform "
                                      "chunk per step; process_lines (extra.c2_sparse / c3 / c5) runs at the product's default of three")
                                     if n_slots == 2 and args.workload in ("c2", "c4") else f"{n_slots} launches in flight per GPU (separate HIP streams)"},
        }
        result.update(shape_rccl_fields(result, rccl_fields, collective))
        if w_pad is not None:
            traffic = None          # HBM bytes per launch of the dominant kernel, from the committed PMC passes
            pmc_meta = {}
            pmc_file = {2: "r06_pmc_summary.json", 3: "r02_pmc_summary.json", 0: "pmc_summary.json"}[split]
            for older in ("r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json"):
                if not os.path.exists(os.path.join(REPO, "profiles", pmc_file)):
                    pmc_file = older
            rows_on = os.environ.get("POCR_CONV_ROWS", "1") != "0"     # (csrc/conv_rows.hpp runs conv9 in the default mode)
            dom_sig = {2: "rows_kernel<5, 1, 2, 1, 1, 1, 2, true, 2, false>" if rows_on else "bf16x3_kernel<5, 1, 2, 1, 1, 1, 2, true, 2, true, 3, 3, 1, 1, false, 2, true, true",
                       3: "bf16x3_kernel<5, 1, 2, 1, 1, 1, 2, true, 2, true, 3, 3, 1, 1, false>", 0: "5, 1, 4, 4, 16, 1, 1, 2, true"}[split]
            try:
                pmc = json.load(open(os.path.join(REPO, "profiles", pmc_file)))
                pmc_meta = pmc.pop("_meta", {})
                for kname, ctr in pmc.items():
                    if dom_sig in kname and "hbm_bytes_per_launch" in ctr and args.workload == "c2":
                        traffic = ctr["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
            fl = conv_flops_per_line(w_pad)
            dom = "conv9"                                    # 24 % of the conv FLOPs, the largest single kernel
            dom_tf = fl[dom] * n_lines / (ms[dom] * 1e-3) / 1e12
            # Since round 4 the backbone of launch k+1 starts behind conv5 of launch k (profiles/r04_backbone_overlap.txt): two backbones
            # share the chip for half a step, so a stage's event-to-event time contains the other launch's kernels and the stage times
            # no longer add up to a step (their sum is reported as `sum_of_stage_events_ms`).  The backbone's rate is therefore its FLOPs
            # per step over the step's WALL time - a lower bound: the step also contains whatever the sequence stage is not hidden behind.
            conv_ms = sum(ms[k] for k in fl)
            step_ms = 1e3 * elapsed / args.steps
            conv_tf = sum(fl.values()) * n_lines / (step_ms * 1e-3) / 1e12
            rows_kernel = split == 2 and os.environ.get("POCR_CONV_ROWS", "1") != "0"          # (csrc/conv_rows.hpp: the default f16x2 / P2 mode)
            kname = ("conv3x3_rows_kernel<TH5,MW1,NS2,leaky+BN,f16x2> (csrc/conv_rows.hpp)" if rows_kernel else
                     f"conv3x3_bf16x3_kernel<TH5,MW1,NS2,leaky+BN,{SPLIT_NAME[split]}>" if split else "conv_igemm_kernel<3x3,TH5,NT256,leaky+BN>")
            nm = MFMA_PER_BLOCK[split]
            result["roofline"] = {
                "bound": "mfma", "kernel": f"{kname} ({dom}, 512->512 @5x{w_pad // 4})",
                "achieved": round(dom_tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(dom_tf / peak, 4), "traffic": traffic,
                "traffic_note": "HBM bytes/launch = 2*FETCH_SIZE + WRITE_SIZE (KiB -> B; gfx950 FETCH_SIZE x2 correction) from "
                                f"separate rocprofv3 --pmc passes of this bench, profiles/{pmc_file}",
                "traffic_source_head": pmc_meta.get("head", "") if traffic is not None else None,
                "traffic_library_source_hash": pmc_meta.get("library_source_hash", "") if traffic is not None else None,
                "flops_per_launch": fl[dom] * n_lines, "avg_launch_ms": round(ms[dom], 4),
                "concurrency": ("in the timed region a launch of this kernel shares the chip with conv2..conv5 of the NEXT launch's backbone (backbones overlap "
                                "from conv5 on, profiles/r04_backbone_overlap.txt) and with the previous launch's sequence stage: its duration is a lower "
                                "bound of the kernel's own rate; `alone` = one chunk at a time, nothing else on the GPU") if n_slots > 1 else "one launch in flight",
                "peak_dtype": {2: "algorithmic fp32 FLOPs on the f16 MFMA pipe: operands as two f16 planes, 3 v_mfma_f32_16x16x32_f16 per 32-deep "
                                  "block -> ceiling = 2500 TFLOP/s dense f16 / 3; executed MFMA rate = 3 x achieved",
                               3: "algorithmic fp32 FLOPs on the bf16 MFMA pipe: exact 3-way bf16 split, 6 v_mfma_f32_16x16x32_bf16 per "
                                  "32-deep block -> ceiling = 2500 TFLOP/s dense bf16 / 6; executed MFMA rate = 6 x achieved",
                               0: "fp32 MFMA (v_mfma_f32_16x16x4_f32), dense"}[split],
                "mfma_pipe_frac": round(nm * dom_tf / (BF16_MFMA_PEAK_TFLOPS if split else F32_MFMA_PEAK_TFLOPS), 4),
                "vs_bf16x3_ceiling_416.7": round(dom_tf / (BF16_MFMA_PEAK_TFLOPS / 6.0), 4),
                "vs_fp32_mfma_peak_157.3": round(dom_tf / F32_MFMA_PEAK_TFLOPS, 4)}
            if split == 2:
                kt = kernel_trace_summary(args.workload, fl, n_lines, peak)
                if kt:
                    result["roofline"]["kernel_trace"] = kt
            if split:
                # measured on this part, register-only loops of independent MFMAs, 2 waves per SIMD, at the package power cap:
                # v_mfma_f32_16x16x32_f16 sustains 2432 TFLOP/s on zero operands and 2027 on random ones (tools/mfma_shape_probe.hip,
                # profiles/r04_mfma_shape_probe.txt); v_mfma_f32_16x16x32_bf16 2300 / 1880 (tools/mfma_bf16_peak.hip,
                # profiles/r02_mfma_bf16_sustained.txt) - constants from those committed files, not measured in this run
                probe = {2: (2027.0, 2432.0, "profiles/r04_mfma_shape_probe.txt"), 3: (1880.0, 2300.0, "profiles/r02_mfma_bf16_sustained.txt")}[split]
                result["roofline"]["sustained_mfma_probe"] = {"random_operands_tflops": probe[0], "zero_operands_tflops": probe[1], "source": probe[2],
                                                              "frac_of_sustained_random": round(nm * dom_tf / probe[0], 4)}
            result["conv_backbone"] = {"achieved": round(conv_tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                       "frac": round(conv_tf / peak, 4),
                                       "vs_bf16x3_ceiling_416.7": round(conv_tf / (BF16_MFMA_PEAK_TFLOPS / 6.0), 4),
                                       "vs_fp32_mfma_peak_157.3": round(conv_tf / F32_MFMA_PEAK_TFLOPS, 4),
                                       "gflop_per_line": round(sum(fl.values()) / 1e9, 3), "ms_per_step": round(step_ms, 3),
                                       "definition": "conv FLOPs of a step / the step's wall time (overlapping backbones: stage events are not additive)",
                                       "sum_of_stage_events_ms": round(conv_ms, 3),
                                       "note": ("conv1 (uint8 crops -> 64 channels, K = 27: one f16x2 product block) is computed inside conv2's prologue for the "
                                                "workgroup's own halo tile: its FLOPs (counted once, not per halo overlap) and its time are conv2's" if split == 2 else
                                                "conv1 (K = 27) stays on its fused uint8 -> fp32-MFMA kernel" if split else "")}
            if spec.arch == netspec.ARCH_SA:
                E, FF, Tn = spec.conv_out, spec.sa_ff, (w_pad // 2) // 2
                enc_fl = spec.sa_layers * (2.0 * Tn * (4 * E * E + 2 * E * FF) + 4.0 * Tn * Tn * E) + 2.0 * Tn * E * spec.num_classes
                result["encoder"] = {"gflop_per_line": round(enc_fl / 1e9, 3), "ms_per_step": round(ms["lstm"] + ms["head"], 3),
                                     "achieved": round(enc_fl * n_lines / ((ms["lstm"] + ms["head"]) * 1e-3) / 1e12, 2),
                                     "peak": round(peak, 1), "unit": "TFLOP/s",
                                     "what": "LayerNorm+PE, encoder layers (QKV / attention / out / FFN) and the head, stage events; the linears run on the "
                                             f"{SPLIT_NAME[split]} kernel in GEMM mode (attention, LayerNorm, head: fp32)" if split else
                                             "LayerNorm+PE, encoder layers (QKV / attention / out / FFN) and the head, stage events"}
            result["stage_ms"] = {k: round(v, 4) for k, v in ms.items()}
            result["stage_ms_note"] = ("event-to-event times on each launch's own streams with several launches in flight: a stage's time contains the kernels of the "
                                       "other launch it shares the chip with, the stages do not add up to ms_per_step; stage_ms_alone: one chunk at a time")
            if ms_alone:
                # the same kernels with the GPU to themselves (one chunk at a time): what the two-chunk overlap costs each stage
                conv_ms_a = sum(ms_alone[k] for k in fl)
                result["stage_ms_alone"] = {k: round(v, 4) for k, v in ms_alone.items()}
                result["roofline"]["alone"] = {"avg_launch_ms": round(ms_alone[dom], 4),
                                               "achieved": round(fl[dom] * n_lines / (ms_alone[dom] * 1e-3) / 1e12, 2),
                                               "frac": round(fl[dom] * n_lines / (ms_alone[dom] * 1e-3) / 1e12 / peak, 4)}
                result["conv_backbone"]["alone"] = {"ms_per_chunk": round(conv_ms_a, 3),
                                                    "achieved": round(sum(fl.values()) * n_lines / (conv_ms_a * 1e-3) / 1e12, 2),
                                                    "frac": round(sum(fl.values()) * n_lines / (conv_ms_a * 1e-3) / 1e12 / peak, 4)}
                if "encoder" in result:
                    enc_ms_a = ms_alone["lstm"] + ms_alone["head"]
                    result["encoder"]["alone"] = {"ms_per_chunk": round(enc_ms_a, 3),
                                                  "achieved": round(result["encoder"]["gflop_per_line"] * 1e9 * n_lines / (enc_ms_a * 1e-3) / 1e12, 2)}
        result.update(extra)
        if args.workload == "c2" and world == 1 and not args.no_extras:
            eng.device_synchronize()
            c2s = result.pop("c2_sparse")
            # the reference's DEFAULT call next to `value`: `value` times launches that return label ids only
            # (process_lines(..., no_logits=True)); PageOCR.process_page calls process_lines(crops) and gets sparse logits back
            result["default_call"] = {"value": c2s["head_x8"]["value"], "unit": "lines/s", "nnz_per_frame": c2s["head_x8"]["nnz_per_frame"],
                                      "same_call_no_logits": c2s["head_x8"]["same_call_no_logits"]["value"],
                                      "streamed": c2s["head_x8"]["streamed"]["value"],
                                      "flat_head": {"value": c2s["seeded_weights"]["value"], "nnz_per_frame": c2s["seeded_weights"]["nnz_per_frame"],
                                                    "same_call_no_logits": c2s["seeded_weights"]["same_call_no_logits"]["value"],
                                                    "streamed": c2s["seeded_weights"]["streamed"]["value"]},
                                      "what": "process_lines(2048 host crops @40x512) with the reference's default arguments -> strings + scipy CSC logits + "
                                              "logit_coords: one call at a time (median of 3), the same call with no_logits=True (the sparse logits' own cost is "
                                              "the difference), and as a stream of calls (call k + 1 begun before call k is ended).  A lone call pays the launch "
                                              "pipeline's fill and tail (~8 ms of ~77), which `value`'s step loop never does; details: extra.c2_sparse"}
            result["extra"] = dict(c2_sparse=c2s, **run_extra_workloads())
        if world == 1 and not args.no_cpu_baseline and args.workload in ("c2", "c4"):
            try:
                result["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as exc:          # noqa: BLE001 - the GPU line must not be lost to a host-side problem
                result["cpu_baseline"] = {"value": None, "unit": "lines/s", "cores": 0, "kind": "port", "sample": f"failed: {type(exc).__name__}: {exc}"}
        print(json.dumps(result), flush=True)
    if transport is not None:
        transport.barrier()
        eng.comm_destroy()
    tmp.cleanup()


if __name__ == "__main__":
    main()
