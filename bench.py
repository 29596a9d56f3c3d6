#!/usr/bin/env python3
"""bench.py — throughput of the batched text-line recognition hot path on MI355X.

Metric (BASELINE.json): text-line crops/s, CTC-decoded, at 40x512.
Workload  (BASELINE.json configs[1]): 256 synthetic 40x512 line crops = ONE reference chunk
(batch_size 274 -> 480*274//512 = 256 lines, W_pad 576, T 144), VGG+BiLSTM+CTC engine,
C = 232 classes, seeded synthetic weights.  One "step" = one pass of the hot path over that
chunk: crops resident in HBM -> staging/normalise -> conv backbone -> BiLSTM -> head -> greedy
CTC -> label ids on the host -> strings.  With N > 1 GPUs every rank runs its own chunk
(weak scaling, chunks are independent units) and the decoded labels are all-gathered over RCCL.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W]
        python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
N_LINES, WIDTH, HEIGHT, N_SYMBOLS = 256, 512, 40, 231
WEIGHT_SEED, CROP_SEED, BATCH_SIZE = 20260929, 305, 274


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


def cpu_baseline(spec, weights, crops, seconds_budget=20.0):
    """The oracle (PyTorch-CPU restatement of the reference path, parity-pinned against the
    imported reference) timed on this box's host cores, on a bounded sample of the workload."""
    import torch
    from oracle import engine_oracle, model_oracle
    net = model_oracle.OracleNet(spec, weights)
    sample = 32
    ids = list(range(sample))
    chars = [""] * spec.num_classes

    def one_pass():
        batch = engine_oracle.assemble_batch(crops, ids, spec.height, WIDTH, 480 * BATCH_SIZE)
        nct = model_oracle.forward_logits(net, batch)
        _best, labels = engine_oracle.greedy_ctc(nct)
        return [engine_oracle.labels_to_text(l, chars) for l in labels]

    one_pass()                                   # warm-up (discarded)
    times = []
    t_end = time.perf_counter() + seconds_budget
    while len(times) < 5 and (time.perf_counter() < t_end or not times):
        t0 = time.perf_counter()
        one_pass()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(sample / med, 2), "unit": "lines/s", "cores": int(torch.get_num_threads()),
            "kind": "port",
            "sample": f"{sample} of the {N_LINES} 40x{WIDTH} crops as one chunk (W_pad 576), median of "
                      f"{len(times)} passes after 1 warm-up, torch {torch.__version__} CPU fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    dist = None
    force_dist = os.environ.get("POCR_FORCE_DIST") == "1"      # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from pero_ocr_amd import _native, netspec, synth, sharding
    from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import labels_to_strings

    chars = synth.make_charset(N_SYMBOLS) + ["\u200b"]
    spec = netspec.NetSpec(num_classes=len(chars), height=HEIGHT)
    weights = netspec.generate_weights(spec, WEIGHT_SEED)
    crops = synth.make_crops(CROP_SEED + 1000 * rank, [WIDTH] * N_LINES, HEIGHT)
    eng = _native.NativeEngine(spec, netspec.pack_weights(spec, weights), local_rank)

    w_pad = WIDTH + 64
    pool = np.concatenate([c.reshape(-1) for c in crops])
    offsets = np.arange(N_LINES, dtype=np.int64) * (HEIGHT * WIDTH * 3)
    n_slots = eng.num_slots
    for sl in range(n_slots):                   # inputs resident in HBM (one staged copy per pipeline slot)
        eng.slot_stage_lines(sl, pool, offsets, np.full(N_LINES, WIDTH, np.int32), w_pad, 32)
    eng.slot_launch(0, want_logits=False)       # also waits for the uploads
    eng.slot_collect(0)
    gather_dev = torch.device("cuda", local_rank) if dist is not None else None
    line_ids = np.arange(N_LINES, dtype=np.int32) + rank * N_LINES

    # Steps are software-pipelined over the engine's two slots: step i is enqueued (conv backbone ->
    # BiLSTM -> head -> CTC -> async D2H) before step i-1 is collected and decoded to strings, so the
    # latency-bound LSTM tail and the host work of one step overlap the MFMA-bound convs of the next.
    # Every step's work, including its string decode and all-gather, completes inside the timed region.
    stage_sum = {}

    def finish(slot):
        _lg, _am, labels, lens = eng.slot_collect(slot)
        for k, v in eng.slot_stage_ms(slot).items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
        if dist is not None:
            labels, lens, _ids = sharding.allgather_labels(labels, lens, line_ids, gather_dev)
        return labels_to_strings(labels, lens, chars)

    def run_steps(k_steps):
        texts = None
        for i in range(k_steps):
            eng.slot_launch(i % n_slots, want_logits=False)
            if i > 0:
                texts = finish((i - 1) % n_slots)
        return finish((k_steps - 1) % n_slots)

    def fence():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    if args.warmup:
        run_steps(args.warmup)
    eng.set_profiling(True)
    stage_sum.clear()
    fence()
    t0 = time.perf_counter()
    texts = run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    assert len(texts) == N_LINES * world
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms = {k: v / args.steps for k, v in stage_sum.items()}
        traffic = None          # HBM bytes per launch of the dominant kernel, from the committed PMC passes
        try:
            pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_summary.json")))
            for kname, ctr in pmc.items():
                if "5, 1, 4, 4, 16, 1, 1, 2, true" in kname and "hbm_bytes_per_launch" in ctr:
                    traffic = ctr["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        fl = conv_flops_per_line(w_pad)
        dom = "conv9"                                    # 24 % of the conv FLOPs, the largest single kernel
        dom_tf = fl[dom] * N_LINES / (ms[dom] * 1e-3) / 1e12
        conv_ms = sum(ms[k] for k in fl)
        conv_tf = sum(fl.values()) * N_LINES / (conv_ms * 1e-3) / 1e12
        result = {
            "metric": "text-line crops/s (CTC-decoded) at 40x512",
            "value": round(N_LINES * world * args.steps / elapsed, 1),
            "unit": "lines/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "c2: 256 lines @40x512 per GPU, one reference chunk (batch_size 274, W_pad 576, "
                                   "T 144), VGG+BiLSTM(2x256)+CTC, C=232, seeded synthetic weights",
                       "lines_per_step_per_gpu": N_LINES, "parallelism": f"chunk-sharded x{world}, RCCL all-gather of labels",
                       "pipelining": f"{n_slots} chunks in flight per GPU (separate HIP streams)"},
            "roofline": {"bound": "mfma", "kernel": f"conv_igemm_kernel<3x3,TH5,NT256,leaky+BN> ({dom}, 512->512 @5x144)",
                         "achieved": round(dom_tf, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(dom_tf / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_note": "HBM bytes/launch = 2*FETCH_SIZE + WRITE_SIZE (KiB -> B; gfx950 FETCH_SIZE x2 correction) from "
                                         "separate rocprofv3 --pmc passes of this bench, profiles/pmc_summary.json",
                         "flops_per_launch": fl[dom] * N_LINES, "avg_launch_ms": round(ms[dom], 4),
                         "peak_dtype": "fp32 MFMA (v_mfma_f32_16x16x4_f32), dense"},
            "conv_backbone": {"achieved": round(conv_tf, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(conv_tf / F32_MFMA_PEAK_TFLOPS, 4), "gflop_per_line": round(sum(fl.values()) / 1e9, 3),
                              "ms_per_step": round(conv_ms, 3)},
            "stage_ms": {k: round(v, 4) for k, v in ms.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(spec, weights, crops)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
