"""Roofline object for the dominant kernel of the sequence-to-sequence decoding loop: the MEMORY
attention, which streams every line's encoder keys and values once per decoder layer and step (HBM-bound).

  algorithmic bytes per launch = lines in the launch x T_mem x 2E x 4 B   (keys | values rows of the encoder output, fp32)
  achieved = bytes / the kernel's average duration in the rocprofv3 kernel trace of the same command
  peak     = 8 TB/s (MI355X_MICROARCH.md: HBM3E peak; ~6.3 TB/s achievable)

usage: python tools/s2s_roofline.py <kernel_stats.txt from tools/rocprof_summary.py> <s2s_bench.json> [lines_per_launch=512]
(the kernel trace is taken with POCR_S2S_DEPTH=1: with several decoding loops side by side a kernel shares the HBM with the others)"""
import json
import re
import sys


def main():
    stats, bench = sys.argv[1], json.load(open(sys.argv[2]))
    lines_per_launch = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    avg_us = calls = None
    for ln in open(stats):
        if "dec_attention_kernel" in ln and re.search(r"dec_attention_kernel<\d+, true>", ln):
            f = ln.split()
            calls, avg_us = int(f[0]), float(f[2])
    if avg_us is None:
        raise SystemExit("no dec_attention_kernel<D, true> row in " + stats)
    E = 512
    w_pad = max(1088, -(-bench["width"] // 32) * 32 + 64)          # the reference centres narrower batches in 1088 columns
    t_mem = w_pad // 4
    n = min(lines_per_launch, bench["lines"])
    by = n * t_mem * 2 * E * 4                                      # a launch with every line still decoding
    if "attention_line_steps_per_pass" in bench:
        # exact average: finished batches are skipped by the kernel, so the bytes of the whole trace are (line, step) pairs x layers x
        # one line's keys + values, and the rate is those bytes over the kernel's total time in the trace
        total_bytes = bench["attention_line_steps_per_pass"] * bench.get("passes", 1) * bench["dec_layers"] * t_mem * 2 * E * 4
        ach = total_bytes / (avg_us * 1e-6 * calls) / 1e9
        how = (f"bytes of the whole trace = {bench['attention_line_steps_per_pass']} (line, step) pairs per pass x {bench.get('passes', 1)} passes x "
               f"{bench['dec_layers']} layers x T_mem {t_mem} x 2E x 4 B (a line is skipped once its reference batch has ended; pairs counted from the "
               f"transcription lengths: batch steps = longest line + 1, +-1 step) over calls x average duration")
    else:
        ach = by / (avg_us * 1e-6) / 1e9
        how = "bytes of a launch with every line alive over the average duration"
    out = {"bound": "hbm", "kernel": "dec_attention_kernel<64, memory> (one workgroup per (line, head); keys / values rows fetched 16 B per lane)",
           "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
           "bytes_per_full_launch": by, "avg_launch_us": avg_us, "launches": calls,
           "note": f"up to {n} lines per launch, T_mem {t_mem} encoder frames, E {E}: keys + values of every line still decoding once per decoder layer "
                   f"and step; {how}; trace taken with one decoding loop at a time (POCR_S2S_DEPTH=1)"}
    print(json.dumps({"roofline": out, "bench": bench}))


if __name__ == "__main__":
    main()
