"""Roofline object for the dominant kernel of the sequence-to-sequence decoding loop: the MEMORY
attention, which streams every line's encoder keys and values once per decoder layer and step (HBM-bound).

  algorithmic bytes per launch = lines in the launch x T_mem x 2E x 4 B   (keys | values rows of the encoder output, fp32)
  achieved = bytes / the kernel's average duration in the rocprofv3 kernel trace of the same command
  peak     = 8 TB/s (MI355X_MICROARCH.md: HBM3E peak; ~6.3 TB/s achievable)

usage: python tools/s2s_roofline.py <kernel_stats.txt from tools/rocprof_summary.py> <s2s_bench.json> [lines_per_launch=256]"""
import json
import re
import sys


def main():
    stats, bench = sys.argv[1], json.load(open(sys.argv[2]))
    lines_per_launch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
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
    by = n * t_mem * 2 * E * 4
    ach = by / (avg_us * 1e-6) / 1e9
    out = {"bound": "hbm", "kernel": "dec_attention_kernel<64, memory> (one workgroup per (line, head); keys / values rows fetched 16 B per lane)",
           "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
           "bytes_per_launch": by, "avg_launch_us": avg_us, "launches": calls,
           "note": f"{n} lines per launch (the last launch of a stream holds fewer: the average duration is over all launches), "
                   f"T_mem {t_mem} encoder frames, E {E}: keys + values of every line once per decoder layer and step"}
    print(json.dumps({"roofline": out, "bench": bench}))


if __name__ == "__main__":
    main()
