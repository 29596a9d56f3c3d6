#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd sqlite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`)
into the per-kernel stats table that is committed under profiles/.
Usage: python tools/rocprof_summary.py gpurun_out/prof/r1_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# source: {path}")
    print("# rocprofv3 --kernel-trace --stats ; durations in microseconds")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, total, avg, pct in rows:
        print(f"{calls:>6} {total:>12.1f} {avg:>10.2f} {pct:>6.2f}  {name}")
    try:
        pm = list(cur.execute("select * from counters_collection limit 0"))
    except Exception:
        pm = None


if __name__ == "__main__":
    main(sys.argv[1])
