#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a per-kernel stats table.

usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/kernel_stats_rNN.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name} order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>8} {'total_ms':>12} {'avg_us':>12} {'min_us':>10} {'max_us':>10} {'pct':>7}  kernel")
    for n, c, s, a, mn, mx in rows:
        print(f"{c:8d} {s / 1e6:12.3f} {a / 1e3:12.3f} {mn / 1e3:10.3f} {mx / 1e3:10.3f} {100.0 * s / tot:7.2f}  {n[:150]}")


if __name__ == "__main__":
    main(sys.argv[1])
