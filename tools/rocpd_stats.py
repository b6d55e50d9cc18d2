#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (`*_results.db`) into the per-kernel stats table
(rocprofv3 --kernel-trace --stats): calls, total / average / min / max duration, percentage.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.csv
"""
import sqlite3
import sys


def main(path: str) -> None:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for n, c, t, a, mn, mx in rows:
        print(f"\"{n}\",{c},{t},{a:.1f},{mn},{mx},{100.0 * t / total:.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
