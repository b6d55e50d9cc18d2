#!/usr/bin/env python3
"""Short per-kernel table from a rocprofv3 `*_kernel_stats.csv` (names cut to their function name): calls, avg us, total ms, %.
usage: python tools/kstats.py <dir or csv> [top N]"""
import csv, glob, os, re, sys
path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
print(f"# {path}: {len(rows)} kernels, {tot / 1e6:.3f} ms of kernel time")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    name = re.sub(r"^void ", "", r["Name"])
    name = re.sub(r"\(.*$", "", name)[:70]
    print(f"{name:70s} {int(r['Calls']):6d} calls  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.3f} ms  "
          f"{100 * float(r['TotalDurationNs']) / tot:5.1f} %")
