#!/usr/bin/env python3
"""Per (kernel, grid size) table from a rocprofv3 `*_kernel_trace.csv`: one kernel name launched at several sizes (the eval render's
hash-grid forward: proposal and field grids) reads as one average in the stats summary.  usage: ktrace_by_grid.py <dir or csv> [substr] [top N]"""
import csv, glob, os, re, sys
from collections import defaultdict
path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
acc = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r.get("Kernel_Name") or r.get("Name") or ""))[:60]
    if sub and sub not in name:
        continue
    grid = "x".join(str(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in r) or str(r.get("Grid_Size", "?"))
    a = acc[(name, grid)]
    a[0] += 1
    a[1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in acc.values()) or 1.0
print(f"# {path}: {tot / 1e3:.3f} ms in {len(acc)} (kernel, grid) groups")
for (name, grid), (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{name:60s} grid {grid:>18s} {n:5d} calls  avg {us / n:9.1f} us  total {us / 1e3:8.3f} ms  {100 * us / tot:5.1f} %")
