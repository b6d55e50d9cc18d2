#!/usr/bin/env python3
"""Per-kernel sums of every counter found under <dir>/pmc_*/ (rocprofv3 --pmc ... --output-format csv: *_counter_collection.csv),
divided by the number of dispatches: one line per (kernel, counter).  usage: python tools/pmc_table.py <dir>"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(lambda: defaultdict(set))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
for k in sorted(acc):
    for c in sorted(acc[k]):
        d = max(len(n[k][c]), 1)
        print(f"{k:48s} {c:36s} {acc[k][c] / d:16.0f} per dispatch  ({d} dispatches)")
