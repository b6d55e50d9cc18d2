#!/usr/bin/env python3
"""Per-kernel HBM bytes per train step from tools/pmc_all.sh (all dispatches of the run are summed per kernel name and divided
by the number of train iterations the run executed: warmup + 3 serial-replay + 8 after the replay + 1 + steps + 1 +
min(steps, 20) fwd/bwd-only).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950 corrections, see tools/pmc_traffic.py)."""
import csv, re, sys, collections
d = sys.argv[1]
iters = float(sys.argv[2]) if len(sys.argv) > 2 else 3 + 3 + 8 + 1 + 6 + 1 + 6
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for r in csv.DictReader(open(f"{d}/pmcall_{c}/pmc_counter_collection.csv")):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("snf::", "")
        tot[name][i] += float(r["Counter_Value"])
        if i == 0:
            tot[name][2] += 1
rows = sorted(((2 * f + w) * 1024 / iters, 2 * f * 1024 / iters, w * 1024 / iters, n / iters, k) for k, (f, w, n) in tot.items())
total = sum(r[0] for r in rows)
print(f"total {total/1e9:.2f} GB per step")
for b, rd, wr, n, k in reversed(rows[-28:]):
    print(f"{k[:60]:60s} {n:5.1f} launches  {b/1e6:9.1f} MB  (read {rd/1e6:8.1f}  write {wr/1e6:8.1f})")
