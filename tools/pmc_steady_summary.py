#!/usr/bin/env python3
"""tools/pmc_steady.sh -> per-kernel HBM bytes of ONE steady-state train step: (counter totals at steps_b) - (totals at steps_a),
divided by steps_b - steps_a.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950 corrections, see tools/pmc_traffic.py)."""
import collections, csv, re, sys
d, sa, sb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])


def totals(steps):
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        for r in csv.DictReader(open(f"{d}/steady_{steps}_{c}/pmc_counter_collection.csv")):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("snf::", "")
            tot[name][i] += float(r["Counter_Value"])
            if i == 0:
                tot[name][2] += 1
    return tot


a, b = totals(sa), totals(sb)
n = float(sb - sa)
rows = []
for k in b:
    f, w, c = (b[k][j] - a.get(k, [0, 0, 0])[j] for j in range(3))
    rows.append(((2 * f + w) * 1024 / n, 2 * f * 1024 / n, w * 1024 / n, c / n, k))
rows.sort()
print(f"total {sum(r[0] for r in rows)/1e9:.2f} GB per steady-state step  (difference of {sb} and {sa} steps)")
for byt, rd, wr, c, k in reversed(rows[-40:]):
    print(f"{k[:60]:60s} {c:5.2f} launches  {byt/1e6:9.1f} MB  (read {rd/1e6:8.1f}  write {wr/1e6:8.1f})")
