#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs).

    python tools/pmc_traffic.py gpurun_out/r01g --out profiles/pmc_traffic.json

Reads <dir>/pmc_FETCH_SIZE/pmc_counter_collection.csv, <dir>/pmc_WRITE_SIZE/... and the bench line the PMC run printed
(<dir>/pmc_FETCH_SIZE.json: `roofline.launches_timed` = launches in the timed region, which are the LAST dispatches of
the run).  Corrections (MI355X_MICROARCH.md §HBM, re-checked on this kernel's access pattern with tools/ubench/pmc_calib.hip,
profiles/r01g_pmc_calibration.txt): counters are in KiB; on gfx950 FETCH_SIZE reports half of a 16 B/lane streaming read,
WRITE_SIZE is exact.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import argparse
import csv
import json
import os
import re


def counter(path, rx):
    rows = [r for r in csv.DictReader(open(path)) if re.search(rx, r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--regex", default=r"k_hg_reduce<8, ?true>")
    ap.add_argument("--out", default=None)
    ap.add_argument("--key", default=None, help="entry point of `roofline_other_kernels` instead of the dominant kernel "
                                                "(e.g. snf_hashgrid_bwd_presorted_adam_xp/F2L16)")
    a = ap.parse_args()
    bench = json.loads(open(os.path.join(a.dir, "pmc_FETCH_SIZE.json")).read().strip().splitlines()[-1])
    roof = bench["roofline"]
    n = int(roof["launches_timed"])
    if a.key:
        roof = next(o for o in bench["roofline_other_kernels"] if o["kernel"] == a.key)
    f = counter(os.path.join(a.dir, "pmc_FETCH_SIZE", "pmc_counter_collection.csv"), a.regex)
    w = counter(os.path.join(a.dir, "pmc_WRITE_SIZE", "pmc_counter_collection.csv"), a.regex)
    if a.key:  # (every launch of that instantiation has the step's shape -- warm-up, serial replay, timed steps: average over all)
        n = min(len(f), len(w))
        f, w = f[:n], w[:n]
    else:
        f, w = f[-n:], w[-n:]
    assert len(f) == n and len(w) == n, (len(f), len(w), n)
    read_b, write_b = 2.0 * sum(f) * 1024 / n, sum(w) * 1024 / n
    res = {"workload": bench["config"]["name"], "kernel": roof["kernel"], "launches": n,
           "fetch_size_kib_per_launch": sum(f) / n, "write_size_kib_per_launch": sum(w) / n,
           "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
           "traffic_bytes_per_launch": read_b + write_b,
           "algorithmic_bytes_per_launch": roof["algorithmic_units_per_launch"],
           "traffic_over_algorithmic": (read_b + write_b) / roof["algorithmic_units_per_launch"],
           "corrections": "KiB -> B; FETCH_SIZE x2 (gfx950, 16 B/lane streaming reads); WRITE_SIZE x1",
           "source": os.path.basename(os.path.normpath(a.dir))}
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
