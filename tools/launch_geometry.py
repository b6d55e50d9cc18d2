"""Grid / workgroup / LDS / register footprint of every kernel of the train step from a rocprofv3 kernel trace: how many workgroups a launch
has, how many fit a CU (LDS, registers, 32 waves), and so how much of the 256-CU chip one launch can occupy alone.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python tools/steady_steps.py 12 ; python tools/launch_geometry.py <dir>"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("snf::", "")[:64]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    key = (name, grid // max(wg, 1), wg, int(r["LDS_Block_Size"]), int(r["VGPR_Count"]), int(r.get("Accum_VGPR_Count", 0) or 0))
    d = agg.setdefault(key, [0, 0.0])
    d[0] += 1
    d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"{'kernel':64s} {'WGs':>7s} {'thr':>4s} {'LDS':>7s} {'regs':>5s} {'WG/CU':>5s} {'chip':>5s} {'calls':>5s} {'avg us':>8s}")
for (name, nwg, wg, lds, v, a), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    regs = -(-(v + a) // 8) * 8
    waves = wg // 64
    by_reg = (4 * min(8, 512 // max(regs, 1))) // max(waves, 1)
    by_lds = (160 * 1024) // lds if lds else 99
    per_cu = max(1, min(by_reg, by_lds, 32 // max(waves, 1)))
    print(f"{name:64s} {nwg:7d} {wg:4d} {lds:7d} {regs:5d} {per_cu:5d} {min(1.0, nwg / (256.0 * per_cu)):5.2f} {n:5d} {us / n:8.1f}")
