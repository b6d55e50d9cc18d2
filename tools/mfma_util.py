#!/usr/bin/env python3
"""Matrix-core busy fraction per kernel from tools/mfma_util.sh (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... + kernel trace).

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration in shader cycles x 1024 SIMDs)
            = the average fraction of the chip's matrix pipes that were busy while the kernel ran
(SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD: 32 per v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32;
duration from the kernel trace at the nominal 2.4 GHz -- the clock under a profiler is lower, so this is a lower bound.)
Writes profiles/r<NN>_mfma_util.json (bench.py reads the newest for `roofline_other_kernels[*].mfma_busy`)."""
import collections, csv, glob, json, os, re, sys

# usage: mfma_util.py <out.json> <dir> [<dir> ...]   (directories of counter passes: train step, render, encoder)
#    or: mfma_util.py <dir | summary.json> [<out.json>]   (round-2 form)
args = sys.argv[1:]
if args and args[0].endswith(".json") and len(args) > 1 and os.path.isdir(args[1]):
    out_path, dirs = args[0], args[1:]
else:
    dirs = [args[0]]
    out_path = args[1] if len(args) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                           "profiles", "r02_mfma_util.json")
d = dirs[0]
CLK_GHZ, SIMDS = 2.4, 1024
by_source = {}
if d.endswith(".json"):  # an earlier summary of this script (the raw CSVs are not kept): only the entry-point table is rebuilt
    by_kernel = json.load(open(d))["by_kernel"]
else:
    # one table per counter pass (train step / render / encoder): the chains and GEMMs of the render pass share kernel names with the
    # train step's but run on 8 x the samples per launch -- their busy fractions must not be averaged together
    for d in dirs:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
        dur = {}
        for r in csv.DictReader(open(kt)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for r in csv.DictReader(open(cc)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("snf::", "")
            a = agg[name]
            a[r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_BUSY_CYCLES":
                a["calls"] += 1
                a["ns"] += dur.get(r["Dispatch_Id"], 0)
        tab = {}
        for name, a in agg.items():
            if a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or a["ns"] <= 0:
                continue
            tab[name] = {"calls": int(a["calls"]), "avg_us": round(a["ns"] / a["calls"] / 1e3, 1),
                         "mfma_busy": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["ns"] * CLK_GHZ * SIMDS), 4),
                         "mfma_mops_f32": a.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0),
                         "mfma_mops_bf16": a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)}
        by_source[os.path.basename(os.path.normpath(d))] = dict(sorted(tab.items(), key=lambda kv: -kv[1]["mfma_busy"]))
    by_kernel = by_source.get("step") or next(iter(by_source.values()))  # the train step's table feeds the step's entry points


def find_in(src, sub):
    v = [x["mfma_busy"] for k, x in by_source.get(src, {}).items() if sub in k]
    return max(v) if v else None


def find(sub):
    v = [x["mfma_busy"] for k, x in by_kernel.items() if sub in k]
    return max(v) if v else None


# C-ABI entry points of bench.py's kernel table -> the kernel that does their matrix work
entry = {
    "snf_mlp64_fwd/31x64x64x3": find("k_mlp_chain_fwd_b3<2,") or find("k_mlp_chain_fwd<2"),
    "snf_mlp64_fwd/32x64x16": find("k_mlp_chain_fwd_b3<1,") or find("k_mlp_chain_fwd<1"),
    "snf_mlp64_fwd_density/32x64x16": find("k_mlp_chain_fwd_b3<1,") or find("k_mlp_chain_fwd<1"),
    "snf_mlp64_bwd_data/31x64x64x3": find("k_mlp_chain_bwd<2"), "snf_mlp64_bwd_data/32x64x16": find("k_mlp_chain_bwd<1"),
    "snf_mlp64_bwd_fused/31x64x64x3": find("k_mlp_chain_bwd_wg<2"), "snf_mlp64_bwd_fused/32x64x16": find("k_mlp_chain_bwd_wg<1"),
    "snf_linear_fwd/192x256": find("k_gemm_ws_b3<true, false, 128, 1, 512, 4, true"),
    "snf_linear_fwd_mean/192x256": find("k_gemm_ws_b3<true, false, 128, 1, 512, 4, true"),
    "snf_linear_fwd/256x256": find("k_gemm_ws_b3<true, false, 128, 1, 512, 4, false"),
    "snf_linear_fwd/256x192": find("k_gemm_ws_b3<true, false, 128, 1, 512, 4, false"),
    "snf_linear_fwd/256x256r": find("k_gemm_ws_b3<true, false, 128, 1, 512, 4, false"),
    "snf_linear_fwd/256x192r": find("k_gemm_ws_b3<true, false, 128, 1, 512, 4, false"),
    "snf_linear_bwd_data/192x256": find("k_gemm_ws_b3<false, true, 128, 1, 512, 4, false, true"),
    "snf_linear_bwd_data_rows/192x256": find("k_gemm_ws_b3<false, true, 96, 1, 512, 4, false, true")
                                        or find("k_gemm_ws_b3<false, true, 128, 1, 512, 4, false, true"),
    "snf_linear_bwd_data/256x256": find("k_gemm_ws_b3<false, true, 128, 1, 512, 4, false, false"),
    "snf_linear_bwd_data/256x192": find("k_gemm_ws_b3<false, true, 128, 1, 512, 4, false, false"),
    "snf_linear_bwd_data_rows/256x256": find("k_gemm_ws_b3<false, true, 128, 1, 512, 4, false, false"),
    "snf_linear_bwd_data_rows/256x192": find("k_gemm_ws_b3<false, true, 128, 1, 512, 4, false, false"),
    "snf_linear_bwd_data/256x256r": find("k_gemm_ws_b3<false, true, 128, 1, 512, 4, false, false"),
    "snf_linear_bwd_data/256x192r": find("k_gemm_ws_b3<false, true, 128, 1, 512, 4, false, false"),
    "snf_linear_bwd_weight_ws/192x256": find("k_wgrad_full_b3"), "snf_linear_bwd_weight_ws/256x256": find("k_wgrad_full_b3"),
    "snf_linear_bwd_weight_ws/256x192": find("k_wgrad_full_b3"),
    "snf_linear_bwd_weight_rows/192x256": find("k_wgrad_full_b3"), "snf_linear_bwd_weight_rows/256x256": find("k_wgrad_full_b3"),
    "snf_linear_bwd_weight_rows/256x192": find("k_wgrad_full_b3"),
    "snf_linear_bwd_weight/64x64": find("k_gemm_wgrad_b3"), "snf_linear_bwd_weight/2304x256": find("k_gemm_wgrad_b3"),
    "snf_linear_bwd_weight/256x256r": find("k_gemm_wgrad_b3"), "snf_linear_bwd_weight/256x192r": find("k_gemm_wgrad_b3"),
    "snf_linear_bwd_weight/31x64": find("k_gemm_wgrad<true>"), "snf_linear_bwd_weight/32x64": find("k_gemm_wgrad<true>"),
    "snf_linear_bwd_weight/64x16": find("k_gemm_wgrad<true>"), "snf_linear_bwd_weight/64x3": find("k_gemm_wgrad<true>"),
    "snf_linear_fwd_ws/2304x256": find("k_gemm_rows_b3<true, false, 64>"),
    # config #5: the render pass's fused grids -> first head layer, the encoder's token GEMMs and attention
    "snf_grid_head_fused_fwd": find_in("render", "k_grid_head_fused"),
    "snf_gemm_planes": find_in("vit", "k_gemm_planes"),
    "snf_attention_planes": find_in("vit", "k_attention_b3"),
}
res = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_{F32,BF16} SQ_BUSY_CYCLES --kernel-trace -- "
                 "python bench.py --steps 6 --warmup 3 | tools/bench_render.py | tools/bench_vit.py (tools/mfma_util.sh)",
       "definition": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel ns x 2.4 GHz x 1024 SIMDs)",
       "by_kernel": dict(sorted(by_kernel.items(), key=lambda kv: -kv[1]["mfma_busy"])),
       "by_source": by_source,
       "by_entry_point": {k: v for k, v in entry.items() if v is not None}}
json.dump(res, open(out_path, "w"), indent=1)
for src, tab in (by_source or {"": res["by_kernel"]}).items():
    print(f"--- {src}")
    for k, v in tab.items():
        print(f"{k[:90]:90s} calls {v['calls']:4d}  avg {v['avg_us']:8.1f} us  mfma_busy {v['mfma_busy']:.3f}")
print("wrote", out_path)
