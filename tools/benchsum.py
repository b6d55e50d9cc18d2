import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"],3), "value %.3e"%d["value"], d["roofline"])
for o in d.get("roofline_other_kernels", []): print("   other:", o["kernel"], o["bound"], o["achieved"], o["unit"], "frac", o["frac"])
for k,v in list(d["kernel_ms_per_step"].items())[:18]: print("  ",k,v)
if d.get("cpu_baseline"): print(d["cpu_baseline"])
