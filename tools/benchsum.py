import json,sys
src=open(sys.argv[1]).read() if len(sys.argv)>1 else sys.stdin.read()
d=json.loads(src.strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"],3), "value %.3e"%d["value"], "serial C-ABI kernel sum", d.get("serial_step_ms"))
print("  roofline:", {k:v for k,v in d["roofline"].items() if k!="serial"})
print("  serial  :", d["roofline"].get("serial"))
for o in d.get("roofline_other_kernels", []): print("   other:", o["kernel"], o["bound"], o["achieved"], o["unit"], "frac", o["frac"])
for k,v in list(d["kernel_ms_per_step_serial"].items())[:18]: print("  ",k,v)
if d.get("cpu_baseline"): print(d["cpu_baseline"])
