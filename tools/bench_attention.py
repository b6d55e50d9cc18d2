import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, samnerf_amd
from samnerf_amd import ops
def t(Bw, n, heads, hd, rel):
    T, C = n*n, heads*hd
    qkv = torch.randn((Bw*T, 3*C), device="cuda")
    rph = torch.randn((2*n-1, hd), device="cuda") if rel else None
    rpw = torch.randn((2*n-1, hd), device="cuda") if rel else None
    ops.enable_kernel_timing("all")
    for _ in range(5): ops.attention(qkv, Bw, T, heads, n, rph, rpw)
    s = ops.kernel_timing_summary()
    fl = 4.0*Bw*heads*T*T*hd
    print(Bw, n, heads, hd, rel, {k: round(v["total_ms"]/v["launches"],3) for k,v in s.items()}, "att TF", round(fl/(s["snf_attention"]["total_ms"]/5*1e-3)/1e12,1))
t(25,14,16,80,True); t(25,14,16,80,False); t(1,64,16,80,True); t(1,64,16,80,False); t(1,64,16,64,False)
