"""ViT-H image encoder forward time on the HIP kernels (1024 x 1024 image, random weights) with the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import samnerf_amd
from samnerf_amd import ops
from samnerf_amd.image_encoder import build_sam_vit_h_encoder

enc = build_sam_vit_h_encoder().eval()
for p in enc.parameters():
    p.data.normal_(0, 0.02)
enc.reset_weight_cache()
x = torch.randn((1, 3, 1024, 1024), device="cuda")
for _ in range(2):
    y = enc(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    y = enc(x)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"ViT-H forward: {ms:.1f} ms / image  ({5.5e12 / (ms * 1e-3) / 1e12:.0f} TFLOP/s of useful fp32-equivalent work), finite={bool(torch.isfinite(y).all())}")
ops.enable_kernel_timing("all")
y = enc(x)
s = ops.kernel_timing_summary()
for k, v in sorted(s.items(), key=lambda kv: -kv[1]["total_ms"])[:12]:
    print(f"  {k:34s} {v['launches']:4d} launches  {v['total_ms']:8.2f} ms")
