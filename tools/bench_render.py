"""Eval path (SURVEY 8f rank 1 / BASELINE config #5, render half): full-image render of RGB / depth + the SAM feature map
(patch-rendered [fh*p, fw*p] ray grid) + the ClipSeg map, no_grad, full-size tables, one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from samnerf_amd.rays import RayBundle

H = W = int(os.environ.get("RES", "512"))
tr = bench.build_trainer(bench.WORKLOADS["distill_4096x128"], 0, 1)
model = tr.pipeline.model
model.eval()
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.rand((H, W, 3), device="cuda", generator=g) - 0.5
d = torch.nn.functional.normalize(torch.randn((H, W, 3), device="cuda", generator=g), dim=-1)
cam = RayBundle(origins=o, directions=d, pixel_area=torch.full((H, W, 1), 1e-6, device="cuda"),
                camera_indices=torch.zeros((H, W, 1), dtype=torch.long, device="cuda"))
for _ in range(2):
    out = model.get_outputs_for_camera_ray_bundle(cam)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    out = model.get_outputs_for_camera_ray_bundle(cam)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
shapes = {k: tuple(v.shape) for k, v in out.items() if torch.is_tensor(v)}
print(f"render {H}x{W}: {ms:.1f} ms per image ({H * W / ms * 1e3 / 1e6:.1f} M rays/s of the RGB pass); outputs {shapes}")
