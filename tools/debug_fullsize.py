"""Where does the composed step at full table size differ from the oracle?  Per-ray errors of the rendered ClipSeg / SAM features,
top-K selection (ids, sharpened weights) and fine weights of the worst rays.  usage: python tools/debug_fullsize.py"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import samnerf_oracle as O
import samnerf_amd  # noqa: F401
from samnerf_amd import configs, tcnn_compat
from samnerf_amd.interop import load_named_params
from samnerf_amd.rays import RayBundle
from samnerf_amd.step_program import StepProgram

R, P, S, K, patch = 512, 64, 128, 16, 4
cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, num_sam_samples=K, patch_size=patch)
params = O.init_params(cfg, seed=21, table_scale=0.05)
o, d = O.synthetic_rays(R, 22)
batch = O.synthetic_batch(cfg, R, 23)
gen = torch.Generator().manual_seed(24)
t_rand, u_rand = torch.rand((R, 1), generator=gen), torch.rand((R, 1), generator=gen)
with torch.no_grad():
    ref = O.forward(params, cfg, o, d, True, t_rand, u_rand, 1.0)
tc = copy.deepcopy(configs.method_configs["samnerf_distill"])
tc.pipeline.datamanager.train_num_rays_per_batch = R
mc = tc.pipeline.model
mc.num_proposal_samples_per_ray, mc.num_nerf_samples_per_ray, mc.num_sam_samples, mc.patch_size = (P,), S, K, patch
tcnn_compat.manual_seed(0)
trainer = tc.setup(device="cuda")
trainer.setup()
model = trainer.pipeline.model
load_named_params(model, params)
rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((R, 1), 1e-6, device="cuda"),
               camera_indices=torch.zeros((R, 1), dtype=torch.long, device="cuda"))
dev_batch = {k: v.cuda() for k, v in batch.items()}
trainer.pipeline.datamanager.next_train = lambda step: (copy.copy(rb), dev_batch)
ps = model.proposal_sampler
ps.initial_sampler.jitter_override, ps.pdf_sampler.jitter_override = t_rand.cuda(), u_rand.cuda()
ps.set_anneal(1.0)
prog = StepProgram(trainer)
trainer.optimizers.enabled = False
prog.run(0)
trainer.synchronize()
torch.cuda.synchronize()
out = prog.outputs()
b = prog.bufs
w1 = b["w1"].cpu()
print("fine weights max err", float((w1 - ref["weights_fine"].reshape(R, S)).abs().max()), "sbins", float((b["sb1"].cpu() - ref["sbins_fine"].reshape(R, S + 1)).abs().max()))
ids = b["ids"].cpu().long()
rid = ref["sam_ids"].reshape(R, K)
same_set = torch.tensor([set(ids[r].tolist()) == set(rid[r].tolist()) for r in range(R)])
print("rays whose top-K SET differs:", int((~same_set).sum()), "of", R)
for name in ("clipseg", "sam_fm"):
    got = (out["clipseg"] if name == "clipseg" else b["sam_fm"]).cpu().reshape(R, -1)
    rf = (ref["clipseg"] if name == "clipseg" else ref["sam_raw"]).reshape(R, -1)
    err = (got - rf).abs().max(dim=1).values
    top = torch.argsort(err, descending=True)[:6]
    print(name, "max err", float(err.max()), "rays above 1e-4:", int((err > 1e-4).sum()), "worst rays", top.tolist(), [f"{float(err[i]):.1e}" for i in top],
          "set differs there:", [bool(~same_set[i]) for i in top])
r = int(torch.argsort(((out["clipseg"].cpu().reshape(R, -1) - ref["clipseg"].reshape(R, -1)).abs().max(dim=1).values), descending=True)[0])
wk = b[f"wk@0"].cpu()
print("worst ray", r, "\n hip ids", sorted(ids[r].tolist()), "\n ref ids", sorted(rid[r].tolist()))
print(" hip wk", sorted([f"{x:.3e}" for x in wk[r].tolist()], reverse=True)[:6], "\n ref wk", sorted([f"{x:.3e}" for x in ref["sam_weights"].reshape(R, K)[r].tolist()], reverse=True)[:6])
ws, order = torch.sort(ref["weights_fine"].reshape(R, S)[r], descending=True)
print(" ref fine weights around the K-th:", [f"{float(x):.6e}" for x in ws[K - 3:K + 3]], " hip same samples:", [f"{float(w1[r][i]):.6e}" for i in order[K - 3:K + 3]])
