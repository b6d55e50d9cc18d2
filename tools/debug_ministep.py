"""Per-tensor gradient errors of the static schedule against a reference-generated ministep fixture (tests/golden/<name>.npz,
default `ministep`; make_golden._ministep(name, R, T, S, table_scale, anneal_step) writes other shapes), the field grid's error per
level, and -- for the base MLP's hidden units whose weight-gradient row is off -- the single sample whose encoding the error
vector is parallel to and that sample's pre-activation: round 2 used this to show that the 0.5-4 % max-normalised gradient errors
of the smoke-shaped step (R=128, S=32, T=12) are ReLU masks of one (sample, unit) pair with |pre| < 2e-6 flipping between the
fp32 CPU evaluation and the HIP path, not a kernel defect.   usage: python tools/debug_ministep.py [fixture]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import samnerf_oracle as O
import samnerf_amd
from samnerf_amd import configs, tcnn_compat
from samnerf_amd.interop import load_named_params, named_grads
from samnerf_amd.rays import RayBundle
from samnerf_amd.step_program import StepProgram

name = sys.argv[1] if len(sys.argv) > 1 else "ministep"
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + ".npz"))
P, S, K, patch, T, R = (int(g[k]) for k in ("P", "S", "K", "patch", "log2_T", "num_rays"))
cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, num_sam_samples=K, patch_size=patch).small(T)
params = O.init_params(cfg, seed=int(g["seed_params"]), table_scale=float(g["table_scale"]))
tc = copy.deepcopy(configs.method_configs["samnerf_distill"])
tc.pipeline.datamanager.train_num_rays_per_batch = R
mc = tc.pipeline.model
mc.num_proposal_samples_per_ray, mc.num_nerf_samples_per_ray, mc.num_sam_samples, mc.patch_size = (P,), S, K, patch
mc.log2_hashmap_size, mc.hashgrid_sizes = min(19, T), (min(19, T),) * 2
mc.proposal_net_args_list = [dict(a, log2_hashmap_size=min(17, T)) for a in mc.proposal_net_args_list]
trainer = tc.setup(device="cuda")
trainer.setup()
model = trainer.pipeline.model
load_named_params(model, params)
o, d = torch.from_numpy(g["origins"]), torch.from_numpy(g["directions"])
batch = {k: v.cuda() for k, v in O.synthetic_batch(cfg, R, int(g["seed_batch"])).items()}
rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((R, 1), 1e-6, device="cuda"),
               camera_indices=torch.zeros((R, 1), dtype=torch.long, device="cuda"))
trainer.pipeline.datamanager.next_train = lambda step: (copy.copy(rb), batch)
ps = model.proposal_sampler
ps.initial_sampler.jitter_override = torch.from_numpy(g["t_rand"]).cuda()
ps.pdf_sampler.jitter_override = torch.from_numpy(g["u_rand"]).cuda()
ps.set_anneal(float(g["anneal"]))
prog = StepProgram(trainer)
trainer.optimizers.enabled = False
prog.run(0)
for st in (trainer._side or {}).values():
    torch.cuda.current_stream().wait_stream(st)
torch.cuda.synchronize()
grads = named_grads(model)
for k in params:
    ref = g["grad_" + k]
    got = grads[k].cpu().numpy().reshape(ref.shape)
    scale = max(float(np.abs(ref).max()), 1e-8)
    err = np.abs(got - ref)
    i = np.unravel_index(err.argmax(), err.shape)
    print(f"{k:16s} rel-to-max {err.max() / scale:.2e}  at {i}  got {got[i]:+.6e} ref {ref[i]:+.6e}  max|ref| {scale:.3e}  "
          f"nbad(>1e-3 max) {(err > 1e-3 * scale).sum()} of {err.size}")
# density / weights diagnostics
w1 = prog.bufs["w1"].cpu().numpy()
print("w_fine max diff", np.abs(w1 - g["w_fine"]).max(), " rows with acc ~0:", int((w1.sum(-1) < 1e-6).sum()))
ref = g["grad_field_table"]; got = grads["field_table"].cpu().numpy().reshape(ref.shape)
L = 16; per = ref.shape[0] // L
for l in range(L):
    r, q = ref[l * per:(l + 1) * per], got[l * per:(l + 1) * per]
    print(f"level {l:2d}: sum|err|/sum|ref| {np.abs(q - r).sum() / max(np.abs(r).sum(), 1e-30):.3e}  nnz ref {(r != 0).sum()} got {(q != 0).sum()}  max|ref| {np.abs(r).max():.2e}")
ref = g["grad_base_w0"]; got = grads["base_w0"].cpu().numpy().reshape(ref.shape)
print("base_w0 shape", ref.shape)
e = np.abs(got - ref)
print("err by axis0 (sum):", np.array2string(e.sum(1) / np.abs(ref).sum(1).clip(1e-30), precision=2, max_line_width=200))
print("err by axis1 (sum):", np.array2string(e.sum(0) / np.abs(ref).sum(0).clip(1e-30), precision=2, max_line_width=200))
# hidden unit 14 / 63 of the base MLP: pre-activation from the HIP encodings in fp64
N1 = R * S
enc = prog.bufs["enc1"].view(16, N1, 2).permute(1, 0, 2).reshape(N1, 32).double()
W0 = torch.as_tensor(params["base_w0"]).detach().cuda().double().view(64, 32)
pre = enc @ W0.t()
hb1 = prog.bufs["hb1"].view(N1, 64)  # (set step_program.CHAIN_RECOMPUTE = False first: by default the schedule does not store the hidden activations)
for j in (14, 63, 0, 5):
    p = pre[:, j]
    print(f"unit {j}: min|pre| {p.abs().min():.3e}  n(|pre|<1e-6) {(p.abs() < 1e-6).sum().item()}  n(|pre|<1e-4) {(p.abs() < 1e-4).sum().item()} "
          f" n(pre>0) {(p > 0).sum().item()}  n(hb1>0) {(hb1[:, j] > 0).sum().item()}  mask mismatches {((p > 0) != (hb1[:, j] > 0)).sum().item()}  max|pre| {p.abs().max():.3e}")
print("hb1 vs relu(pre) max diff", (hb1.double() - pre.clamp(min=0)).abs().max().item())
ref = torch.from_numpy(g["grad_base_w0"]).cuda().double(); got = grads["base_w0"].double().view(64, 32)
for j in (14, 63):
    e = (got - ref)[j]
    cos = (enc @ e) / (enc.norm(dim=1) * e.norm() + 1e-300)
    top = cos.abs().topk(4)
    print(f"unit {j}: |e| {e.norm():.3e}; top |cos| {top.values.tolist()} at samples {top.indices.tolist()} (ray, s) {[(i // S, i % S) for i in top.indices.tolist()]}")
    for i in top.indices.tolist()[:2]:
        print(f"   sample {i}: pre {pre[i, j].item():+.3e} hb1 {hb1[i, j].item():+.3e}  implied delta dH0 {(enc[i] @ e / (enc[i] @ enc[i])).item():+.3e}")
