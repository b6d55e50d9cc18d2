import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import samnerf_amd
from samnerf_amd import ops
m = ops
g = torch.Generator(device="cuda").manual_seed(0)
N = 1 << 17
for (I, nh, O, act) in ((31, 2, 3, ops.ACT_SIGMOID), (32, 1, 16, ops.ACT_NONE)):
    x = torch.randn((N, 32), device="cuda", generator=g) * 0.5
    w0 = (torch.rand((64, I), device="cuda", generator=g) - 0.5) * 0.6
    w1 = (torch.rand((64, 64), device="cuda", generator=g) - 0.5) * 0.4
    wo = (torch.rand((O, 64), device="cuda", generator=g) - 0.5) * 0.4
    h1 = torch.empty((N, 64), device="cuda"); h2 = torch.empty((N, 64), device="cuda"); y = torch.empty((N, O), device="cuda")
    st = m._stream()
    m._launch("snf_mlp64_fwd", m._p(x), 32, m._p(w0), I, m._p(w1) if nh == 2 else None, m._p(wo), nh, O, act, N, m._p(h1), m._p(h2) if nh == 2 else None, m._p(y), O, st)
    torch.cuda.synchronize()
    xd = x[:, :I].double()
    r1 = torch.relu(xd @ w0.double().T)
    r = torch.relu(r1 @ w1.double().T) if nh == 2 else r1
    ry = r @ wo.double().T
    if act == ops.ACT_SIGMOID: ry = torch.sigmoid(ry)
    print(f"I={I} nh={nh}: h1 rel {float((h1.double()-r1).abs().max()/r1.abs().max()):.2e}  y rel {float((y.double()-ry).abs().max()/ry.abs().max()):.2e}  mask flips h1 {int(((h1>0)!=(r1>0)).sum())} of {h1.numel()}")
