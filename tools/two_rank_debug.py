import os, subprocess, sys, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from tests.test_model_gpu import _TWO_RANK_SCRIPT as S
base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
base.update(SNF_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT="29417", SNF_NSTEP=os.environ.get("NSTEP", "1"))
base.update({k: v for k, v in os.environ.items() if k.startswith("SNF_")})
r = subprocess.run([sys.executable, "-c", S], env=dict(base, SNF_MODE="ref", SNF_OUT="/tmp/ref.pt"), capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-2000:]
ps = [subprocess.Popen([sys.executable, "-c", S], env=dict(base, SNF_MODE="ranks", SNF_OUT="/tmp/rk.pt", SNF_DIST_BACKEND="gloo",
      RANK=str(i), LOCAL_RANK=str(i), WORLD_SIZE="2"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(2)]
for p in ps:
    so, se = p.communicate()
    assert p.returncode == 0, se[-2000:]
a, b = torch.load("/tmp/ref.pt"), torch.load("/tmp/rk.pt")
for k in ("sam_field", "fields", "proposal_networks", "conv"):
    m0, m1 = a[k + ".exp_avg"].double(), b[k + ".exp_avg"].double()
    d = (m0 - m1).abs()
    mx = float(m0.abs().max())
    print(k, "max|m|", mx, "max abs diff", float(d.max()), "rel to max", float(d.max()) / mx)
    for lo in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5):
        sel = m0.abs() > lo * mx
        if int(sel.sum()):
            print(f"   |m| > {lo:g} max: n={int(sel.sum())}  max rel diff {float((d[sel] / m0.abs()[sel]).max()):.3e}")
    if k == "sam_field":
        rel = d / m0.abs().clamp_min(1e-30)
        bad = torch.nonzero((rel > 1e-3) & (m0.abs() > 1e-5 * mx)).flatten()
        print("   bad elems:", int(bad.numel()), "by level-chunk:", torch.bincount(bad // 131072).tolist()[:60])
        for i in bad[:8]:
            print("     ", int(i), float(m0[i]), float(m1[i]))
