"""Run-to-run spread of the 14-step trajectory the test `test_trajectory_follows_the_eager_path` compares (eager vs schedule):
eager x2, schedule x2, per-step largest relative difference over the loss terms for eager-eager, schedule-schedule and
eager-schedule.  Tells rounding noise of the float atomics (amplified by Adam) from a real difference between the paths."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import samnerf_amd
from test_step_program_gpu import _trainer, _run

def rel(a, b):
    out = []
    for x, y in zip(a, b):
        m, mk = 0.0, None
        for k, v in x.items():
            r = abs(y[k] - v) / max(1e-3, abs(v))
            if r > m: m, mk = r, k
        out.append((m, mk))
    return out

def go(static, overlap):
    t = _trainer("samnerf_distill", static, 256, 12)
    t.overlap = overlap; t.pipeline_steps = overlap
    return _run(t, 14)

overlap = True
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
E = [go(False, overlap) for _ in range(reps)]
S = [go(True, overlap) for _ in range(reps)]
def worst(pairs):
    w = [(0.0, None)] * 14
    for a, b in pairs:
        r = rel(a, b)
        w = [max(x, y, key=lambda t: t[0]) for x, y in zip(w, r)]
    return w
ee = worst([(E[i], E[j]) for i in range(reps) for j in range(i + 1, reps)])
ss = worst([(S[i], S[j]) for i in range(reps) for j in range(i + 1, reps)])
es = worst([(e, s) for e in E for s in S])
print("step  eager-eager        sched-sched        eager-sched")
for i in range(14):
    print(f"{i:3d}  {ee[i][0]:.2e} {str(ee[i][1]):12s} {ss[i][0]:.2e} {str(ss[i][1]):12s} {es[i][0]:.2e} {str(es[i][1]):12s}")
