"""Which schedule of test_prologue_on_the_side_stream_keeps_the_trajectory varies from run to run, and where."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import samnerf_amd
from samnerf_amd import step_program
from test_step_program_gpu import _trainer

def run(on, steps):
    step_program.XSTEP_PROLOGUE = on
    tr = _trainer("samnerf_no_distill", True, 1024, 13, P=64, S=64)
    tr.pipeline_steps = True
    torch.manual_seed(17)
    for step in range(steps):
        tr.train_iteration(step)
    tr.synchronize(); torch.cuda.synchronize()
    a = tr.optimizers.arenas["proposal_networks"]
    return a.exp_avg.clone(), a.param.clone(), list(a.offsets.items()) if hasattr(a, "offsets") else None

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ref_m, ref_p, offs = run(False, steps)
print("offsets", [(k, v) for k, v in (offs or [])][:6])
for mode in (False, False, True, True, True, True, True, True):
    m, p, _ = run(mode, steps)
    d = (m - ref_m).abs()
    idx = torch.nonzero(d > 1e-6 * float(ref_m.abs().max())).flatten()
    print("xstep" if mode else "serial", "rel diff", float(d.max() / ref_m.abs().max()), "n differing", int(idx.numel()),
          "first idx", idx[:5].tolist(), "last idx", idx[-3:].tolist(), "param diff", float((p - ref_p).abs().max()))

# where do the parameters differ after ONE step between two runs, and how small were those elements' gradients?
print("---- one step, element-wise")
base_m, base_p, _ = run(False, 1)
for _ in range(6):
    m, p, _ = run(False, 1)
    dp = (p - base_p).abs()
    idx = torch.nonzero(dp > 1e-5).flatten()
    print("param elements differing > 1e-5:", idx.tolist()[:12], "their |exp_avg|", [float(x) for x in base_m[idx[:12]].abs()],
          "largest |exp_avg|", float(base_m.abs().max()))
