#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s8; mkdir -p $O; echo "SNF_STREAM_CACHE=${SNF_STREAM_CACHE:-1}"
python - > $O/spin.txt 2>/dev/null <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
import bench
def blocks(tr, n0, nblocks):
    out = []
    for b in range(nblocks):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10):
            tr.train_iteration(n0 + b * 10 + i)
        torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) / 10 * 1e3, 3))
    return out
for spin in (0, 0):
    tr = bench.build_trainer(dict(bench.WORKLOADS["distill_4096x128"], world=1), 0, 1)
    tr.train_iteration(0); torch.cuda.synchronize()
    if spin:  # ~1 s of unrelated HBM + matrix load before the steps
        a = torch.empty((1 << 28,), device="cuda"); m = torch.randn((4096, 4096), device="cuda")
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            a.add_(1.0); m = (m @ m) * 1e-4
            torch.cuda.synchronize()
        del a, m
    print("spin", spin, "blocks of 10 steps:", blocks(tr, 1, 12))
    # anneal / update pattern held: same trainer, later blocks after an idle pause of 2 s
    time.sleep(2.0)
    print("   after 2 s idle:", blocks(tr, 121, 6))
    bench._free(tr)
PY
cat $O/spin.txt
