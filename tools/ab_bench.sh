#!/bin/bash
# A/B of two prebuilt libraries on ONE GPU box (box-to-box variation is larger than most kernel changes):
#   tools/ab/libA.so, tools/ab/libB.so are copied over the in-tree library in turn; bench.py runs ROUNDS times each.
#   KEYS="substr substr": serial per-kernel times printed beside the step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
LIB=segment-anything-in-nerf_amd/lib/libsamnerf_hip.so
cp $LIB /tmp/lib_keep.so
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in ${VARIANTS:-A B}; do
    cp tools/ab/lib$v.so $LIB
    timeout 200 python bench.py --steps ${STEPS:-60} --warmup 10 --cpu-baseline-seconds 0 --other-workloads none --steady-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_serial']
keys='${KEYS:-mlp64}'.split()
print('$v', round(d['ms_per_step'],3), 'fb', round(d['fwd_bwd_only']['ms_per_step'],3), 'serial', d['serial_step_ms'], {n:v for n,v in k.items() if any(q in n for q in keys)})"
  done
done
cp /tmp/lib_keep.so $LIB
