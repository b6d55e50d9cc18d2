#!/bin/bash
# same-box ALTERNATING A/B of a compile-time switch on the step:  tools/ab_define_alt.sh <source.hip> "<-DNAME=VALUE ...>" [kernel-key ...]
# builds the variant library once (on the GPU box's scratch copy), then runs bench.py stock, variant, stock, variant.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
src="$1"; defs="$2"; shift 2
keys="$*"
PKG=segment-anything-in-nerf_amd
LIB=$PKG/lib/libsamnerf_hip.so
python -c "import sys; sys.path.insert(0,'.'); import samnerf_amd; from samnerf_amd import _lib; _lib.build()" > /dev/null 2>&1
cp $LIB /tmp/lib_stock.so
obj=$PKG/lib/obj/${src%.hip}.o
cp $obj /tmp/obj_stock.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $defs -I include -c $PKG/csrc/$src -o $obj 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/obj/*.o -o /tmp/lib_variant.so || exit 1
cp /tmp/obj_stock.o $obj
one() {
  cp /tmp/lib_$1.so $LIB
  python bench.py --steps ${STEPS:-40} --warmup 8 --cpu-baseline-seconds 0 --other-workloads ${OTHERS:-none} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_serial']
o=d.get('other_workloads') or {}
print('$1'.ljust(8), 'step', round(d['ms_per_step'],3), 'fb', round(d['fwd_bwd_only']['ms_per_step'],3), 'serial', d['serial_step_ms'], {q: k.get(q) for q in '$keys'.split()}, {n: v['ms_per_step'] for n, v in o.items()})"
}
one stock; one variant; one stock; one variant
cp /tmp/lib_stock.so $LIB
