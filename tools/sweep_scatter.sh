cd $GRAFT_REPO_ROOT
PKG=segment-anything-in-nerf_amd
one() { python bench.py --steps 40 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_serial']
print('$1'.ljust(16), 'step', round(d['ms_per_step'],3), 'sort L16', k.get('snf_hashgrid_sort/L16'), 'L12', k.get('snf_hashgrid_sort/L12'), 'L5', k.get('snf_hashgrid_sort/L5'))"; }
rebuild() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $1 -I include -c $PKG/csrc/hashgrid.hip -o $PKG/lib/obj/hashgrid.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/obj/*.o -o $PKG/lib/libsamnerf_hip.so; }
one "SB_SPT=2"
rebuild "-DSNF_HG_SB_SPT=1"; one "SB_SPT=1"
rebuild "-DSNF_HG_SB_SPT=4"; one "SB_SPT=4"
rebuild "-DSNF_HG_SB_SPT=2"; one "SB_SPT=2"
