cd $GRAFT_REPO_ROOT
one() { # label, env
  env $2 python bench.py --steps 40 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=d['kernel_ms_per_step_serial']
print('$1'.ljust(28), 'step', round(d['ms_per_step'],3), 'pair serial', round(r['serial']['avg_launch_ms'],4), 'F2L16', k.get('snf_hashgrid_bwd_presorted_adam/F2L16'), 'wgrad_rows', k.get('snf_linear_bwd_weight_rows/192x256'))"
}
one base _X=1
one "HG_LONG=8" SNF_HG_LONG=8
one "HG_LONG=32" SNF_HG_LONG=32
one "WGRAD_CHUNKS=128" SNF_WGRAD_FULL_CHUNKS=128
one "WGRAD_CHUNKS=512" SNF_WGRAD_FULL_CHUNKS=512
one base2 _X=1
PKG=segment-anything-in-nerf_amd
rebuild() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $1 -I include -c $PKG/csrc/hashgrid.hip -o $PKG/lib/obj/hashgrid.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/obj/*.o -o $PKG/lib/libsamnerf_hip.so; }
rebuild "-DSNF_HG_EPI=1"; one "HG_EPI=1" _X=1
rebuild "-DSNF_FX_T=1024"; one "FX_T=1024" _X=1
rebuild "-DSNF_HG_EPI=4"; one "HG_EPI=4" _X=1
