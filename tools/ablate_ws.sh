#!/bin/bash
# tools/ablate_ws.sh: microbench_ws.py with parts of k_gemm_ws_b3 compiled out (SNF_WS_ABLATE bits: 1 epilogue, 2 MFMAs + B reads, 4 A loads)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
PKG=segment-anything-in-nerf_amd
echo "== stock"; python tools/microbench_ws.py
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $a -I include -c $PKG/csrc/linear_b3.hip -o $PKG/lib/obj/linear_b3.o 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/obj/*.o -o $PKG/lib/libsamnerf_hip.so || exit 1
  echo "== $a"; python tools/microbench_ws.py
done
