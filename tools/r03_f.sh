#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
out=$ROOT/gpurun_out/${1:-r03f}; mkdir -p $out
python -m pytest tests/test_ops_gpu.py -q -x -k "reachable or pair" 2>&1 | tail -2
SP=1 CASES=f8a POS=real REPS=20 python tools/microbench_hgadam.py 2>/dev/null | grep "^f8a" | cut -c1-200
ROWS_ADAM=1 CASES=f8a POS=real REPS=20 python tools/microbench_hgadam.py 2>/dev/null | grep "^f8a" | cut -c1-200
export SNF_HG_SPARSE_LEVELS=0
tools/ablate_step.sh $out/ablate.txt snf_hashgrid_bwd_presorted_adam_pair snf_hashgrid_bwd_presorted_adam/F2L16 snf_mlp64_bwd_fused/31x64x64x3 snf_mlp64_bwd_fused/32x64x16 snf_mlp64_fwd/31x64x64x3 snf_mlp64_fwd/32x64x16 snf_hashgrid_fwd/F2L16 snf_hashgrid_fwd/F8L12 snf_linear_bwd_weight_rows snf_linear_bwd_data_rows snf_linear_fwd_mean 2304x256 snf_adam_step_rows snf_linear_ snf_mlp64_ snf_hashgrid_bwd_presorted_adam/F2L5
