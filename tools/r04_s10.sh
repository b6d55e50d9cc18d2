#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s10; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_step_program_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -x -q -k "colour_net or sh16 or head_input or ministep or trajectory or composed or schedule or render or patch_render" 2>&1 | tail -6 > $O/tests.txt
timeout 900 bash tools/ab_env.sh SNF_FUSED_SH_INPUT=0 snf_mlp64_fwd/31x64x64x3 snf_mlp64_fwd_sh/31x64x64x3 snf_mlp64_bwd_fused/31x64x64x3 snf_mlp64_bwd_fused_sh/31x64x64x3 snf_mlp64_bwd_fused/32x64x16 snf_head_input > $O/ab_sh.txt 2>&1
( python tools/bench_render.py; SNF_FUSED_SH_INPUT=0 python tools/bench_render.py; python tools/bench_render.py; SNF_FUSED_SH_INPUT=0 python tools/bench_render.py ) 2>/dev/null | grep "^render" | cut -c1-60 > $O/render.txt
cat $O/tests.txt $O/ab_sh.txt $O/render.txt
