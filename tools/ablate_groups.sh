#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tools/ablate_step.sh gpurun_out/r03n_ablation.txt \
 "snf_hashgrid_bwd_presorted_adam/F2L16" \
 "snf_hashgrid_bwd_presorted_adam/F2L16,snf_hashgrid_sort/L16" \
 "snf_hashgrid_bwd_presorted_adam_pair" \
 "snf_hashgrid_bwd_presorted_adam_pair,snf_hashgrid_sort/L12" \
 "snf_composite,snf_rowmse,snf_trunc_exp,snf_weights,snf_distortion,snf_interlevel,snf_add_scaled,snf_nerf_loss_summary,snf_head_input" \
 "snf_patch,snf_feature_mean,256x256r,256x192r" \
 "snf_mlp_tiny,snf_hashgrid_fwd/F2L5,snf_hashgrid_bwd_presorted_adam/F2L5,snf_hashgrid_sort/L5" \
 "snf_adam_step" \
 "snf_sample_spacing,snf_pdf_resample"
