"""method_configs["samnerf_no_distill"] / ["samnerf_distill"] with the reference's values (samnerf/samconfigs.py:51-164).

Dataparser / viewer / wandb entries of the reference configs belong to subsystems outside the hot path and are not
reproduced; every model, datamanager and optimizer field is."""
from __future__ import annotations

from typing import Dict

from .engine import AdamOptimizerConfig, ExponentialDecaySchedulerConfig
from .model import SAMModelConfig
from .pipeline import CameraOptimizerConfig, SAMDataManagerConfig, SamPipelineConfig, TrainerConfig

method_configs: Dict[str, TrainerConfig] = {}


def _opt(lr, lr_final, max_steps):
    return {"optimizer": AdamOptimizerConfig(lr=lr, eps=1e-15),
            "scheduler": ExponentialDecaySchedulerConfig(lr_final=lr_final, max_steps=max_steps)}


method_configs["samnerf_no_distill"] = TrainerConfig(
    method_name="samnberf_no_distill",  # (sic) samconfigs.py:52
    steps_per_eval_batch=50000, steps_per_eval_image=10000000, steps_per_save=2000, max_num_iterations=30000,
    mixed_precision=True,
    pipeline=SamPipelineConfig(
        datamanager=SAMDataManagerConfig(use_dino_feature=False, train_num_rays_per_batch=4096 * 4,
                                         eval_num_rays_per_batch=4096 * 4,
                                         camera_optimizer=CameraOptimizerConfig(mode="off"), patch_size=1,
                                         distill_sam=False),
        model=SAMModelConfig(distill_sam=False, kernel_size=3, use_clipseg_feature=False,
                             eval_num_rays_per_chunk=1 << 15, use_appearance_embedding=False, hidden_layers=1,
                             patch_size=1, sam_loss_weight=1.0, num_proposal_iterations=1,
                             num_proposal_samples_per_ray=(64,), num_sam_samples=3, num_nerf_samples_per_ray=32)),
    optimizers={"proposal_networks": _opt(1e-2, 0.0005, 30000), "fields": _opt(1e-2, 0.0005, 30000)},
    vis="viewer+wandb")

method_configs["samnerf_distill"] = TrainerConfig(
    method_name="samnerf_distill",
    steps_per_eval_batch=5000000, steps_per_eval_image=10000000, steps_per_save=2000, max_num_iterations=10000,
    mixed_precision=True,
    pipeline=SamPipelineConfig(
        datamanager=SAMDataManagerConfig(use_dino_feature=False, train_num_rays_per_batch=4096 * 4,
                                         eval_num_rays_per_batch=4096 * 4,
                                         camera_optimizer=CameraOptimizerConfig(mode="off"), patch_size=4,
                                         distill_sam=True, use_clipseg_feature=True),
        model=SAMModelConfig(distill_sam=True, kernel_size=3, use_clipseg_feature=True,
                             eval_num_rays_per_chunk=1 << 15, use_appearance_embedding=False, hidden_layers=1,
                             patch_size=4, sam_loss_weight=1.0, num_proposal_iterations=1,
                             num_proposal_samples_per_ray=(64,), num_sam_samples=16, num_nerf_samples_per_ray=32)),
    optimizers={"proposal_networks": _opt(1e-2, 0.0005, 10000), "fields": _opt(1e-2, 0.0005, 10000),
                "conv": _opt(5e-4, 0.0001, 10000), "sam_field": _opt(5e-4, 0.0001, 10000)},
    vis="viewer+wandb")

for _k in method_configs:
    method_configs[_k].wandb_name = _k
