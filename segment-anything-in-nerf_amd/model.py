"""SAMModel / NerfactoModel surface of the reference (samnerf/sam_model.py:140-335, nerfstudio/models/nerfacto.py:68-344,
nerfstudio/models/base_model.py:57-226) on the MI355X kernels.

Same config field names and defaults, same method names / kwargs, same output-dict keys, same parameter groups
(`proposal_networks`, `fields`, `sam_field`, `conv`).  Differences that are deliberate:
  * perception models (SAM ViT-H predictor, ClipSeg decoder, LanguageSAM) are NOT loaded by `populate_modules`
    -- they are outside the hot path and the reference cannot even be constructed without CUDA
    and checkpoint files (sam_model.py:210-224);
  * parameters live in flat per-group arenas (arena.py) so the optimiser and the RCCL all-reduce see four
    contiguous buffers instead of dozens of tensors.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Tuple, Type

import numpy as np
import torch
from torch import nn
from torch.nn import Parameter

from . import ops
from .arena import ParamGroupArena
from .fields import FieldHeadNames, HashMLPDensityField, SAMField, TCNNNerfactoField
from .losses import MSELoss, distortion_loss, interlevel_loss
from .rays import RayBundle, RaySamples
from .renderers import AccumulationRenderer, DepthRenderer, MeanRenderer, RGBRenderer
from .samplers import ProposalNetworkSampler
from .spatial_distortions import SceneContraction


# ---------------------------------------------------------------------------------------------
# config plumbing (nerfstudio/configs/base_config.py:52-59)
# ---------------------------------------------------------------------------------------------
@dataclass
class InstantiateConfig:
    _target: Type = None

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)


@dataclass
class SceneBox:
    """nerfstudio/data/scene_box.py: axis-aligned box [2,3]."""
    aabb: torch.Tensor = field(default_factory=lambda: torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]))


class NearFarCollider(nn.Module):
    """nerfstudio/model_components/scene_colliders.py:170-189 (near plane 0 in eval mode)."""

    def __init__(self, near_plane: float, far_plane: float) -> None:
        super().__init__()
        self.near_plane, self.far_plane = near_plane, far_plane

    def set_nears_and_fars(self, ray_bundle: RayBundle) -> RayBundle:
        ones = torch.ones_like(ray_bundle.origins[..., 0:1])
        near_plane = self.near_plane if self.training else 0
        ray_bundle.nears = ones * near_plane
        ray_bundle.fars = ones * self.far_plane
        return ray_bundle

    def forward(self, ray_bundle: RayBundle) -> RayBundle:
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            return ray_bundle
        return self.set_nears_and_fars(ray_bundle)


@dataclass
class ModelConfig(InstantiateConfig):
    """nerfstudio/models/base_model.py:40-54."""
    _target: Type = field(default_factory=lambda: Model)
    enable_collider: bool = True
    collider_params: Optional[Dict[str, float]] = field(default_factory=lambda: {"near_plane": 2.0, "far_plane": 6.0})
    loss_coefficients: Dict[str, float] = field(default_factory=lambda: {"rgb_loss_coarse": 1.0, "rgb_loss_fine": 1.0})
    eval_num_rays_per_chunk: int = 4096


class Model(nn.Module):
    """nerfstudio/models/base_model.py:57-226."""

    config: ModelConfig

    def __init__(self, config: ModelConfig, scene_box: SceneBox, num_train_data: int, **kwargs) -> None:
        super().__init__()
        self.config = config
        self.scene_box = scene_box
        self.num_train_data = num_train_data
        self.kwargs = kwargs
        self.collider = None
        self.populate_modules()
        self.callbacks = None
        dev = kwargs.get("device", None) or ("cuda" if torch.cuda.is_available() else "cpu")
        self.device_indicator_param = nn.Parameter(torch.empty(0, device=dev))

    @property
    def device(self):
        return self.device_indicator_param.device

    def populate_modules(self):
        pass

    def get_training_callbacks(self, training_callback_attributes=None) -> List:
        return []

    def get_param_groups(self) -> Dict[str, List[Parameter]]:  # pragma: no cover - interface
        raise NotImplementedError

    def get_outputs(self, ray_bundle: RayBundle) -> Dict[str, torch.Tensor]:  # pragma: no cover - interface
        raise NotImplementedError

    def forward(self, ray_bundle: RayBundle, **kwargs) -> Dict[str, torch.Tensor]:
        if self.collider is not None:
            ray_bundle = self.collider(ray_bundle)
        return self.get_outputs(ray_bundle, **kwargs)

    def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
        return {}

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:  # pragma: no cover
        raise NotImplementedError


@dataclass
class TrainingCallback:
    """nerfstudio/engine/callbacks.py:28-103 (the subset the model registers)."""
    where_to_run: List[str]
    func: Callable
    update_every_num_iters: Optional[int] = None

    def run_callback_at_location(self, step: int, location: str) -> None:
        if location in self.where_to_run and (self.update_every_num_iters is None or step % self.update_every_num_iters == 0):
            self.func(step)


BEFORE_TRAIN_ITERATION, AFTER_TRAIN_ITERATION = "before_train_iteration", "after_train_iteration"


@dataclass
class NerfactoModelConfig(ModelConfig):
    """nerfstudio/models/nerfacto.py:68-144 (field names and defaults kept)."""
    _target: Type = field(default_factory=lambda: NerfactoModel)
    near_plane: float = 0.05
    far_plane: float = 1000.0
    background_color: str = "last_sample"
    hidden_dim: int = 64
    hidden_dim_color: int = 64
    hidden_dim_transient: int = 64
    num_levels: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    proposal_update_every: int = 5
    proposal_warmup: int = 5000
    num_proposal_iterations: int = 2
    use_same_proposal_network: bool = False
    proposal_net_args_list: List[Dict] = field(default_factory=lambda: [
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 128, "use_linear": False},
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256, "use_linear": False},
    ])
    proposal_initial_sampler: str = "piecewise"
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    orientation_loss_mult: float = 0.0001
    pred_normal_loss_mult: float = 0.001
    use_proposal_weight_anneal: bool = True
    use_average_appearance_embedding: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    use_single_jitter: bool = True
    predict_normals: bool = False
    disable_scene_contraction: bool = False
    use_appearance_embedding: bool = True


class NerfactoModel(Model):
    """nerfstudio/models/nerfacto.py:147-344."""

    config: NerfactoModelConfig

    def populate_modules(self):
        super().populate_modules()
        c = self.config
        if c.disable_scene_contraction:
            raise NotImplementedError("scene contraction is always on in the samnerf configs")
        if c.predict_normals:
            raise NotImplementedError("predict_normals is off in the samnerf configs")
        dev = self.kwargs.get("device", None)
        scene_contraction = SceneContraction(order=float("inf"))
        self.field = TCNNNerfactoField(
            self.scene_box.aabb, hidden_dim=c.hidden_dim, num_levels=c.num_levels, max_res=c.max_res,
            log2_hashmap_size=c.log2_hashmap_size, hidden_dim_color=c.hidden_dim_color,
            hidden_dim_transient=c.hidden_dim_transient, spatial_distortion=scene_contraction,
            num_images=self.num_train_data, use_pred_normals=c.predict_normals,
            use_average_appearance_embedding=c.use_average_appearance_embedding,
            use_appearance_embedding=c.use_appearance_embedding, device=dev)
        self.density_fns = []
        num_prop_nets = c.num_proposal_iterations
        self.proposal_networks = torch.nn.ModuleList()
        if c.use_same_proposal_network:
            assert len(c.proposal_net_args_list) == 1, "Only one proposal network is allowed."
            network = HashMLPDensityField(self.scene_box.aabb, spatial_distortion=scene_contraction, device=dev,
                                          **c.proposal_net_args_list[0])
            self.proposal_networks.append(network)
            self.density_fns.extend([network.density_fn for _ in range(num_prop_nets)])
            if num_prop_nets > 1:  # one backward launch per use: the optimizer step cannot ride on any single one of them
                network.mlp_base.encoding.params.no_fused_adam = True
        else:
            for i in range(num_prop_nets):
                prop_net_args = c.proposal_net_args_list[min(i, len(c.proposal_net_args_list) - 1)]
                network = HashMLPDensityField(self.scene_box.aabb, spatial_distortion=scene_contraction, device=dev,
                                              **prop_net_args)
                self.proposal_networks.append(network)
            self.density_fns.extend([network.density_fn for network in self.proposal_networks])
        update_schedule = lambda step: np.clip(  # noqa: E731
            np.interp(step, [0, c.proposal_warmup], [0, c.proposal_update_every]), 1, c.proposal_update_every)
        if c.proposal_initial_sampler != "piecewise":
            raise NotImplementedError("the samnerf configs use the piecewise initial sampler")
        self.proposal_sampler = ProposalNetworkSampler(
            num_nerf_samples_per_ray=c.num_nerf_samples_per_ray,
            num_proposal_samples_per_ray=c.num_proposal_samples_per_ray,
            num_proposal_network_iterations=c.num_proposal_iterations, single_jitter=c.use_single_jitter,
            update_sched=update_schedule, initial_sampler=None)
        self.collider = NearFarCollider(near_plane=c.near_plane, far_plane=c.far_plane)
        self.renderer_rgb = RGBRenderer(background_color=c.background_color)
        self.renderer_accumulation = AccumulationRenderer()
        self.renderer_depth = DepthRenderer()
        self.rgb_loss = MSELoss()
        self.arenas: Dict[str, ParamGroupArena] = {}

    # -- metrics --------------------------------------------------------------------------
    @staticmethod
    def psnr(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """torchmetrics PeakSignalNoiseRatio(data_range=1.0) (nerfacto.py:232,319)."""
        return -10.0 * torch.log10(ops.mse_loss(pred, target))  # (device tensors only: ops raises on a CPU tensor)

    @staticmethod
    def ssim(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """torchmetrics `structural_similarity_index_measure` with its defaults, as nerfacto.py:233,360 calls it on [1, C, H, W] images:
        11 x 11 Gaussian window (sigma 1.5), k1 = 0.01, k2 = 0.03, data range = the larger of the two images' value ranges, reflect
        padding, mean over the un-padded map.  An eval-time logging metric (plain torch, any device) -- NOT part of the hot path and,
        torchmetrics being absent from the build image, restated from its published definition: parity unpinned."""
        pred, target = pred.float(), target.float()
        C = pred.shape[1]
        k, sigma, pad = 11, 1.5, 5
        g = torch.exp(-((torch.arange(k, dtype=torch.float32, device=pred.device) - (k - 1) / 2) ** 2) / (2 * sigma * sigma))
        g = g / g.sum()
        win = (g[:, None] * g[None, :]).expand(C, 1, k, k).contiguous()
        data_range = torch.maximum(pred.max() - pred.min(), target.max() - target.min())
        c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
        p_, t_ = (torch.nn.functional.pad(x, (pad, pad, pad, pad), mode="reflect") for x in (pred, target))
        stack = torch.cat((p_, t_, p_ * p_, t_ * t_, p_ * t_))
        out = torch.nn.functional.conv2d(stack, win, groups=C)
        mu_p, mu_t, pp, tt, pt = out.split(pred.shape[0])
        s_p, s_t, s_pt = pp - mu_p * mu_p, tt - mu_t * mu_t, pt - mu_p * mu_t
        m = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_p + s_t + c2))
        return m[..., pad:-pad, pad:-pad].mean()

    def get_image_metrics_and_images(self, outputs: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor]):
        """nerfacto.py:346-383 / sam_model.py:550-577, the scalar half: psnr and ssim of the rendered image against the batch's.  LPIPS
        (a pretrained network) and the colour-mapped viewer images are out of scope: the images dict carries the
        side-by-side RGB only."""
        image = batch["image"].to(outputs["rgb"].device)
        rgb = outputs["rgb"]
        a, b = torch.moveaxis(image, -1, 0)[None, ...], torch.moveaxis(rgb, -1, 0)[None, ...]
        mse = torch.mean((a.float() - b.float()) ** 2)
        metrics = {"psnr": float(-10.0 * torch.log10(mse)), "ssim": float(NerfactoModel.ssim(a, b))}
        return metrics, {"img": torch.cat([image, rgb], dim=1)}

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        return {"proposal_networks": list(self.proposal_networks.parameters()),
                "fields": list(self.field.parameters())}

    def get_training_callbacks(self, training_callback_attributes=None) -> List[TrainingCallback]:
        callbacks = []
        if self.config.use_proposal_weight_anneal:
            N = self.config.proposal_weights_anneal_max_num_iters

            def set_anneal(step):
                train_frac = np.clip(step / N, 0, 1)
                bias = lambda x, b: (b * x) / ((b - 1) * x + 1)  # noqa: E731
                self.proposal_sampler.set_anneal(bias(train_frac, self.config.proposal_weights_anneal_slope))

            callbacks.append(TrainingCallback([BEFORE_TRAIN_ITERATION], set_anneal, 1))
            callbacks.append(TrainingCallback([AFTER_TRAIN_ITERATION], self.proposal_sampler.step_cb, 1))
        return callbacks

    def get_outputs(self, ray_bundle: RayBundle):
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(ray_bundle, density_fns=self.density_fns)
        field_outputs = self.field(ray_samples, compute_normals=self.config.predict_normals)
        weights = ray_samples.get_weights(field_outputs[FieldHeadNames.DENSITY])
        weights_list.append(weights)
        ray_samples_list.append(ray_samples)
        outputs = {"rgb": self.renderer_rgb(rgb=field_outputs[FieldHeadNames.RGB], weights=weights),
                   "accumulation": self.renderer_accumulation(weights=weights),
                   "depth": self.renderer_depth(weights=weights, ray_samples=ray_samples)}
        if self.training:
            outputs["weights_list"] = weights_list
            outputs["ray_samples_list"] = ray_samples_list
        for i in range(self.config.num_proposal_iterations):
            outputs[f"prop_depth_{i}"] = self.renderer_depth(weights=weights_list[i], ray_samples=ray_samples_list[i])
        return outputs

    def get_metrics_dict(self, outputs, batch):
        metrics_dict = {}
        image = batch["image"].to(self.device)
        metrics_dict["psnr"] = self.psnr(outputs["rgb"].detach(), image)
        if self.training:
            metrics_dict["distortion"] = distortion_loss(outputs["weights_list"], outputs["ray_samples_list"])
        return metrics_dict

    def get_loss_dict(self, outputs, batch, metrics_dict=None):
        loss_dict = {}
        image = batch["image"].to(self.device)
        loss_dict["rgb_loss"] = self.rgb_loss(image, outputs["rgb"])  # ops.mse_loss on the GPU (losses.MSELoss)
        if self.training:
            loss_dict["interlevel_loss"] = self.config.interlevel_loss_mult * interlevel_loss(
                outputs["weights_list"], outputs["ray_samples_list"])
            assert metrics_dict is not None and "distortion" in metrics_dict
            loss_dict["distortion_loss"] = self.config.distortion_loss_mult * metrics_dict["distortion"]
        return loss_dict

    # -- arenas ---------------------------------------------------------------------------
    def build_arenas(self, with_optimizer_state: bool = True) -> Dict[str, ParamGroupArena]:
        """Move every parameter of every group into a flat per-group arena (see arena.py)."""
        self.arenas = {}
        for gname, params in self.get_param_groups().items():
            if len(params) == 0:
                continue
            dev = params[0].device
            arena = ParamGroupArena(gname, [(str(i), tuple(p.shape)) for i, p in enumerate(params)], dev,
                                    with_optimizer_state)
            for i, p in enumerate(params):
                view = arena.view(arena.param, str(i))
                view.copy_(p.data)
                p.data = view
                p.main_grad = arena.view(arena.grad, str(i))
                if getattr(p, "hash_table_of", None) is not None:
                    arena.tables[str(i)] = p.hash_table_of
                # autograd-produced gradients (conv head) accumulate in place into the same arena slice
                p.grad = p.main_grad
            self.arenas[gname] = arena
        return self.arenas

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle, **kwargs) -> Dict[str, torch.Tensor]:
        """base_model.py:166-188: chunked full-image render."""
        num_rays_per_chunk = self.config.eval_num_rays_per_chunk
        image_height, image_width = camera_ray_bundle.origins.shape[:2]
        num_rays = len(camera_ray_bundle)
        outputs_lists: Dict[str, List[torch.Tensor]] = {}
        for i in range(0, num_rays, num_rays_per_chunk):
            ray_bundle = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, i + num_rays_per_chunk)
            outputs = self.forward(ray_bundle=ray_bundle, **kwargs)
            for name, out in outputs.items():
                if torch.is_tensor(out):
                    outputs_lists.setdefault(name, []).append(out)
        return {name: torch.cat(lst).view(image_height, image_width, -1) for name, lst in outputs_lists.items()}


def feature_ray_bundle(camera_ray_bundle: RayBundle, feature_h: int, feature_w: int, p: int) -> RayBundle:
    """samnerf/sam_model.py:371-380: the [H,W] camera bundle sub-sampled with linspace indices to [fh*p, fw*p] rays and
    regrouped so that a row-major walk (get_row_major_sliced_ray_bundle) visits whole p x p patches one after another."""
    sz = camera_ray_bundle.shape
    dev = camera_ray_bundle.origins.device
    h_indices = torch.linspace(0, sz[0] - 1, feature_h * p, dtype=torch.long, device=dev)
    w_indices = torch.linspace(0, sz[1] - 1, feature_w * p, dtype=torch.long, device=dev)
    hind, wind = torch.meshgrid(h_indices, w_indices, indexing="ij")
    fb = camera_ray_bundle[hind.flatten(), wind.flatten()]
    return fb.reshape((feature_h, p, feature_w, p))._apply_fn_to_fields(lambda x: x.transpose(1, 2))


def feature_pixel_ids(H: int, W: int, feature_h: int, feature_w: int, p: int) -> torch.Tensor:
    """Row-major pixel index (h * W + w) of every ray of `feature_ray_bundle` (p > 1: patch order [fh, fw, p, p]) or, with p = 1,
    of `clipseg_ray_bundle`, in the order the chunk loop walks them -- the same linspace(...).long() indices
    (samnerf/sam_model.py:371-377,392-398), as int64 on the CPU."""
    h_indices = torch.linspace(0, H - 1, feature_h * p, dtype=torch.long)
    w_indices = torch.linspace(0, W - 1, feature_w * p, dtype=torch.long)
    hind, wind = torch.meshgrid(h_indices, w_indices, indexing="ij")
    pix = hind * W + wind  # [fh*p, fw*p]
    return pix.reshape(feature_h, p, feature_w, p).transpose(1, 2).reshape(-1)


def clipseg_ray_bundle(camera_ray_bundle: RayBundle, feature_h: int = 32, feature_w: int = 32) -> RayBundle:
    """samnerf/sam_model.py:389-398: the 32 x 32 linspace sub-sampling for the ClipSeg map."""
    sz = camera_ray_bundle.shape
    dev = camera_ray_bundle.origins.device
    h_indices = torch.linspace(0, sz[0] - 1, feature_h, dtype=torch.long, device=dev)
    w_indices = torch.linspace(0, sz[1] - 1, feature_w, dtype=torch.long, device=dev)
    hind, wind = torch.meshgrid(h_indices, w_indices, indexing="ij")
    return camera_ray_bundle[hind.flatten(), wind.flatten()].reshape((feature_h, feature_w))


@dataclass
class SAMModelConfig(NerfactoModelConfig):
    """samnerf/sam_model.py:140-162."""
    _target: Type = field(default_factory=lambda: SAMModel)
    sam_loss_weight: float = 1.0
    use_dino_feature: bool = False
    use_clipseg_feature: bool = False
    dino_loss_weight: float = 1.0
    clipseg_loss_weight: float = 1.0
    n_scales: int = 30
    max_scale: float = 1.5
    num_sam_samples: int = 24
    hidden_layers: int = 2
    hashgrid_layers: Tuple[int, ...] = (12, 12)
    hashgrid_resolutions: Tuple[Tuple[int, int], ...] = ((16, 128), (128, 512))
    hashgrid_sizes: Tuple[int, ...] = (19, 19)
    patch_size: int = 1
    kernel_size: int = 3
    distill_sam: bool = True
    sam_checkpoint: str = "samnerf/segment_anything/sam_vit_h_4b8939.pth"
    sharpening_temperature: float = 10.0


class SAMModel(NerfactoModel):
    """samnerf/sam_model.py:179-335."""

    config: SAMModelConfig

    def populate_modules(self):
        super().populate_modules()
        self.sam_capable = True
        self.text_prompt_capable = True
        self.prompts = None
        self.renderer_mean = MeanRenderer()
        self.feature_streams = None  # set by the trainer: {head: HIP stream} for the (independent) feature heads
        c = self.config
        dev = self.kwargs.get("device", None)
        if c.distill_sam:
            self.sam_field = SAMField(c.hashgrid_layers, c.hashgrid_sizes, c.hashgrid_resolutions,
                                      hidden_layers=c.hidden_layers, use_dino_features=c.use_dino_feature,
                                      use_clipseg_features=c.use_clipseg_feature, device=dev)
            pad = (c.kernel_size - 1) // 2
            self.conv_head = nn.Sequential(
                nn.Conv2d(256, 256, c.kernel_size, stride=1, padding=pad), nn.ReLU(inplace=True),
                nn.Conv2d(256, 256, c.kernel_size, stride=1, padding=pad))
            if dev is not None:
                self.conv_head.to(dev)
            elif torch.cuda.is_available():
                self.conv_head.to("cuda")
            # The SAM ViT-H predictor / ClipSeg decoder of sam_model.py:210-222 are prompt-time tools outside the
            # render-and-distill path; they are attached lazily by the caller (self.predictor, self.clipseg).
            self.predictor = None
            self.clipseg = None
        else:
            self.lang_sam = None  # LanguageSAM (sam_model.py:224): 2-D inference utility, not on the hot path

    def get_outputs(self, ray_bundle: RayBundle, get_rgbsigma=True, get_feature=["sam", "dino", "clipseg"], fast=False):  # noqa: B006
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(ray_bundle, density_fns=self.density_fns)
        ray_samples_list.append(ray_samples)
        nerfacto_field_outputs, outputs, weights = self._get_outputs_nerfacto(ray_samples, fast=fast)
        weights_list.append(weights)
        if self.training:
            outputs["weights_list"] = weights_list
            outputs["ray_samples_list"] = ray_samples_list
        if not fast:
            for i in range(self.config.num_proposal_iterations):
                outputs[f"prop_depth_{i}"] = self.renderer_depth(weights=weights_list[i], ray_samples=ray_samples_list[i])
        if self.config.distill_sam and len(get_feature) > 0:
            # The feature heads read the nerfacto branch only through detached tensors (weights, sample bins) and do not
            # depend on each other, so each head is an independent task: when the trainer hands us side streams, a head's
            # forward (and, by autograd's stream rule, its backward) is enqueued on its own stream and runs beside the nerf
            # losses / backward on the main stream.
            fs = self.feature_streams if (self.training and weights.is_cuda) else None
            heads = [h for h in ("sam", "clipseg") if h in get_feature and (h == "sam" or self.config.use_clipseg_feature)]
            selected = self._select_feature_samples(ray_samples, weights) if heads else None
            for h in heads:
                st = fs.get(h) if fs else None
                if st is not None:
                    st.wait_stream(torch.cuda.current_stream())
                    # tensors produced on the main stream and read on the head's stream: tell the caching allocator, so
                    # their memory is not recycled for main-stream work while the head task is still running (the
                    # trainer may already be enqueueing the next step's forward)
                    pos = selected[0].__dict__.get("_positions_cache", {}).get((ops.CONTRACT_L2, False), ())
                    shared = [selected[1], selected[0].ids, *pos]
                    if pos:
                        shared += [w for w, _ in pos[0].__dict__.get("_snf_sorted", {}).values()]
                        gathered = pos[0].__dict__.get("_snf_tp_gathered")
                        if gathered is not None:
                            shared += [gathered] + [w for w, _ in gathered.__dict__.get("_snf_sorted", {}).values()]
                    for t in (weights, ray_samples.euclid_bins, ray_samples.spacing_bins, ray_samples.ray_bundle.origins,
                              ray_samples.ray_bundle.directions, *shared):
                        if t is not None:
                            t.record_stream(st)
                    with torch.cuda.stream(st):
                        self._feature_head(h, selected, outputs)
                else:
                    self._feature_head(h, selected, outputs)
        return outputs

    def _render_program(self):
        """The eval path's static launch schedule (render_program.RenderProgram), or None when this configuration is outside
        its scope / SNF_STATIC_RENDER=0 (the chunk loop then goes through `forward`)."""
        import os
        if os.environ.get("SNF_STATIC_RENDER", "1") != "1" or not self.device.type == "cuda":
            return None
        if self.training:
            # the schedule bakes in eval semantics (near plane 0, no jitter); a model in train() mode goes through `forward` and
            # the collider like the reference's chunk loop (base_model.py:152-175)
            return None
        prog = self.__dict__.get("_render_prog")
        if prog is None:
            from .render_program import RenderProgram
            if RenderProgram.unsupported_reason(self) is not None:
                return None
            prog = self.__dict__["_render_prog"] = RenderProgram(self)
        return prog

    def _select_feature_samples(self, ray_samples: RaySamples, weights):
        """sam_model.py:244-255: top-K by weight, sharpen w^T, renormalise, gather the samples -- shared by the heads, as are
        the contracted positions of the selected samples (computed here once, on the caller's stream)."""
        sam_weights, best_ids = ops.topk_sharpen(weights[..., 0].detach(), self.config.num_sam_samples,
                                                 self.config.sharpening_temperature)
        sam_samples = ray_samples.gather(best_ids)
        if weights.is_cuda:
            from .fields import _positions_of
            u, _, _ = _positions_of(sam_samples, self.sam_field.spatial_distortion, False)  # fills the position cache
            if self.training and torch.is_grad_enabled():
                # the backward sorts of the feature grids depend on these positions and the level geometry only: done
                # here, once per geometry (the SAM and ClipSeg grids have the same two), while the GPU is still lightly
                # loaded, instead of four times inside the backward of the heads
                heads = [list(self.sam_field.clip_encs)] + ([list(self.sam_field.clipseg_encs)]
                                                           if self.config.use_clipseg_feature else [])
                layouts = [ops.table_parallel_layout(tuple(e.spec for e in h)) for h in heads]
                for h, layout in zip(heads, layouts):
                    if layout is not None:
                        # table-parallel head: positions of all ranks gathered once, sorted for the levels owned here
                        ops.tp_presort(u, tuple(e.spec for e in h), layout)
                    else:
                        for enc in h:
                            ops.hashgrid_presort(u, enc.scalings, enc.n_levels, enc.log2_hashmap_size)
        return sam_samples, sam_weights[..., None]

    def _feature_head(self, head: str, selected, outputs) -> None:
        """One head of samnerf/sam_model.py:243-277 (SAMField head, weighted mean, conv head for 'sam')."""
        sam_samples, sam_weights = selected
        field_out = self.sam_field.get_outputs(sam_samples, get_feautre=[head])
        feat_out = self.renderer_mean(embeds=field_out[head], weights=sam_weights.detach())
        if head == "sam" and self.config.patch_size > 1:
            c0, c1 = self.conv_head[0], self.conv_head[2]
            if feat_out.is_cuda:
                # Conv2d -> ReLU -> Conv2d -> patch mean as GEMMs on channel-last rows (csrc/patchconv.hip)
                feat_out = ops.conv_head(feat_out, c0.weight, c0.bias, c1.weight, c1.bias, self.config.patch_size)
            else:
                raise RuntimeError("SAMModel: the conv head runs on the HIP kernels only (no CPU path)")
        outputs[head] = feat_out

    def _get_outputs_nerfacto(self, ray_samples: RaySamples, fast=False):
        field_outputs = self.field(ray_samples, compute_normals=self.config.predict_normals)
        weights = ray_samples.get_weights(field_outputs[FieldHeadNames.DENSITY])
        rgb = self.renderer_rgb(rgb=field_outputs[FieldHeadNames.RGB], weights=weights)
        depth = self.renderer_depth(weights=weights, ray_samples=ray_samples)
        if not fast:
            outputs = {"rgb": rgb, "accumulation": self.renderer_accumulation(weights=weights), "depth": depth}
        else:
            outputs = {"rgb": rgb, "depth": depth}
        return field_outputs, outputs, weights

    def get_loss_dict(self, outputs, batch, metrics_dict=None):
        loss_dict = super().get_loss_dict(outputs, batch, metrics_dict)
        if self.training and self.config.distill_sam:
            fs = self.feature_streams if outputs["sam"].is_cuda else None
            for head, key, wgt in (("sam", "sam_loss", self.config.sam_loss_weight),
                                   ("clipseg", "clipseg_loss", self.config.clipseg_loss_weight)):
                if head == "clipseg" and not self.config.use_clipseg_feature:
                    continue
                st = fs.get(head) if fs else None
                if st is not None:
                    batch[head].record_stream(st)
                    with torch.cuda.stream(st):  # stays on the head's stream
                        loss_dict[key] = ops.rowmse_nanmean_loss(outputs[head], batch[head], wgt)
                else:
                    loss_dict[key] = ops.rowmse_nanmean_loss(outputs[head], batch[head], wgt)
        return loss_dict

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle, points=None, intrin=None, c2w=None,
                                          text_prompt=None, topk=5, thresh=0.5, fast=False) -> Dict[str, torch.Tensor]:
        """samnerf/sam_model.py:337-419, passes 1-3: full-image render in chunks, the SAM feature map from a
        linspace-subsampled [fh*p, fw*p] ray grid regrouped into p x p patches, the 32 x 32 ClipSeg map.  Prompt encoding and
        mask decoding (sam_model.py:420-548) belong to the perception models outside this path: `points` / `text_prompt`
        are accepted and ignored here; the caller feeds outputs['sam'] to SamPredictor.set_feature."""
        num_rays_per_chunk = self.config.eval_num_rays_per_chunk
        image_height, image_width = camera_ray_bundle.origins.shape[:2]
        num_rays = len(camera_ray_bundle)
        outputs_lists: Dict[str, List[torch.Tensor]] = {}

        from . import distributed as D

        prog = self._render_program()

        def run(bundle, granule=1, **kw):
            """Render `bundle` in chunks.  In a data-parallel run the rays of the image are SHARDED over the ranks (contiguous
            shares in whole `granule`s, so p x p patches stay on one rank) and the rows gathered (eval)."""
            n = len(bundle)
            # (every rank takes the same decision: with fewer granules than ranks each rank simply renders everything)
            shard = D.collectives_on() and (n + granule - 1) // granule >= D.world_size()
            lo, hi = D.split_range(n, granule) if shard else (0, n)
            local: Dict[str, List[torch.Tensor]] = {}
            if prog is not None and hi > lo:
                # the chunk loop as a recorded launch schedule (render_program.py): rays read in place, rows written in place
                feats = kw.get("get_feature", [])
                mode = feats[0] if feats else "rgb"
                o = bundle.origins.reshape(-1, 3).contiguous()
                d = bundle.directions.reshape(-1, 3).contiguous()
                res = prog.render(o[lo:hi], d[lo:hi], mode, fast=bool(kw.get("fast", False)), chunk=num_rays_per_chunk)
                for name, rows in res.items():
                    outputs_lists.setdefault(name, []).append(D.all_gather_rows(rows) if shard else rows)
                return
            for i in range(lo, hi, num_rays_per_chunk):
                rb = bundle.get_row_major_sliced_ray_bundle(i, min(i + num_rays_per_chunk, hi))
                rb.nears, rb.fars = None, None
                for name, out in self.forward(ray_bundle=rb, **kw).items():
                    if torch.is_tensor(out):
                        local.setdefault(name, []).append(out)
            for name, lst in local.items():
                rows = torch.cat(lst)
                outputs_lists.setdefault(name, []).append(D.all_gather_rows(rows) if shard else rows)

        feature_h = feature_w = feature_h_clipseg = feature_w_clipseg = None
        from . import render_program as _rp
        if (prog is not None and self.config.distill_sam and _rp.REUSE_PASS1 and not D.collectives_on() and num_rays > 0):
            # One rank, recorded schedule: the feature passes' rays are an index subset of the camera's rays
            # (sam_model.py:371-377,392-398) and eval sampling is deterministic, so pass 1 keeps the selected samples of those
            # rays while their chunk is resident and passes 2-3 run the heads only (render_program.RenderProgram.render_heads).
            from .sam_utils import get_feature_size
            feature_h, feature_w = get_feature_size(image_height, image_width)
            p = self.config.patch_size
            H, W = image_height, image_width
            collect = {"sam": (("sam", H, W, feature_h, feature_w, p),
                               lambda: feature_pixel_ids(H, W, feature_h, feature_w, p))}
            if self.config.use_clipseg_feature:
                feature_h_clipseg, feature_w_clipseg = 32, 32
                collect["clipseg"] = (("clipseg", H, W, 32, 32), lambda: feature_pixel_ids(H, W, 32, 32, 1))
            o = camera_ray_bundle.origins.reshape(-1, 3).contiguous()
            d = camera_ray_bundle.directions.reshape(-1, 3).contiguous()
            outputs = {name: rows.view(image_height, image_width, -1)
                       for name, rows in prog.render(o, d, "rgb", fast=bool(fast), chunk=num_rays_per_chunk, collect=collect).items()}
            outputs["sam"] = prog.render_heads("sam", chunk=num_rays_per_chunk)["sam"].view(feature_h, feature_w, -1)
            if self.config.use_clipseg_feature:
                outputs["clipseg"] = prog.render_heads("clipseg", chunk=num_rays_per_chunk)["clipseg"].view(32, 32, -1)
            return outputs
        run(camera_ray_bundle, get_feature=[], fast=fast)
        sz = camera_ray_bundle.shape
        if self.config.distill_sam:
            from .sam_utils import get_feature_size
            feature_h, feature_w = get_feature_size(image_height, image_width)
            p = self.config.patch_size
            dev = camera_ray_bundle.origins.device
            fb = feature_ray_bundle(camera_ray_bundle, feature_h, feature_w, p)
            saved = {k: outputs_lists.pop(k) for k in list(outputs_lists)}
            run(fb, granule=p * p, get_feature=["sam"])
            sam_list = outputs_lists.get("sam", [])
            outputs_lists.clear()
            outputs_lists.update(saved)
            outputs_lists["sam"] = sam_list
            if self.config.use_clipseg_feature:
                feature_h_clipseg, feature_w_clipseg = 32, 32
                cb = clipseg_ray_bundle(camera_ray_bundle, feature_h_clipseg, feature_w_clipseg)
                saved = {k: outputs_lists.pop(k) for k in list(outputs_lists)}
                run(cb, get_feature=["clipseg"])
                clip_list = outputs_lists.get("clipseg", [])
                outputs_lists.clear()
                outputs_lists.update(saved)
                outputs_lists["clipseg"] = clip_list
        outputs = {}
        for name, lst in outputs_lists.items():
            if name == "sam":
                outputs[name] = torch.cat(lst).view(feature_h, feature_w, -1)
            elif name == "clipseg":
                outputs[name] = torch.cat(lst).view(feature_h_clipseg, feature_w_clipseg, -1)
            else:
                outputs[name] = torch.cat(lst).view(image_height, image_width, -1)
        return outputs

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        param_groups = super().get_param_groups()
        if self.config.distill_sam:
            param_groups["sam_field"] = list(self.sam_field.parameters())
            param_groups["conv"] = list(self.conv_head.parameters())
        return param_groups
