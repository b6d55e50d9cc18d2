"""The batch builder in front of the hot path, with the reference's class names.

    PixelSampler / PatchPixelSampler   nerfstudio/data/pixel_samplers.py:27-300
    Cameras (pinhole) / RayGenerator   nerfstudio/cameras/cameras.py:284-311,576-722, model_components/ray_generators.py:27-63
    FeatureDataloader                  samnerf/data/feature_loader.py:13-56
    SAMDataManager.next_train          samnerf/datamanager.py:97-117
    NerfstudioDataParser (subset)      nerfstudio/data/dataparsers/nerfstudio_dataparser.py:83-300 (+ camera_utils.py:404-487)

Images, SAM / ClipSeg feature maps and cameras are loaded ONCE (host I/O, numpy / PIL / torch.load) and stay resident in
HBM; a training batch is then three HIP launches (`snf_pixel_indices`, `snf_generate_rays`, `snf_gather_nearest` x2-3)
driven by one device-side uniform draw -- the reference does the same work with ~15 small torch launches on partly
host-resident data.  On-disk formats: `transforms*.json` (keys w, h, fl_x, fl_y, cx, cy, frames[{file_path,
transform_matrix}]), SAM `sam_features/<stem>.npy` fp32 [256, fh, 64], ClipSeg `clipseg_features/<stem>.pt` dict with
`activations`: 3 x [1025, 1, 64]."""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Type

import numpy as np
import torch

from . import ops
from .model import InstantiateConfig
from .rays import RayBundle


# ---------------------------------------------------------------------------------------------
# cameras + ray generation
# ---------------------------------------------------------------------------------------------
class Cameras:
    """Pinhole cameras (CameraType.PERSPECTIVE, no distortion): camera_to_worlds [N,3,4], fx/fy/cx/cy [N], one image size."""

    def __init__(self, camera_to_worlds: torch.Tensor, fx, fy, cx, cy, width: int, height: int):
        c2w = torch.as_tensor(camera_to_worlds, dtype=torch.float32)
        if c2w.dim() == 2:
            c2w = c2w[None]
        n = c2w.shape[0]
        self.camera_to_worlds = c2w[:, :3, :4].contiguous()
        vec = lambda v: torch.as_tensor(v, dtype=torch.float32).reshape(-1).expand(n).contiguous()  # noqa: E731
        self.fx, self.fy, self.cx, self.cy = vec(fx), vec(fy), vec(cx), vec(cy)
        self.width, self.height = int(width), int(height)

    def __len__(self) -> int:
        return self.camera_to_worlds.shape[0]

    @property
    def device(self):
        return self.camera_to_worlds.device

    def to(self, device) -> "Cameras":
        c = Cameras(self.camera_to_worlds.to(device), self.fx.to(device), self.fy.to(device), self.cx.to(device),
                    self.cy.to(device), self.width, self.height)
        return c

    def intrinsics(self) -> torch.Tensor:
        """[N,4] = fx, fy, cx, cy (the layout `snf_generate_rays` reads)."""
        return torch.stack([self.fx, self.fy, self.cx, self.cy], dim=-1).contiguous()

    def get_image_coords(self, pixel_offset: float = 0.5) -> torch.Tensor:
        ys, xs = torch.meshgrid(torch.arange(self.height, device=self.device), torch.arange(self.width, device=self.device),
                                indexing="ij")
        return torch.stack([ys, xs], dim=-1) + pixel_offset

    def generate_rays_for_indices(self, ray_indices: torch.Tensor) -> RayBundle:
        """(camera, row, col) int64 [R,3] -> RayBundle; rows/cols are pixel indices (pixel centres are +0.5)."""
        o, d, pa, ci = ops.generate_rays(ray_indices, self.camera_to_worlds, self.intrinsics())
        return RayBundle(origins=o, directions=d, pixel_area=pa, camera_indices=ci)

    def generate_rays(self, camera_indices: int, keep_shape: bool = True) -> RayBundle:
        """All rays of one camera as an [H, W] bundle (the eval / render entry of cameras.py:313-460)."""
        H, W = self.height, self.width
        ys, xs = torch.meshgrid(torch.arange(H, device=self.device), torch.arange(W, device=self.device), indexing="ij")
        idx = torch.stack([torch.full_like(ys, int(camera_indices)), ys, xs], dim=-1).reshape(-1, 3)
        rb = self.generate_rays_for_indices(idx)
        return rb.reshape((H, W)) if keep_shape else rb


class RayGenerator(torch.nn.Module):
    """ray_generators.py:27-63 with the camera optimizer off (samnerf/samconfigs.py:74,128)."""

    def __init__(self, cameras: Cameras, pose_optimizer=None) -> None:
        super().__init__()
        if pose_optimizer is not None and getattr(getattr(pose_optimizer, "config", None), "mode", "off") != "off":
            raise NotImplementedError("camera optimization is off in the samnerf configs")
        self.cameras = cameras

    def forward(self, ray_indices: torch.Tensor) -> RayBundle:
        return self.cameras.generate_rays_for_indices(ray_indices)


# ---------------------------------------------------------------------------------------------
# pixel samplers
# ---------------------------------------------------------------------------------------------
class PixelSampler:
    """pixel_samplers.py:27-203 for tensor image batches without masks."""

    def __init__(self, num_rays_per_batch: int, keep_full_image: bool = False, generator: Optional[torch.Generator] = None,
                 **kwargs) -> None:
        self.kwargs = kwargs
        self.num_rays_per_batch = num_rays_per_batch
        self.keep_full_image = keep_full_image
        self.generator = generator
        self.patch_size = 1

    def set_num_rays_per_batch(self, num_rays_per_batch: int) -> None:
        self.num_rays_per_batch = num_rays_per_batch

    def _draw(self, n: int, device) -> torch.Tensor:
        return torch.rand((n, 3), device=device, generator=self.generator)

    def sample_method(self, batch_size: int, num_images: int, image_height: int, image_width: int, mask=None,
                      device="cuda", u: Optional[torch.Tensor] = None) -> torch.Tensor:
        if mask is not None:
            raise NotImplementedError("masked sampling is not used by the samnerf configs")
        p = self.patch_size
        u = self._draw(batch_size // (p * p), device) if u is None else u
        return ops.pixel_indices(u, batch_size, p, num_images, image_height, image_width)

    def sample(self, image_batch: Dict) -> Dict:
        """collate_image_dataset_batch: {"image" [N,H,W,3], "image_idx" [N]} -> {"image" [R,3], "indices" [R,3]}."""
        images = image_batch["image"]
        if not torch.is_tensor(images):
            raise NotImplementedError("list image batches (images of different sizes) are not supported")
        n, H, W, _ = images.shape
        indices = self.sample_method(self.num_rays_per_batch, n, H, W, device=images.device)
        out = {"image": ops.gather_nearest(indices, images, (H, W))}
        idx = image_batch.get("image_idx")
        if idx is not None and not torch.equal(idx.to(indices.device), torch.arange(n, device=indices.device)):
            indices = indices.clone()
            indices[:, 0] = idx.to(indices.device)[indices[:, 0]]
        out["indices"] = indices
        if self.keep_full_image:
            out["full_image"] = images
        return out


class PatchPixelSampler(PixelSampler):
    """pixel_samplers.py:246-300: random p x p patches (top-left corner U * (H - p, W - p))."""

    def __init__(self, num_rays_per_batch: int, keep_full_image: bool = False, **kwargs) -> None:
        patch_size = kwargs["patch_size"]
        num_rays = (num_rays_per_batch // (patch_size ** 2)) * (patch_size ** 2)
        super().__init__(num_rays, keep_full_image, **kwargs)
        self.patch_size = patch_size


# ---------------------------------------------------------------------------------------------
# feature maps
# ---------------------------------------------------------------------------------------------
def clipseg_activations_to_map(x: Dict) -> torch.Tensor:
    """samnerf/datamanager.py:90-94: cat the three activation tensors, drop the CLS token, 32 x 32 grid."""
    return torch.cat(x["activations"], dim=-1).squeeze()[1:, ...].reshape(512 // 16, 512 // 16, -1)


class FeatureDataloader:
    """samnerf/data/feature_loader.py:13-56.  `npy_paths` may also be an already stacked [N, fh, fw, C] tensor."""

    def __init__(self, device, npy_paths, image_shape: Sequence[int], patch_size: int = 1,
                 get_feature: Callable = lambda x: x):
        self.device = device
        self.npy_path = npy_paths
        self.image_shape = list(image_shape)
        self.patch_size = patch_size
        if torch.is_tensor(npy_paths):
            feats = npy_paths
        elif npy_paths[0].endswith(".npy"):
            feats = torch.from_numpy(np.stack([np.transpose(np.load(p), (1, 2, 0)) for p in npy_paths], axis=0))  # n h w c
        else:
            assert npy_paths[0].endswith(".pt")
            feats = torch.stack([get_feature(torch.load(p, map_location="cpu")) for p in npy_paths], dim=0)
        self.features = feats.to(device=device, dtype=torch.float32).contiguous()

    def __call__(self, img_points: torch.Tensor, point_stride: int = 1, point_offset: int = 0) -> torch.Tensor:
        """img_points [B,3] (img_ind, row, col) -> [B, C]; with a stride/offset only every stride-th point is looked up
        (the patch centres of datamanager.py:106-110 without materialising `center_indices`)."""
        return ops.gather_nearest(img_points, self.features, self.image_shape, point_stride, point_offset)


# ---------------------------------------------------------------------------------------------
# dataparser subset: transforms.json -> cameras (poses oriented "up", centred, scaled) + file names
# ---------------------------------------------------------------------------------------------
def rotation_matrix(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """camera_utils.py:404-429 (rotation taking a to b)."""
    a, b = a / torch.linalg.norm(a), b / torch.linalg.norm(b)
    v, c = torch.linalg.cross(a, b), torch.dot(a, b)
    if c < -1 + 1e-8:
        return rotation_matrix(a + (torch.rand(3) - 0.5) * 0.01, b)
    s = torch.linalg.norm(v)
    k = torch.tensor([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    return torch.eye(3) + k + k @ k * ((1 - c) / (s ** 2 + 1e-8))


def auto_orient_and_center_poses(poses: torch.Tensor, method: str = "up", center_poses: bool = True):
    """camera_utils.py:432-487 for method in {"up", "none"}: poses [N,4,4] -> ([N,3,4], transform [3,4])."""
    translation = poses[..., :3, 3]
    mean_translation = torch.mean(translation, dim=0)
    translation = mean_translation if center_poses else torch.zeros_like(mean_translation)
    if method == "up":
        up = torch.mean(poses[:, :3, 1], dim=0)
        up = up / torch.linalg.norm(up)
        rotation = rotation_matrix(up, torch.tensor([0.0, 0.0, 1.0]))
        transform = torch.cat([rotation, rotation @ -translation[..., None]], dim=-1)
    elif method == "none":
        transform = torch.eye(4)
        transform[:3, 3] = -translation
        transform = transform[:3, :]
    else:
        raise NotImplementedError("orientation_method 'pca' is not supported")
    return transform @ poses, transform


@dataclass
class NerfstudioDataParserConfig(InstantiateConfig):
    """nerfstudio_dataparser.py:44-68 (fields the samnerf configs touch)."""
    _target: Type = field(default_factory=lambda: NerfstudioDataParser)
    data: str = "data/nerfstudio/poster"
    scale_factor: float = 1.0
    scene_scale: float = 1.0
    orientation_method: str = "up"
    center_poses: bool = True
    auto_scale_poses: bool = True
    train_val_json_split: bool = False


@dataclass
class DataparserOutputs:
    image_filenames: List[str]
    cameras: Cameras
    dataparser_scale: float = 1.0
    dataparser_transform: Optional[torch.Tensor] = None


class NerfstudioDataParser:
    """transforms{,_train,_test}.json -> DataparserOutputs (pinhole, one image size, no masks / depth / distortion)."""

    def __init__(self, config: NerfstudioDataParserConfig):
        self.config = config

    def get_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        c = self.config
        data = str(c.data)
        name = f"transforms_{split}.json" if c.train_val_json_split else "transforms.json"
        path = data if data.endswith(".json") else os.path.join(data, name)
        meta = json.load(open(path))
        root = os.path.dirname(path)
        frames = meta["frames"]
        if not c.train_val_json_split and "transforms_train" not in os.path.basename(path):
            pass  # the 90/10 index split of nerfstudio_dataparser.py:196-214 applies only to single-json datasets: use all
        files = [os.path.join(root, f["file_path"]) for f in frames]
        poses = torch.from_numpy(np.array([f["transform_matrix"] for f in frames], dtype=np.float32))
        poses, transform = auto_orient_and_center_poses(poses, c.orientation_method, c.center_poses)
        scale = 1.0
        if c.auto_scale_poses:
            scale /= float(torch.max(torch.abs(poses[:, :3, 3])))
        scale *= c.scale_factor
        poses[:, :3, 3] *= scale
        get = lambda k: (torch.tensor([float(f[k]) for f in frames]) if k in frames[0] else float(meta[k]))  # noqa: E731
        cams = Cameras(poses[:, :3, :4], get("fl_x"), get("fl_y"), get("cx"), get("cy"), int(meta["w"]), int(meta["h"]))
        return DataparserOutputs(files, cams, scale, transform)


def load_images(filenames: Sequence[str]) -> torch.Tensor:
    """[N,H,W,3] float32 in [0,1] (InputDataset.get_image: uint8 / 255, alpha dropped)."""
    from PIL import Image
    out = []
    for f in filenames:
        if not os.path.exists(f):
            for ext in (".png", ".jpg", ".jpeg", ".JPG"):
                if os.path.exists(f + ext):
                    f = f + ext
                    break
        img = np.asarray(Image.open(f).convert("RGB"), dtype=np.uint8)
        out.append(torch.from_numpy(img.astype(np.float32) / 255.0))
    return torch.stack(out, dim=0)


# ---------------------------------------------------------------------------------------------
# datamanager on real data
# ---------------------------------------------------------------------------------------------
@dataclass
class DiskSAMDataManagerConfig(InstantiateConfig):
    """samnerf/datamanager.py:23-30 + the VanillaDataManagerConfig fields the samnerf configs set."""
    _target: Type = field(default_factory=lambda: DiskSAMDataManager)
    dataparser: NerfstudioDataParserConfig = field(default_factory=NerfstudioDataParserConfig)
    train_num_rays_per_batch: int = 4096 * 4
    eval_num_rays_per_batch: int = 4096 * 4
    patch_size: int = 1
    distill_sam: bool = True
    use_dino_feature: bool = False
    use_clipseg_feature: bool = False
    seed: int = 0


def _feature_paths(image_filenames: Sequence[str], folder: str, ext: str) -> List[str]:
    """datamanager.py:50-55,75-81: <data>/<folder>/<image stem><ext> next to the images' parent directory."""
    return [os.path.join(os.path.dirname(os.path.dirname(n)), folder, os.path.basename(n).split(".")[0] + ext)
            for n in image_filenames]


class DiskSAMDataManager:
    """SAMDataManager (samnerf/datamanager.py:33-117) with everything resident on the device."""

    def __init__(self, config: DiskSAMDataManagerConfig, device="cuda", test_mode="val", world_size: int = 1,
                 local_rank: int = 0, images: Optional[torch.Tensor] = None, cameras: Optional[Cameras] = None,
                 sam_features: Optional[torch.Tensor] = None, clipseg_features: Optional[torch.Tensor] = None, **kw):
        if config.use_dino_feature:
            raise NotImplementedError("DINO features are off in the samnerf configs")
        self.config = config
        self.device = torch.device(device)
        self.train_count = 0
        if images is None:  # load from disk
            outs = config.dataparser.setup().get_dataparser_outputs("train")
            self.train_dataparser_outputs = outs
            images, cameras = load_images(outs.image_filenames), outs.cameras
            if config.distill_sam and sam_features is None:
                sam_features = FeatureDataloader("cpu", _feature_paths(outs.image_filenames, "sam_features", ".npy"),
                                                 images.shape[1:3]).features
            if config.use_clipseg_feature and clipseg_features is None:
                clipseg_features = FeatureDataloader("cpu", _feature_paths(outs.image_filenames, "clipseg_features", ".pt"),
                                                     images.shape[1:3], get_feature=clipseg_activations_to_map).features
        self.images = images.to(self.device, torch.float32).contiguous()
        self.cameras = cameras.to(self.device)
        self.num_train_data = self.images.shape[0]
        H, W = self.images.shape[1:3]
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(config.seed + local_rank)  # per-rank ray draws (samnerf/train.py:87)
        if config.patch_size > 1:
            self.train_pixel_sampler = PatchPixelSampler(config.train_num_rays_per_batch, patch_size=config.patch_size,
                                                         generator=self.gen)
        else:
            self.train_pixel_sampler = PixelSampler(config.train_num_rays_per_batch, generator=self.gen)
        self.train_ray_generator = RayGenerator(self.cameras)
        self.sam_loader = self.clipseg_loader = None
        if config.distill_sam:
            self.sam_loader = FeatureDataloader(self.device, sam_features, [H, W], patch_size=config.patch_size)
        if config.use_clipseg_feature:
            self.clipseg_loader = FeatureDataloader(self.device, clipseg_features, [H, W], patch_size=1)

    def next_train(self, step: int) -> Tuple[RayBundle, Dict]:
        self.train_count += 1
        batch = self.train_pixel_sampler.sample({"image": self.images})
        ray_indices = batch["indices"]
        ray_bundle = self.train_ray_generator(ray_indices)
        p = self.config.patch_size
        if self.config.distill_sam:
            batch["sam"] = self.sam_loader(ray_indices, point_stride=p * p, point_offset=(p // 2) * p + p // 2)
        if self.config.use_clipseg_feature:
            batch["clipseg"] = self.clipseg_loader(ray_indices)
        return ray_bundle, batch

    def get_param_groups(self) -> Dict[str, List]:
        return {}
