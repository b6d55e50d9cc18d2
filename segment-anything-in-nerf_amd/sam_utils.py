"""samnerf/sam_utils.py:7-14."""
import math


def get_feature_size(h: int, w: int, largesize: int = 64):
    """Feature-map size with the long image side mapped to `largesize` (the reference leaves h == w undefined; a square
    image maps to largesize x largesize here)."""
    if h < w:
        return int(math.ceil((h / w) * largesize)), largesize
    if h > w:
        return largesize, int(math.ceil((w / h) * largesize))
    return largesize, largesize


def set_feature(feature, original_image_size, img_size: int = 1024):
    """SamPredictor.set_feature (samnerf/segment_anything/predictor.py:100-127) without the predictor object: the rendered
    SAM feature map [C, fh, fw] (sam_model.py:486 passes outputs["sam"].permute(2, 0, 1)) becomes the [1, C, S, S] embedding
    the mask decoder expects -- zero rows appended below a landscape map (h < w), zero columns to the right of a portrait one.
    Returns (features [1, C, S, S], input_size (h, w) of the resized image in the encoder's img_size frame).

    The reference's portrait branch concatenates its [1, C, h, h - w] zero block along dim 2, which torch rejects unless
    w == h / 2; the evident intent (square map, data in the top-left corner) is implemented here (dim 3)."""
    import numpy as np
    import torch
    h, w = original_image_size
    if h <= w:
        input_size = (int(math.ceil(h / w * img_size)), img_size)
    else:
        input_size = (img_size, int(math.ceil(w / h * img_size)))
    feats = torch.from_numpy(feature) if isinstance(feature, np.ndarray) else feature
    assert isinstance(feats, torch.Tensor) and feats.dim() == 3
    feats = feats.unsqueeze(0)
    c, fh, fw = feats.shape[-3:]
    if fh < fw:
        feats = torch.cat([feats, feats.new_zeros((1, c, fw - fh, fw))], dim=2)
    elif fh > fw:
        feats = torch.cat([feats, feats.new_zeros((1, c, fh, fh - fw))], dim=3)
    return feats, input_size


PIXEL_MEAN = (123.675, 116.28, 103.53)  # build_sam.py:99-100
PIXEL_STD = (58.395, 57.12, 57.375)


def crop_embedding(feature, original_image_size):
    """The offline target writer's crop (samnerf/preprocessing/get_image_embeddings.py:29-35): the encoder sees the image
    zero-padded to a square, so the rows (landscape) / columns (portrait) of its [.., fs, fs] map that lie in the padding are cut
    off -- ceil(h / w * fs) rows or ceil(w / h * fs) columns stay.  The result, squeezed, is what the `.npy` files hold
    ([256, ceil(h / w * 64), 64] for a landscape image) and what `set_feature` pads back."""
    h, w = original_image_size
    fs = feature.shape[-1]
    if h < w:
        return feature[..., :int(math.ceil((h / w) * fs)), :]
    if h > w:
        return feature[..., :, :int(math.ceil((w / h) * fs))]
    return feature


class SamImageEmbedder:
    """The part of SamPredictor that produces the distillation targets (samnerf/segment_anything/predictor.py:70-97 +
    modeling/sam.py:164-174 + get_image_embeddings.py:23-61): a long-side-`img_size` image -> normalise + zero-pad on the device
    (snf_sam_preprocess) -> ImageEncoderViT -> [1, 256, 64, 64] features; `embedding()` crops them the way the offline writer
    does.  Resizing the camera image to the long side (ResizeLongestSide, a PIL / torchvision resize) stays with the caller, as
    `set_torch_image` expects; the prompt encoder and mask decoder are outside this path."""

    def __init__(self, image_encoder, pixel_mean=PIXEL_MEAN, pixel_std=PIXEL_STD):
        import torch
        self.image_encoder = image_encoder
        dev = next(image_encoder.parameters()).device
        self.pixel_mean = torch.tensor(pixel_mean, dtype=torch.float32, device=dev)
        self.pixel_std = torch.tensor(pixel_std, dtype=torch.float32, device=dev)
        self.features = None
        self.original_size = self.input_size = None

    def preprocess(self, x):
        from . import ops
        return ops.sam_preprocess(x, self.pixel_mean, self.pixel_std, self.image_encoder.img_size)

    def set_torch_image(self, transformed_image, original_image_size):
        """predictor.py:70-97: `transformed_image` [B, 3, H, W] (uint8 or float, RGB, long side == img_size)."""
        S = self.image_encoder.img_size
        assert (len(transformed_image.shape) == 4 and transformed_image.shape[1] == 3
                and max(*transformed_image.shape[2:]) == S), f"set_torch_image input must be BCHW with long side {S}."
        self.original_size = tuple(original_image_size)
        self.input_size = tuple(transformed_image.shape[-2:])
        self.features = self.image_encoder(self.preprocess(transformed_image.to(self.pixel_mean.device)))
        return self.features

    def embedding(self):
        """What get_image_embeddings.py saves for the image last set: the features without their padding rows / columns."""
        assert self.features is not None, "set_torch_image first"
        return crop_embedding(self.features, self.original_size).squeeze()

    def save(self, path: str) -> None:
        """Write the image's distillation target as get_image_embeddings.py:57-60 does: `np.save(path, feature.squeeze())`, fp32,
        [C, rows, cols] without the padding -- the file `FeatureDataloader` / `set_feature` read back."""
        import numpy as np
        np.save(path, self.embedding().detach().float().cpu().numpy())
