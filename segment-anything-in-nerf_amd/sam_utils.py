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
