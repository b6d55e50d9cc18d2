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
