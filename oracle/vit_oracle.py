"""CPU ORACLE for the SAM image encoder (ViT) forward -- SURVEY.md 8(f) rank 3.  TEST INFRASTRUCTURE ONLY.

Plain-torch restatement of samnerf/segment_anything/modeling/image_encoder.py (+ common.py:13-43), operating on a
{state_dict key: tensor} map with the reference's own key names, so a SAM checkpoint's `image_encoder.*` entries drop in.
Pinned by tests/golden/make_golden.py::fx_vit, which runs the reference's ImageEncoderViT (loaded from /root/reference in the
build container) on a small configuration and asserts agreement with this file before writing the fixture.
Only tests/ and __graft_entry__.smoke() may import it."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class ViTConfig:
    """Constructor arguments of ImageEncoderViT (image_encoder.py:17-36); defaults = build_sam_vit_h (build_sam.py:14-22,53-80)."""
    img_size: int = 1024
    patch_size: int = 16
    in_chans: int = 3
    embed_dim: int = 1280
    depth: int = 32
    num_heads: int = 16
    mlp_ratio: float = 4.0
    out_chans: int = 256
    window_size: int = 14
    global_attn_indexes: Tuple[int, ...] = (7, 15, 23, 31)
    use_rel_pos: bool = True
    ln_eps: float = 1e-6  # norm_layer = partial(LayerNorm, eps=1e-6) (build_sam.py:66); LayerNorm2d default (common.py:32)

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size


def init_weights(cfg: ViTConfig, seed: int = 0, scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Random weights under the reference's state_dict names (rel-pos tables random too: zeros would hide them)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, k=1.0: torch.randn(*s, generator=g) * k * scale  # noqa: E731
    C, hd, G = cfg.embed_dim, cfg.embed_dim // cfg.num_heads, cfg.grid
    M = int(C * cfg.mlp_ratio)
    sd = {"pos_embed": r(1, G, G, C, k=0.02),
          "patch_embed.proj.weight": r(C, cfg.in_chans, cfg.patch_size, cfg.patch_size,
                                       k=(cfg.in_chans * cfg.patch_size ** 2) ** -0.5),
          "patch_embed.proj.bias": r(C, k=0.02)}
    for i in range(cfg.depth):
        ws = cfg.window_size if i not in cfg.global_attn_indexes else 0
        n = ws if ws > 0 else G
        p = f"blocks.{i}."
        sd.update({p + "norm1.weight": 1 + r(C, k=0.1), p + "norm1.bias": r(C, k=0.1),
                   p + "attn.qkv.weight": r(3 * C, C, k=C ** -0.5), p + "attn.qkv.bias": r(3 * C, k=0.02),
                   p + "attn.proj.weight": r(C, C, k=C ** -0.5), p + "attn.proj.bias": r(C, k=0.02),
                   p + "norm2.weight": 1 + r(C, k=0.1), p + "norm2.bias": r(C, k=0.1),
                   p + "mlp.lin1.weight": r(M, C, k=C ** -0.5), p + "mlp.lin1.bias": r(M, k=0.02),
                   p + "mlp.lin2.weight": r(C, M, k=M ** -0.5), p + "mlp.lin2.bias": r(C, k=0.02)})
        if cfg.use_rel_pos:
            sd[p + "attn.rel_pos_h"] = r(2 * n - 1, hd, k=0.1)
            sd[p + "attn.rel_pos_w"] = r(2 * n - 1, hd, k=0.1)
    O = cfg.out_chans
    sd.update({"neck.0.weight": r(O, C, 1, 1, k=C ** -0.5), "neck.1.weight": 1 + r(O, k=0.1), "neck.1.bias": r(O, k=0.1),
               "neck.2.weight": r(O, O, 3, 3, k=(9 * O) ** -0.5), "neck.3.weight": 1 + r(O, k=0.1),
               "neck.3.bias": r(O, k=0.1)})
    return sd


def window_partition(x: torch.Tensor, ws: int):
    """image_encoder.py:239-261."""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph > 0 or pw > 0:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(windows: torch.Tensor, ws: int, pad_hw, hw) -> torch.Tensor:
    """image_encoder.py:264-287."""
    (Hp, Wp), (H, W) = pad_hw, hw
    B = windows.shape[0] // (Hp * Wp // ws // ws)
    x = windows.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous() if (Hp > H or Wp > W) else x


def get_rel_pos(q_size: int, k_size: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """image_encoder.py:290-320 (the table already has 2*max(q,k)-1 rows in every SAM configuration: no interpolation)."""
    assert rel_pos.shape[0] == 2 * max(q_size, k_size) - 1
    q = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q - k) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel_pos[rel.long()]


def attention(sd, p: str, x: torch.Tensor, num_heads: int, use_rel_pos: bool) -> torch.Tensor:
    """Attention.forward + add_decomposed_rel_pos (image_encoder.py:185-236,323-361).  x [B,H,W,C]."""
    B, H, W, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B, H * W, 3, num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * num_heads, H * W, -1).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    if use_rel_pos:
        Rh, Rw = get_rel_pos(H, H, sd[p + "rel_pos_h"]), get_rel_pos(W, W, sd[p + "rel_pos_w"])
        rq = q.reshape(B * num_heads, H, W, hd)
        rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
        rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
        attn = (attn.view(-1, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(-1, H * W, H * W)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).view(B, num_heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])


def block(sd, i: int, x: torch.Tensor, cfg: ViTConfig) -> torch.Tensor:
    """Block.forward (image_encoder.py:164-182)."""
    p = f"blocks.{i}."
    C = x.shape[-1]
    ws = cfg.window_size if i not in cfg.global_attn_indexes else 0
    shortcut = x
    x = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.ln_eps)
    if ws > 0:
        H, W = x.shape[1], x.shape[2]
        x, pad_hw = window_partition(x, ws)
    x = attention(sd, p + "attn.", x, cfg.num_heads, cfg.use_rel_pos)
    if ws > 0:
        x = window_unpartition(x, ws, pad_hw, (H, W))
    x = shortcut + x
    y = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.ln_eps)
    y = F.linear(F.gelu(F.linear(y, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"])), sd[p + "mlp.lin2.weight"],
                 sd[p + "mlp.lin2.bias"])
    return x + y


def layer_norm_2d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    """common.py:31-43 (normalises over the channel axis of NCHW)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, cfg: ViTConfig, return_tokens: bool = False):
    """ImageEncoderViT.forward (image_encoder.py:106-117): x [B,3,S,S] -> [B,out_chans,S/16,S/16]."""
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=cfg.patch_size).permute(0, 2, 3, 1)
    t = t + sd["pos_embed"]
    trace = [t]
    for i in range(cfg.depth):
        t = block(sd, i, t, cfg)
        trace.append(t)
    y = F.conv2d(t.permute(0, 3, 1, 2), sd["neck.0.weight"])
    y = layer_norm_2d(y, sd["neck.1.weight"], sd["neck.1.bias"], cfg.ln_eps)
    y = F.conv2d(y, sd["neck.2.weight"], padding=1)
    y = layer_norm_2d(y, sd["neck.3.weight"], sd["neck.3.bias"], cfg.ln_eps)
    return (y, trace) if return_tokens else y


def sam_preprocess(x: torch.Tensor, mean, std, img_size: int) -> torch.Tensor:
    """Sam.preprocess (samnerf/segment_anything/modeling/sam.py:164-174): per-channel (x - mean) / std, then zero-pad right and
    bottom to img_size x img_size.  x [B, 3, h, w], uint8 or float."""
    m = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)
    y = (x - m) / s
    h, w = y.shape[-2:]
    return torch.nn.functional.pad(y, (0, img_size - w, 0, img_size - h))
