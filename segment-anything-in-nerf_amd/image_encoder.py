"""SAM image encoder (ViT) forward on the HIP kernels -- SURVEY.md 8(f) rank 3.

`ImageEncoderViT` keeps the constructor arguments, submodule names and state_dict keys of
samnerf/segment_anything/modeling/image_encoder.py:17-117 (so `sam.image_encoder` checkpoints load with load_state_dict),
but its forward runs: patchify + GEMM (+ pos_embed) -> per block { LayerNorm -> window partition -> qkv GEMM -> decomposed
rel-pos + flash attention on the fp32 matrix cores -> proj GEMM -> window merge + residual -> LayerNorm -> MLP GEMMs with a
GELU epilogue -> residual folded into the next LayerNorm } -> neck (1x1 conv = GEMM, LayerNorm2d, 3x3 conv = unfold + GEMM,
LayerNorm2d).  Tokens stay channel-last rows [B*T, C] throughout; inference only (no autograd)."""
from __future__ import annotations

from functools import partial
from typing import Optional, Tuple, Type

import os

import torch
import torch.nn as nn

from . import ops

_GEMM_LDS = os.environ.get("SNF_VIT_GEMM_LDS", "1") != "0"

# the blocks' GEMMs on operands split by their producers (csrc/gemm_planes.hip); False: every block through the tiled kernel that
# splits fp32 operands itself (tests compare the two)
PLANES_PATH = True


class LayerNorm2d(nn.Module):
    """common.py:31-43 (parameters only; applied on channel-last rows by ops.layernorm)."""

    def __init__(self, num_channels: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps


class MLPBlock(nn.Module):
    def __init__(self, embedding_dim: int, mlp_dim: int, act: Type[nn.Module] = nn.GELU) -> None:
        super().__init__()
        if act is not nn.GELU:
            raise NotImplementedError("the image encoder uses nn.GELU")
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)


class Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = True, use_rel_pos: bool = False,
                 rel_pos_zero_init: bool = True, input_size: Optional[Tuple[int, int]] = None) -> None:
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        if use_rel_pos:
            assert input_size is not None, "Input size must be provided if using relative positional encoding."
            self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
            self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_dim))


class Block(nn.Module):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = True,
                 norm_layer: Type[nn.Module] = nn.LayerNorm, act_layer: Type[nn.Module] = nn.GELU, use_rel_pos: bool = False,
                 rel_pos_zero_init: bool = True, window_size: int = 0, input_size: Optional[Tuple[int, int]] = None) -> None:
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, use_rel_pos=use_rel_pos,
                              rel_pos_zero_init=rel_pos_zero_init,
                              input_size=input_size if window_size == 0 else (window_size, window_size))
        self.norm2 = norm_layer(dim)
        self.mlp = MLPBlock(embedding_dim=dim, mlp_dim=int(dim * mlp_ratio), act=act_layer)
        self.window_size = window_size


class PatchEmbed(nn.Module):
    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans: int = 3, embed_dim: int = 768) -> None:
        super().__init__()
        if tuple(kernel_size) != tuple(stride) or tuple(padding) != (0, 0):
            raise NotImplementedError("non-overlapping patches only (kernel == stride, no padding)")
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)


class ImageEncoderViT(nn.Module):
    def __init__(self, img_size: int = 1024, patch_size: int = 16, in_chans: int = 3, embed_dim: int = 768, depth: int = 12,
                 num_heads: int = 12, mlp_ratio: float = 4.0, out_chans: int = 256, qkv_bias: bool = True,
                 norm_layer: Type[nn.Module] = nn.LayerNorm, act_layer: Type[nn.Module] = nn.GELU, use_abs_pos: bool = True,
                 use_rel_pos: bool = False, rel_pos_zero_init: bool = True, window_size: int = 0,
                 global_attn_indexes: Tuple[int, ...] = ()) -> None:
        super().__init__()
        self.img_size, self.patch_size = img_size, patch_size
        self.patch_embed = PatchEmbed(kernel_size=(patch_size, patch_size), stride=(patch_size, patch_size),
                                      in_chans=in_chans, embed_dim=embed_dim)
        self.pos_embed: Optional[nn.Parameter] = None
        if use_abs_pos:
            self.pos_embed = nn.Parameter(torch.zeros(1, img_size // patch_size, img_size // patch_size, embed_dim))
        self.blocks = nn.ModuleList()
        for i in range(depth):
            self.blocks.append(Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                     norm_layer=norm_layer, act_layer=act_layer, use_rel_pos=use_rel_pos,
                                     rel_pos_zero_init=rel_pos_zero_init,
                                     window_size=window_size if i not in global_attn_indexes else 0,
                                     input_size=(img_size // patch_size, img_size // patch_size)))
        self._wcache, self._pcache = {}, {}  # split weights / operand-plane buffers of the split-operand block path
        self.neck = nn.Sequential(nn.Conv2d(embed_dim, out_chans, kernel_size=1, bias=False), LayerNorm2d(out_chans),
                                  nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False),
                                  LayerNorm2d(out_chans))

    # ---- the blocks on pre-split GEMM operands (csrc/gemm_planes.hip) --------------------------------------------------------
    def _planes_ok(self, blk) -> bool:
        """The split-operand path needs: the bf16-split gemm mode, widths the plane kernels tile (rows held in registers by the
        norm: C a multiple of 256; GEMM outputs multiples of 128; k multiples of 64), heads that start on a k-block."""
        if not PLANES_PATH or int(ops._L().snf_get_gemm_mode()) == 0:  # (exact-fp32 mode: the plain path)
            return False
        C, M = blk.norm1.weight.shape[0], blk.mlp.lin1.weight.shape[0]
        hd = C // blk.attn.num_heads
        return C % 256 == 0 and C // 256 in (1, 2, 3, 4, 5, 6, 8) and M % 64 == 0 and hd % 8 == 0 and hd <= 96

    def _wplanes(self, lin, M: int = 0):
        """bf16 hi / lo planes of a layer's weight, split once and re-split when the parameter is written (load_state_dict).
        M (rows of the product) given: k-blocked planes where 256 x 320 tiles fill the chip in one round -- the GEMM with both
        operands through LDS (ViT-H: attn.qkv, mlp.lin1; SNF_VIT_GEMM_LDS=0: never)."""
        Nc, K = lin.weight.shape
        kb = bool(M) and _GEMM_LDS and K % 64 == 0 and Nc % 320 == 0 and 160 <= -(-M // 256) * (Nc // 320) <= 256
        key, tag = (id(lin.weight), kb), (lin.weight._version, lin.weight.data_ptr(), str(lin.weight.device))
        hit = self._wcache.get(key)
        if hit is None or hit[0] != tag:  # (writes through `.data` bump no version: call reset_weight_cache() after those)
            w = lin.weight.detach().float().contiguous()
            hit = (tag, ops.split_weight_planes_kb(w) if kb else ops.split_weight_planes(w))
            self._wcache[key] = hit
        return hit[1]

    def reset_weight_cache(self) -> None:
        """Forget the split weights (after changing parameters in a way autograd's version counter does not see)."""
        self._wcache.clear()

    # `.data` writes bump no version counter: the two module-level ways parameters are replaced wholesale drop the cache themselves
    def _apply(self, fn, *args, **kw):
        self._wcache.clear()
        self._pcache.clear()
        return super()._apply(fn, *args, **kw)

    def load_state_dict(self, *args, **kw):
        self._wcache.clear()
        return super().load_state_dict(*args, **kw)

    def _load_from_state_dict(self, *args, **kw):
        # (reached when the state is loaded through a PARENT module, e.g. `sam.load_state_dict`, which never calls the override above)
        self._wcache.clear()
        return super()._load_from_state_dict(*args, **kw)

    def _pbuf(self, tag: str, M: int, K: int, device, zero: bool = False):
        key = (tag, M, K, str(device))
        buf = self._pcache.get(key)
        if buf is None:
            buf = self._pcache[key] = ops.Planes.empty(M, K, device, zero=zero)
        return buf

    def _block_planes(self, blk, shortcut, pending, B: int, G: int):
        """Block.forward (image_encoder.py:164-182) with every GEMM operand born split: norm1 writes qkv's operand at the window
        partition's rows, the attention writes proj's, norm2 lin1's, lin1's GELU epilogue lin2's."""
        a, ws, dev = blk.attn, blk.window_size, shortcut.device
        C, Mh = blk.norm1.weight.shape[0], blk.mlp.lin1.weight.shape[0]
        if ws > 0:
            nW = (G + ws - 1) // ws
            n, Bw = ws, B * nW * nW
            y = self._pbuf("n1w", Bw * ws * ws, C, dev, zero=True)  # padded rows: written by nobody, zero for good
            grid = (G, G, ws)
        else:
            n, Bw = G, B
            y = self._pbuf("n1", B * G * G, C, dev)
            grid = None
        if pending is None:
            ops.layernorm_planes(shortcut, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, y, grid=grid)
        else:
            _, shortcut = ops.layernorm_planes(shortcut, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, y, residual=pending,
                                               want_sum=True, grid=grid)
        qkv = ops.linear_planes(y, self._wplanes(a.qkv, y.M), a.qkv.bias)
        o = ops.attention_planes(qkv, Bw, n * n, a.num_heads, n, self._pbuf("att", Bw * n * n, C, dev),
                                 a.rel_pos_h if a.use_rel_pos else None, a.rel_pos_w if a.use_rel_pos else None)
        o = ops.linear_planes(o, self._wplanes(a.proj), a.proj.bias)
        # x = shortcut + window_unpartition(proj) inside norm2 (its residual input, read at the token's window row)
        n2 = self._pbuf("n2", B * G * G, C, dev)
        if ws > 0:
            y2, shortcut = ops.layernorm_planes_merge(shortcut, o, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, n2, B, G, G, ws)
        else:
            y2, shortcut = ops.layernorm_planes(shortcut, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, n2, residual=o, want_sum=True)
        h = ops.linear_planes(y2, self._wplanes(blk.mlp.lin1, y2.M), blk.mlp.lin1.bias, ops.ACT_GELU,
                              out=self._pbuf("h", B * G * G, Mh, dev))
        pending = ops.linear_planes(h, self._wplanes(blk.mlp.lin2), blk.mlp.lin2.bias)
        return shortcut, pending

    @torch.no_grad()
    def forward(self, x: torch.Tensor, trace_blocks=()):
        """[B, in_chans, S, S] -> [B, out_chans, S/patch, S/patch] (image_encoder.py:106-117).
        trace_blocks (tests): block indices whose OUTPUT tokens [B, G, G, C] are returned as well (-1: the tokens entering block
        0) -> (y, {index: tokens}); each costs one extra add, the block's residual being folded into the next block's norm."""
        if not x.is_cuda:
            raise RuntimeError("ImageEncoderViT runs on the HIP kernels only (no CPU path)")
        B, _, S, _ = x.shape
        G = S // self.patch_size
        T = G * G
        pe = self.patch_embed.proj
        t = ops.linear_nograd(ops.patchify(x.float().contiguous(), self.patch_size), pe.weight.view(pe.weight.shape[0], -1),
                              pe.bias)
        if self.pos_embed is not None:
            t = t.view(B, T, -1).add_(self.pos_embed.view(1, T, -1)).view(B * T, -1)
        shortcut, pending = t, None  # `pending`: the previous block's MLP output, still to be added to `shortcut`
        trace = {}
        if -1 in trace_blocks:
            trace[-1] = t.view(B, G, G, -1).clone()
        for bi, blk in enumerate(self.blocks):
            a, ws = blk.attn, blk.window_size
            if self._planes_ok(blk):
                shortcut, pending = self._block_planes(blk, shortcut, pending, B, G)
                if bi in trace_blocks:
                    trace[bi] = ops.window_merge_add(pending, shortcut, B, G, G, 0).view(B, G, G, -1)
                continue
            if pending is None:
                y = ops.layernorm(shortcut, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
            else:  # x = x + mlp(norm2(x)) of the previous block folds into this norm
                y, shortcut = ops.layernorm(shortcut, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, residual=pending,
                                            want_sum=True)
            if ws > 0:
                y = ops.window_partition(y, B, G, G, ws)
                n, Bw = ws, y.shape[0] // (ws * ws)
            else:
                n, Bw = G, B
            qkv = ops.linear_nograd(y, a.qkv.weight, a.qkv.bias)
            o = ops.attention(qkv, Bw, n * n, a.num_heads, n, a.rel_pos_h if a.use_rel_pos else None,
                              a.rel_pos_w if a.use_rel_pos else None)
            o = ops.linear_nograd(o, a.proj.weight, a.proj.bias)
            shortcut = ops.window_merge_add(o, shortcut, B, G, G, ws)  # window_unpartition + `shortcut + x`
            y = ops.layernorm(shortcut, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            y = ops.linear_nograd(y, blk.mlp.lin1.weight, blk.mlp.lin1.bias, ops.ACT_GELU)
            pending = ops.linear_nograd(y, blk.mlp.lin2.weight, blk.mlp.lin2.bias)
            if bi in trace_blocks:
                trace[bi] = ops.window_merge_add(pending, shortcut, B, G, G, 0).view(B, G, G, -1)
        t = ops.window_merge_add(pending, shortcut, B, G, G, 0) if pending is not None else shortcut
        c0, n0, c1, n1 = self.neck[0], self.neck[1], self.neck[2], self.neck[3]
        y = ops.linear_nograd(t, c0.weight.view(c0.weight.shape[0], -1))
        y = ops.layernorm(y, n0.weight, n0.bias, n0.eps)
        y = ops.linear_nograd(ops.patch_unfold(y, G, 3), c1.weight.view(c1.weight.shape[0], -1))
        y = ops.layernorm(y, n1.weight, n1.bias, n1.eps)
        y = y.view(B, G, G, -1).permute(0, 3, 1, 2).contiguous()
        return (y, trace) if trace_blocks else y


def build_sam_vit_h_encoder(device="cuda") -> ImageEncoderViT:
    """The encoder of build_sam_vit_h (build_sam.py:14-22,53-80): 1024 px, patch 16, dim 1280, depth 32, 16 heads, window 14,
    global attention at blocks 7/15/23/31, rel-pos, 256 output channels."""
    return ImageEncoderViT(depth=32, embed_dim=1280, img_size=1024, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                           num_heads=16, patch_size=16, qkv_bias=True, use_rel_pos=True, global_attn_indexes=(7, 15, 23, 31),
                           window_size=14, out_chans=256).to(device)
