"""GPU tests of the SAM image encoder forward: the reference-generated fixture (small configuration with a
padded windowed block and a global block), per-kernel checks against torch, and ViT-H-shaped layers against the oracle."""
import numpy as np
import pytest
import torch

from oracle import vit_oracle as V

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fp32_gemm():
    from samnerf_amd import ops
    ops.set_gemm_mode("fp32")
    yield
    ops.set_gemm_mode("bf16x3")


def _build(cfg, sd):
    from functools import partial
    from samnerf_amd.image_encoder import ImageEncoderViT
    enc = ImageEncoderViT(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
                          num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, out_chans=cfg.out_chans, qkv_bias=True,
                          norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), use_rel_pos=cfg.use_rel_pos,
                          window_size=cfg.window_size, global_attn_indexes=cfg.global_attn_indexes)
    enc.load_state_dict(sd, strict=True)
    return enc.cuda().eval()


def test_encoder_vs_reference_fixture(golden):
    g = golden("vit_small")
    cfg = V.ViTConfig(img_size=224, patch_size=16, embed_dim=32, depth=2, num_heads=2, mlp_ratio=2.0, out_chans=16,
                      window_size=5, global_attn_indexes=(1,))
    sd = {k[2:]: torch.from_numpy(np.ascontiguousarray(g[k])) for k in g.keys() if k.startswith("w:")}
    enc = _build(cfg, sd)
    x = torch.from_numpy(g["x"])
    y = enc(x.cuda()).cpu()
    ref = torch.from_numpy(g["y"])
    assert y.shape == ref.shape
    assert float((y - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    from samnerf_amd import ops
    ops.set_gemm_mode("bf16x3")  # the product default
    y3 = enc(x.cuda()).cpu()
    assert float((y3 - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("Bw,n,heads,hd,rel", [(3, 14, 2, 80, True), (1, 9, 1, 16, False), (2, 5, 3, 40, True)])
def test_attention_kernel_vs_torch(Bw, n, heads, hd, rel, mode):
    """fp32: the exact-fp32 matrix-core kernel; bf16x3: the 3-term-split kernel (product default), N(0,1) inputs make scores of
    magnitude ~10, so its 1e-6 relative product error shows as ~1e-5 in the softmax weights."""
    from samnerf_amd import ops
    ops.set_gemm_mode(mode)
    g = torch.Generator().manual_seed(n + hd)
    T, C = n * n, heads * hd
    qkv = torch.randn((Bw * T, 3 * C), generator=g)
    rph, rpw = torch.randn((2 * n - 1, hd), generator=g) * 0.3, torch.randn((2 * n - 1, hd), generator=g) * 0.3
    sd = {"qkv.weight": torch.eye(3 * C), "qkv.bias": torch.zeros(3 * C), "proj.weight": torch.eye(C), "proj.bias": torch.zeros(C),
          "rel_pos_h": rph, "rel_pos_w": rpw}
    # the oracle's attention() with identity qkv / proj layers: x = the qkv rows' "input" is qkv itself only if C_in = 3C;
    # compute the expected result directly instead
    q4 = qkv.view(Bw, T, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, Bw * heads, T, hd).double()
    q, k, v = q4.unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    if rel:
        Rh, Rw = V.get_rel_pos(n, n, rph.double()), V.get_rel_pos(n, n, rpw.double())
        rq = q.reshape(-1, n, n, hd)
        attn = (attn.view(-1, n, n, n, n) + torch.einsum("bhwc,hkc->bhwk", rq, Rh)[:, :, :, :, None]
                + torch.einsum("bhwc,wkc->bhwk", rq, Rw)[:, :, :, None, :]).view(-1, T, T)
    ref = (attn.softmax(-1) @ v).view(Bw, heads, T, hd).permute(0, 2, 1, 3).reshape(Bw * T, C)
    out = ops.attention(qkv.cuda(), Bw, T, heads, n, rph.cuda() if rel else None, rpw.cuda() if rel else None).cpu().double()
    assert float((out - ref).abs().max()) <= (2e-5 if mode == "fp32" else 1e-4)


def test_layernorm_window_kernels_vs_torch():
    from samnerf_amd import ops
    g = torch.Generator().manual_seed(0)
    x, r = torch.randn((300, 96), generator=g), torch.randn((300, 96), generator=g)
    w, b = torch.randn((96,), generator=g), torch.randn((96,), generator=g)
    y, s = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6, residual=r.cuda(), want_sum=True)
    ref = torch.nn.functional.layer_norm((x + r).double(), (96,), w.double(), b.double(), 1e-6)
    assert float((y.cpu().double() - ref).abs().max()) <= 1e-5 and torch.equal(s.cpu(), x + r)
    # rows of 256 NV columns take the register-resident kernel (ViT-H: 1280): with / without residual, a row count that is not a
    # multiple of the four rows of a workgroup
    for C2, with_res in ((1280, True), (1280, False), (256, True), (768, False)):
        x2, r2 = torch.randn((301, C2), generator=g), torch.randn((301, C2), generator=g)
        w2, b2 = torch.randn((C2,), generator=g), torch.randn((C2,), generator=g)
        if with_res:
            y2, s2 = ops.layernorm(x2.cuda(), w2.cuda(), b2.cuda(), 1e-6, residual=r2.cuda(), want_sum=True)
            assert torch.equal(s2.cpu(), x2 + r2)
            ref2 = torch.nn.functional.layer_norm((x2 + r2).double(), (C2,), w2.double(), b2.double(), 1e-6)
        else:
            y2 = ops.layernorm(x2.cuda(), w2.cuda(), b2.cuda(), 1e-6)
            y2 = y2[0] if isinstance(y2, tuple) else y2
            ref2 = torch.nn.functional.layer_norm(x2.double(), (C2,), w2.double(), b2.double(), 1e-6)
        assert float((y2.cpu().double() - ref2).abs().max()) <= 2e-5
    B, H, W, C, ws = 2, 14, 14, 8, 5
    t = torch.randn((B, H, W, C), generator=g)
    win, pad_hw = V.window_partition(t, ws)
    got = ops.window_partition(t.view(-1, C).cuda(), B, H, W, ws).cpu()
    assert torch.equal(got, win.reshape(-1, C))
    sc = torch.randn((B * H * W, C), generator=g)
    merged = ops.window_merge_add(got.cuda(), sc.cuda(), B, H, W, ws).cpu()
    assert torch.equal(merged, sc + V.window_unpartition(win, ws, pad_hw, (H, W)).reshape(-1, C))
    img = torch.randn((2, 3, 64, 64), generator=g)
    rows = ops.patchify(img.cuda(), 16).cpu()
    ref_rows = torch.nn.functional.unfold(img, 16, stride=16).transpose(1, 2).reshape(-1, 3 * 256)
    assert torch.equal(rows, ref_rows)


@pytest.mark.parametrize("mode,planes", [("fp32", False), ("bf16x3", False), ("bf16x3", True)])
def test_vit_h_shaped_block_vs_oracle(mode, planes, monkeypatch):
    """One windowed and one global block at ViT-H width (1280, 16 heads of 80, window 14, 64 x 64 tokens): oracle on the GPU
    (torch fp32) against the HIP forward -- exact-fp32 GEMMs, the bf16-split tiled kernel, and the product default: the blocks'
    GEMMs on operands split by their producers (csrc/gemm_planes.hip)."""
    from samnerf_amd import image_encoder, ops
    ops.set_gemm_mode(mode)
    monkeypatch.setattr(image_encoder, "PLANES_PATH", planes)
    cfg = V.ViTConfig(depth=2, global_attn_indexes=(1,))
    sd = V.init_weights(cfg, seed=1)
    enc = _build(cfg, sd)
    assert enc._planes_ok(enc.blocks[0]) == planes
    x = torch.randn((1, 3, 1024, 1024), generator=torch.Generator().manual_seed(2))
    y = enc(x.cuda())
    with torch.no_grad():
        ref = V.forward({k: v.cuda() for k, v in sd.items()}, x.cuda(), cfg)
    assert y.shape == (1, 256, 64, 64)
    err = float((y - ref).abs().max())
    assert err <= 2e-4 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_full_depth_vit_h_on_one_image(mode):
    """(fp32: exact GEMMs; bf16x3: the product default, the blocks on the split-operand GEMMs.)  BASELINE config #5, encoder half, AT SIZE: the full ViT-H (32 blocks, 1280 wide, 16 heads, window 14, global attention at
    blocks 7 / 15 / 23 / 31; build_sam.py:14-21,53-80) on one 1024 x 1024 image -- output finite and of the right shape, and
    block-wise parity on the FIRST block (windowed, 64 x 64 tokens padded to 70 x 70) and the LAST one (global, 4096 x 4096
    attention with decomposed relative positions): the oracle's block on the very tokens the HIP forward fed its block."""
    from samnerf_amd import ops
    from samnerf_amd.image_encoder import build_sam_vit_h_encoder
    ops.set_gemm_mode(mode)
    cfg = V.ViTConfig()  # ViT-H defaults
    assert (cfg.depth, cfg.embed_dim, cfg.num_heads, cfg.window_size) == (32, 1280, 16, 14)
    sd = V.init_weights(cfg, seed=3)
    enc = build_sam_vit_h_encoder().eval()
    enc.load_state_dict(sd, strict=True)
    assert enc._planes_ok(enc.blocks[0]) == (mode == "bf16x3")
    x = torch.randn((1, 3, 1024, 1024), generator=torch.Generator().manual_seed(4))
    y, trace = enc(x.cuda(), trace_blocks=(-1, 0, 30, 31))
    assert y.shape == (1, 256, 64, 64) and bool(torch.isfinite(y).all())
    for i in (0, 31):
        t_in = trace[i - 1].cpu()
        with torch.no_grad():
            ref = V.block(sd, i, t_in, cfg)
        got = trace[i].cpu()
        scale = max(1.0, float(ref.abs().max()))
        assert float((got - ref).abs().max()) <= 2e-4 * scale, (i, float((got - ref).abs().max()), scale)
    # the neck on the last block's tokens
    with torch.no_grad():
        t = trace[31].cpu()
        n = torch.nn.functional.conv2d(t.permute(0, 3, 1, 2), sd["neck.0.weight"])
        n = V.layer_norm_2d(n, sd["neck.1.weight"], sd["neck.1.bias"], cfg.ln_eps)
        n = torch.nn.functional.conv2d(n, sd["neck.2.weight"], padding=1)
        n = V.layer_norm_2d(n, sd["neck.3.weight"], sd["neck.3.bias"], cfg.ln_eps)
    assert float((y.cpu() - n).abs().max()) <= 2e-4 * max(1.0, float(n.abs().max()))


def test_sam_preprocess_kernel(golden):
    """snf_sam_preprocess against the reference's own Sam.preprocess (modeling/sam.py:164-174; fixture from
    tests/golden/make_golden.py): uint8 and float, landscape / portrait / square; then the embedder's crop on a small encoder."""
    from samnerf_amd import ops
    g = golden("sam_preprocess")
    mean, std = torch.from_numpy(g["mean"]).cuda(), torch.from_numpy(g["std"]).cuda()
    for k in "abc":
        got = ops.sam_preprocess(torch.from_numpy(g[k]).cuda(), mean, std, 64).cpu()
        want = torch.from_numpy(g[k + "_out"])
        assert got.shape == want.shape
        assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()), k
        h, w = g[k].shape[-2:]
        assert float(got[..., h:, :].abs().max() if h < 64 else 0.0) == 0.0 and float(got[..., :, w:].abs().max() if w < 64 else 0.0) == 0.0
    from samnerf_amd.sam_utils import SamImageEmbedder
    cfg = V.ViTConfig(img_size=224, patch_size=16, embed_dim=32, depth=2, num_heads=2, mlp_ratio=2.0, out_chans=16, window_size=5,
                      global_attn_indexes=(1,))
    sd = V.init_weights(cfg, seed=4)
    enc = _build(cfg, sd)
    emb = SamImageEmbedder(enc)
    img = torch.randint(0, 256, (1, 3, 150, 224), dtype=torch.uint8, generator=torch.Generator().manual_seed(6))
    feats = emb.set_torch_image(img.cuda(), (300, 448))
    with torch.no_grad():
        ref = V.forward(sd, V.sam_preprocess(img, emb.pixel_mean.cpu(), emb.pixel_std.cpu(), 224), cfg)
    assert float((feats.cpu() - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))
    assert emb.embedding().shape == (16, 10, 14)  # ceil(300 / 448 * 14) rows of the 14 x 14 map
    import os, tempfile
    with tempfile.TemporaryDirectory() as d:  # the offline writer's file (get_image_embeddings.py:57-60) and the reader's way back
        emb.save(os.path.join(d, "frame_00001.npy"))
        back = np.load(os.path.join(d, "frame_00001.npy"))
    assert back.dtype == np.float32 and back.shape == (16, 10, 14)
    assert np.array_equal(back, emb.embedding().cpu().numpy())


@pytest.mark.parametrize("M,K,Nc", [(4900, 1280, 3840), (4096, 5120, 1280), (4096, 1280, 5120), (100, 64, 128), (333, 192, 256)])
def test_split_operand_gemm_vs_fp64(M, K, Nc):
    """snf_linear_planes_fwd (both tile shapes: 4900 x 3840 takes the 256 x 128 tile, the others 128 x 128) against fp64 on the
    operands it was given: fp32 output with bias, and the GELU output written as the next GEMM's operand planes."""
    from samnerf_amd import ops
    ops.set_gemm_mode("bf16x3")
    g = torch.Generator(device="cuda").manual_seed(M + K + Nc)
    a = torch.randn((M, K), device="cuda", generator=g)
    w = torch.randn((Nc, K), device="cuda", generator=g) * K ** -0.5
    b = torch.randn((Nc,), device="cuda", generator=g) * 0.1
    ap, wp = ops.split_planes_kb(a), ops.split_weight_planes(w)
    # the planes carry x to 2^-17 relative (hi + lo of a 3-term split); rows of the k-blocked layout land where they should
    assert float((ap.float() - a).abs().max()) <= 2 ** -16 * float(a.abs().max())
    assert float(((wp[0].float() + wp[1].float()) - w).abs().max()) <= 2 ** -16 * float(w.abs().max())
    ref = a.double() @ w.double().T + b.double()
    y = ops.linear_planes(ap, wp, b)
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 2e-5 * scale
    out = ops.Planes.empty(M, Nc, "cuda")
    ops.linear_planes(ap, wp, b, ops.ACT_GELU, out=out)
    gref = torch.nn.functional.gelu(ref)
    assert float((out.float().double() - gref).abs().max()) <= 3e-5 * scale
    # against the tiled kernel that splits fp32 operands itself: the same products, another summation order
    y_t = ops.linear_nograd(a, w, b)
    assert float((y - y_t).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("M,K,Nc", [(700, 192, 640), (4900, 1280, 1280)])
def test_split_operand_gemm_tile_shapes_agree(M, K, Nc):
    """Every tile shape of snf_linear_planes_fwd (snf_linear_planes_fwd_shape: 4-wave workgroups of 128 / 256 rows x 64 / 128 / 160
    columns, 8-wave workgroups of 256 rows) runs the same products in the same order per output: bit-identical results, fp32 and
    plane outputs, ragged last row tile included."""
    from samnerf_amd import ops
    ops.set_gemm_mode("bf16x3")
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn((M, K), device="cuda", generator=g)
    w = torch.randn((Nc, K), device="cuda", generator=g) * K ** -0.5
    b = torch.randn((Nc,), device="cuda", generator=g) * 0.1
    ap, wp = ops.split_planes_kb(a), ops.split_weight_planes(w)
    y0 = ops.linear_planes(ap, wp, b)
    p0 = ops.linear_planes(ap, wp, b, ops.ACT_GELU, out=ops.Planes.empty(M, Nc, "cuda"))
    for sh in [(1, 2), (2, 2), (1, 4), (2, 4), (1, 5), (2, 5), (-1, 4), (-1, 5)]:
        y = ops.linear_planes(ap, wp, b, shape=sh)
        assert torch.equal(y, y0), sh
        p = ops.linear_planes(ap, wp, b, ops.ACT_GELU, out=ops.Planes.empty(M, Nc, "cuda"), shape=sh)
        assert torch.equal(p.hi, p0.hi) and torch.equal(p.lo, p0.lo), sh


@pytest.mark.parametrize("M,K,Nc", [(700, 192, 640), (4900, 1280, 3840), (4096, 1280, 5120), (300, 128, 512), (257, 64, 256)])
def test_split_operand_gemm_with_both_operands_through_lds(M, K, Nc):
    """snf_linear_planes_kb_fwd (k-blocked weights, 256 x 320 / 256 x 256 tiles, LDS-DMA staging) == snf_linear_planes_fwd bit for
    bit: fp32 output with bias, GELU plane output, ragged last row tile; repeated launches agree (the staging has no race)."""
    from samnerf_amd import ops
    ops.set_gemm_mode("bf16x3")
    g = torch.Generator(device="cuda").manual_seed(7 + M)
    a = torch.randn((M, K), device="cuda", generator=g)
    w = torch.randn((Nc, K), device="cuda", generator=g) * K ** -0.5
    b = torch.randn((Nc,), device="cuda", generator=g) * 0.1
    ap, wp, wkb = ops.split_planes_kb(a), ops.split_weight_planes(w), ops.split_weight_planes_kb(w)
    y0 = ops.linear_planes(ap, wp, b)
    p0 = ops.linear_planes(ap, wp, b, ops.ACT_GELU, out=ops.Planes.empty(M, Nc, "cuda"))
    for _ in range(5):
        y = ops.linear_planes(ap, wkb, b)
        assert torch.equal(y, y0)
        p = ops.linear_planes(ap, wkb, b, ops.ACT_GELU, out=ops.Planes.empty(M, Nc, "cuda"))
        assert torch.equal(p.hi, p0.hi) and torch.equal(p.lo, p0.lo)
    y = ops.linear_planes(ap, wkb, None)
    assert torch.equal(y, ops.linear_planes(ap, wp, None))


def test_producers_write_operand_planes(monkeypatch):
    """snf_layernorm_planes (with the window partition's row map and untouched zero padding) and snf_attention_planes against
    the fp32 kernels they replace: the planes hold the same values to the split's 2^-16."""
    from samnerf_amd import ops
    ops.set_gemm_mode("bf16x3")
    g = torch.Generator(device="cuda").manual_seed(11)
    B, G, C, ws = 2, 20, 512, 7
    x = torch.randn((B * G * G, C), device="cuda", generator=g)
    r = torch.randn((B * G * G, C), device="cuda", generator=g)
    w = 1 + 0.1 * torch.randn((C,), device="cuda", generator=g)
    b = 0.1 * torch.randn((C,), device="cuda", generator=g)
    y_ref, s_ref = ops.layernorm(x, w, b, 1e-6, residual=r, want_sum=True)
    # without windows
    out = ops.Planes.empty(B * G * G, C, "cuda")
    _, s = ops.layernorm_planes(x, w, b, 1e-6, out, residual=r, want_sum=True)
    assert torch.equal(s, s_ref)
    assert float((out.float() - y_ref).abs().max()) <= 2 ** -16 * float(y_ref.abs().max())
    # at the window partition's rows (20 -> 3 x 3 windows of 7: one padded row and column of windows)
    part = ops.window_partition(y_ref, B, G, G, ws)
    outw = ops.Planes.empty(part.shape[0], C, "cuda", zero=True)
    ops.layernorm_planes(x, w, b, 1e-6, outw, residual=r, grid=(G, G, ws))
    assert float((outw.float() - part).abs().max()) <= 2 ** -16 * float(y_ref.abs().max())
    assert bool((outw.float()[part.abs().sum(1) == 0] == 0).all())
    # norm2 with the window un-partition + residual add folded in == snf_window_merge_add, then the plain kernel (bit for bit)
    wins = torch.randn((part.shape[0], C), device="cuda", generator=g)
    merged = ops.window_merge_add(wins, x, B, G, G, ws)
    ref_pl = ops.layernorm_planes(merged, w, b, 1e-6, ops.Planes.empty(B * G * G, C, "cuda"))
    got_pl, got_sum = ops.layernorm_planes_merge(x, wins, w, b, 1e-6, ops.Planes.empty(B * G * G, C, "cuda"), B, G, G, ws)
    assert torch.equal(got_sum, merged) and torch.equal(got_pl.hi, ref_pl.hi) and torch.equal(got_pl.lo, ref_pl.lo)
    # attention
    Bw, n, heads, hd = 3, 14, 4, 80
    qkv = torch.randn((Bw * n * n, 3 * heads * hd), device="cuda", generator=g)
    rh = 0.1 * torch.randn((2 * n - 1, hd), device="cuda", generator=g)
    rw = 0.1 * torch.randn((2 * n - 1, hd), device="cuda", generator=g)
    o_ref = ops.attention(qkv, Bw, n * n, heads, n, rh, rw)
    from samnerf_amd import ops_vit
    monkeypatch.setattr(ops_vit, "FUSED_WINDOW_RELPOS", False)  # snf_relpos + snf_attention_planes: the same scores
    op = ops.attention_planes(qkv, Bw, n * n, heads, n, ops.Planes.empty(Bw * n * n, heads * hd, "cuda"), rh, rw)
    assert float((op.float() - o_ref).abs().max()) <= 2 ** -16 * float(o_ref.abs().max())
    # the position terms formed inside the kernel on the bf16-split matrix cores (snf_attention_planes_rp): scores of magnitude ~10
    # move by ~1e-6 relative, the softmax weights by ~1e-5
    monkeypatch.setattr(ops_vit, "FUSED_WINDOW_RELPOS", True)
    of = ops.attention_planes(qkv, Bw, n * n, heads, n, ops.Planes.empty(Bw * n * n, heads * hd, "cuda"), rh, rw)
    assert float((of.float() - o_ref).abs().max()) <= 1e-4
    q4 = qkv.view(Bw, n * n, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, Bw * heads, n * n, hd).double()
    qq, kk, vv = q4.unbind(0)
    att = (qq * hd ** -0.5) @ kk.transpose(-2, -1)
    Rh, Rw = V.get_rel_pos(n, n, rh.double().cpu()).cuda(), V.get_rel_pos(n, n, rw.double().cpu()).cuda()
    rq = qq.reshape(-1, n, n, hd)
    att = (att.view(-1, n, n, n, n) + torch.einsum("bhwc,hkc->bhwk", rq, Rh)[:, :, :, :, None]
           + torch.einsum("bhwc,wkc->bhwk", rq, Rw)[:, :, :, None, :]).view(-1, n * n, n * n)
    ref64 = (att.softmax(-1) @ vv).view(Bw, heads, n * n, hd).permute(0, 2, 1, 3).reshape(Bw * n * n, heads * hd)
    assert float((of.float().double() - ref64).abs().max()) <= 1e-4


@pytest.mark.parametrize("n,heads,hd", [(32, 2, 80), (64, 2, 80), (32, 1, 64), (64, 1, 32)])
def test_position_terms_of_large_grids_on_the_matrix_cores(n, heads, hd):
    """snf_relpos on grids with n >= 32 (the encoder's 64 x 64 global blocks) in the bf16-split gemm mode runs k_relpos_b3 (query x
    table products on the matrix cores): against fp64 and against the fp32 vector-ALU kernel of gemm mode 0; every (query, j) written."""
    from samnerf_amd import ops
    from samnerf_amd._opcore import _launch, _p, _stream
    g = torch.Generator(device="cuda").manual_seed(n + hd)
    T, C, Bw = n * n, heads * hd, 1
    qkv = torch.randn((Bw * T, 3 * C), device="cuda", generator=g)
    rh = 0.3 * torch.randn((2 * n - 1, hd), device="cuda", generator=g)
    rw = 0.3 * torch.randn((2 * n - 1, hd), device="cuda", generator=g)

    def run(mode):
        ops.set_gemm_mode(mode)
        rel = torch.full((Bw * heads * T, 2 * n), float("nan"), device="cuda")
        _launch("snf_relpos", _p(qkv), Bw, T, heads, hd, n, _p(rh), _p(rw), _p(rel), _stream())
        torch.cuda.synchronize()
        return rel

    try:
        r_b3, r_f32 = run("bf16x3"), run("fp32")
    finally:
        ops.set_gemm_mode("bf16x3")
    q = qkv.view(Bw, T, 3, heads, hd)[:, :, 0].permute(0, 2, 1, 3).reshape(Bw * heads, n, n, hd).double()
    Rh, Rw = V.get_rel_pos(n, n, rh.double().cpu()).cuda(), V.get_rel_pos(n, n, rw.double().cpu()).cuda()
    ref = torch.cat([torch.einsum("bhwc,hkc->bhwk", q, Rh), torch.einsum("bhwc,wkc->bhwk", q, Rw)], dim=-1).reshape(-1, 2 * n)
    scale = float(ref.abs().max())
    assert torch.isfinite(r_b3).all() and torch.isfinite(r_f32).all()
    assert float((r_f32.double() - ref).abs().max()) <= 2e-6 * scale
    # (3-term split: every product carries ~2^-16 of |q| |r|, hd = 80 of them per sum -- measured 6.4e-6 of the largest term)
    assert float((r_b3.double() - ref).abs().max()) <= 1.5e-5 * scale


@pytest.mark.parametrize("Bw,n,heads,hd", [(2, 9, 1, 16), (2, 5, 3, 40), (1, 16, 2, 64), (3, 14, 16, 80), (2, 7, 2, 96)])
def test_windowed_attention_with_in_kernel_position_terms(Bw, n, heads, hd):
    """snf_attention_planes_rp (relative-position terms formed inside the attention kernel from the 2n-1-row tables) against fp64:
    grids from 5 x 5 to 16 x 16 (2n - 1 = 31 table rows: the most one MFMA tile takes), head dims of one, two and three 32-blocks,
    a ragged last query tile and key tile."""
    from samnerf_amd import ops
    ops.set_gemm_mode("bf16x3")
    g = torch.Generator(device="cuda").manual_seed(n * 100 + hd)
    T, C = n * n, heads * hd
    qkv = torch.randn((Bw * T, 3 * C), device="cuda", generator=g)
    rh = 0.3 * torch.randn((2 * n - 1, hd), device="cuda", generator=g)
    rw = 0.3 * torch.randn((2 * n - 1, hd), device="cuda", generator=g)
    out = ops.attention_planes(qkv, Bw, T, heads, n, ops.Planes.empty(Bw * T, C, "cuda"), rh, rw).float().double()
    q4 = qkv.view(Bw, T, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, Bw * heads, T, hd).double()
    q, k, v = q4.unbind(0)
    att = (q * hd ** -0.5) @ k.transpose(-2, -1)
    Rh, Rw = V.get_rel_pos(n, n, rh.double().cpu()).cuda(), V.get_rel_pos(n, n, rw.double().cpu()).cuda()
    rq = q.reshape(-1, n, n, hd)
    att = (att.view(-1, n, n, n, n) + torch.einsum("bhwc,hkc->bhwk", rq, Rh)[:, :, :, :, None]
           + torch.einsum("bhwc,wkc->bhwk", rq, Rw)[:, :, :, None, :]).view(-1, T, T)
    ref = (att.softmax(-1) @ v).view(Bw, heads, T, hd).permute(0, 2, 1, 3).reshape(Bw * T, C)
    assert float((out - ref).abs().max()) <= 1e-4
