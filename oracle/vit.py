"""ViT / MAE encoder-decoder oracle (TEST INFRASTRUCTURE).  Plain PyTorch CPU ops, functional form:
`params` is a dict keyed exactly like the reference's `MaskedAutoencoderViT.state_dict()`.

Follows models/mae.py (PatchEmbed :25-43, AttentionKBiasZero :102-141, BlockKBiasZero :144-163,
MaskedAutoencoderViT :166-469) and models/pos_embed.py (:16-64, :97-109).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import rounding as R   # identity hooks unless rounding.mirror_hip_bf16() is active (see oracle/rounding.py)

LN_EPS = 1e-6  # models/mae.py:495 partial(nn.LayerNorm, eps=1e-6)

VIT_SIZES = {  # models/mae.py:492-517 (+ "large": extension for BASELINE config 5)
    "tiny": dict(embed_dim=192, depth=12, num_heads=3),
    "small": dict(embed_dim=384, depth=12, num_heads=6),
    "base": dict(embed_dim=768, depth=12, num_heads=12),
    "large": dict(embed_dim=1024, depth=24, num_heads=16),
}


# ------------------------------------------------------------------ positional tables (models/pos_embed.py)
def sincos_1d(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim, grid_sizes, cls_token=True):
    """get_2d_sincos_pos_embed (models/pos_embed.py:16-34): meshgrid 'w first', first half encodes w."""
    gH, gW = grid_sizes
    gw, gh = np.meshgrid(np.arange(gW, dtype=np.float32), np.arange(gH, dtype=np.float32))
    emb = np.concatenate([sincos_1d(embed_dim // 2, gw), sincos_1d(embed_dim // 2, gh)], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def sinusoid_table(n_position, d_hid, cls_token=True):
    """get_sinusoid_encoding_table (models/pos_embed.py:97-109)."""
    j = np.arange(d_hid)
    ang = np.arange(n_position, dtype=np.float64)[:, None] / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    tab = ang.copy()
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    if cls_token:
        tab = np.concatenate([np.zeros([1, d_hid]), tab], axis=0)
    return tab


def _cubic_w(t):
    A = -0.75
    c1 = lambda x: ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = lambda x: ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return np.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], 0)


def _taps_half_pixel(n_in, n_out, scale_factor):
    """bicubic, align_corners=False, explicit scale_factor: src = (dst+0.5)/scale - 0.5 (not clamped)."""
    inv = np.float32(1.0 / scale_factor)
    src = (inv * (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5)).astype(np.float64)
    fl = np.floor(src)
    idx = np.stack([np.clip(fl + k, 0, n_in - 1) for k in (-1, 0, 1, 2)], 0).astype(np.int64)
    return idx, _cubic_w(src - fl)


def interpolate_pos_embed(pos_embed, grid, freq_bins, frames, patch=(16, 16)):
    """interpolate_pos_encoding (models/mae.py:367-392).  pos_embed [1, 1+gh*gw, d] (numpy) ->
    [1, 1+nf*nt, d] for an input of `freq_bins` x `frames`."""
    pos_embed = np.asarray(pos_embed, dtype=np.float64)
    gh, gw = grid
    nf, nt = freq_bins // patch[0], frames // patch[1]
    if nf * nt == gh * gw:
        return pos_embed
    d = pos_embed.shape[-1]
    P = pos_embed[0, 1:].reshape(gh, gw, d)
    sf, st = (nf + 0.1) / gh, (nt + 0.1) / gw
    of, ot = int(math.floor(gh * sf)), int(math.floor(gw * st))
    assert of == nf and ot == nt
    iy, wy = _taps_half_pixel(gh, of, sf)
    ix, wx = _taps_half_pixel(gw, ot, st)
    rows = sum(P[iy[k]] * wy[k][:, None, None] for k in range(4))            # [of, gw, d]
    out = sum(rows[:, ix[k]] * wx[k][None, :, None] for k in range(4))       # [of, ot, d]
    return np.concatenate([pos_embed[:, :1], out.reshape(1, of * ot, d)], axis=1)


# ------------------------------------------------------------------ parameter construction
def init_params(size="tiny", img_size=(64, 96), patch=(16, 16), seed=0, use_decoder=False,
                decoder_embed_dim=384, decoder_depth=4, decoder_num_heads=6, dtype=torch.float32, **dims):
    """Random-init parameters with the reference's shapes / init families (models/mae.py:242-279):
    xavier-uniform Linear weights, zero biases, N(0, 0.02) cls/mask tokens, sin-cos pos tables.
    (Same distributions, not the same RNG stream; parity tests load explicit state dicts.)"""
    cfg = dict(VIT_SIZES[size]) if size in VIT_SIZES else {}
    cfg.update(dims)
    d, depth = cfg["embed_dim"], cfg["depth"]
    g = torch.Generator().manual_seed(seed)
    grid = (img_size[0] // patch[0], img_size[1] // patch[1])

    def xavier(o, i):
        a = math.sqrt(6.0 / (i + o))
        return (torch.rand(o, i, generator=g, dtype=torch.float64) * 2 - 1).mul_(a).to(dtype)

    p = {}
    p["cls_token"] = (0.02 * torch.randn(1, 1, d, generator=g, dtype=torch.float64)).to(dtype)
    p["pos_embed"] = torch.from_numpy(sincos_2d(d, grid)).to(dtype).unsqueeze(0)
    p["patch_embed.proj.weight"] = xavier(d, patch[0] * patch[1]).reshape(d, 1, patch[0], patch[1])
    bnd = 1.0 / math.sqrt(patch[0] * patch[1])  # nn.Conv2d default bias init (never re-initialised)
    p["patch_embed.proj.bias"] = ((torch.rand(d, generator=g, dtype=torch.float64) * 2 - 1) * bnd).to(dtype)

    def block(prefix, dim):
        p[prefix + "norm1.weight"], p[prefix + "norm1.bias"] = torch.ones(dim, dtype=dtype), torch.zeros(dim, dtype=dtype)
        p[prefix + "attn.q_bias"], p[prefix + "attn.v_bias"] = torch.zeros(dim, dtype=dtype), torch.zeros(dim, dtype=dtype)
        p[prefix + "attn.qkv.weight"] = xavier(3 * dim, dim)
        p[prefix + "attn.proj.weight"], p[prefix + "attn.proj.bias"] = xavier(dim, dim), torch.zeros(dim, dtype=dtype)
        p[prefix + "norm2.weight"], p[prefix + "norm2.bias"] = torch.ones(dim, dtype=dtype), torch.zeros(dim, dtype=dtype)
        p[prefix + "mlp.fc1.weight"], p[prefix + "mlp.fc1.bias"] = xavier(4 * dim, dim), torch.zeros(4 * dim, dtype=dtype)
        p[prefix + "mlp.fc2.weight"], p[prefix + "mlp.fc2.bias"] = xavier(dim, 4 * dim), torch.zeros(dim, dtype=dtype)

    for i in range(depth):
        block(f"blocks.{i}.", d)
    p["norm.weight"], p["norm.bias"] = torch.ones(d, dtype=dtype), torch.zeros(d, dtype=dtype)
    if use_decoder:
        dd = decoder_embed_dim
        p["decoder_embed.weight"], p["decoder_embed.bias"] = xavier(dd, d), torch.zeros(dd, dtype=dtype)
        p["mask_token"] = (0.02 * torch.randn(1, 1, dd, generator=g, dtype=torch.float64)).to(dtype)
        p["decoder_pos_embed"] = torch.from_numpy(sinusoid_table(grid[0] * grid[1], dd)).to(dtype).unsqueeze(0)
        for i in range(decoder_depth):
            block(f"decoder_blocks.{i}.", dd)
        p["decoder_norm.weight"], p["decoder_norm.bias"] = torch.ones(dd, dtype=dtype), torch.zeros(dd, dtype=dtype)
        p["decoder_pred.weight"] = xavier(patch[0] * patch[1], dd)
        p["decoder_pred.bias"] = torch.zeros(patch[0] * patch[1], dtype=dtype)
    return p


def infer_arch(params, prefix="blocks."):
    depth = 1 + max(int(k.split(".")[1]) for k in params if k.startswith(prefix))
    return depth


# ------------------------------------------------------------------ forward
def attention(x, p, pre, num_heads):
    """AttentionKBiasZero.forward (models/mae.py:122-141): k-bias fixed at zero."""
    B, N, C = x.shape
    qb, vb = p[pre + "q_bias"], p[pre + "v_bias"]
    bias = torch.cat((qb, torch.zeros_like(vb), vb))
    qkv = R.q(F.linear(x, R.qw(p[pre + "qkv.weight"]), bias)).reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    attn = R.qf(R.qb((q @ k.transpose(-2, -1)) * (C // num_heads) ** -0.5).softmax(dim=-1))
    x = R.q((attn @ v).transpose(1, 2).reshape(B, N, C))
    return R.qb(F.linear(x, R.qw(p[pre + "proj.weight"]), p[pre + "proj.bias"]))


def block(x, p, pre, num_heads):
    """BlockKBiasZero.forward (models/mae.py:157-163); Mlp = fc1 -> exact-erf GELU -> fc2."""
    C = x.shape[-1]
    x = x + attention(R.q(F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], LN_EPS)), p, pre + "attn.", num_heads)
    h = R.q(F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], LN_EPS))
    h = R.gelu(F.linear(h, R.qw(p[pre + "mlp.fc1.weight"]), p[pre + "mlp.fc1.bias"]))
    return x + R.qb(F.linear(h, R.qw(p[pre + "mlp.fc2.weight"]), p[pre + "mlp.fc2.bias"]))


CONVSTEM_STRIDES = {  # models/mae.py:58-67
    (16, 16): [(2, 2)] * 4,
    (16, 8): [(2, 2)] * 3 + [(2, 1)],
    (8, 8): [(2, 2)] * 3 + [(1, 1)],
    (64, 2): [(2, 2)] + [(2, 1)] * 5,
}


def conv_stem(x, p, patch, bn_stats=None):
    """ConvStem.forward (models/mae.py:89-99): [3x3 conv (no bias) -> BatchNorm2d (train mode: batch statistics) -> ReLU] per
    stride entry, then a 1x1 conv with bias; tokens = flatten(2).transpose(1, 2).  `bn_stats` (list) receives (mean, biased var, n)
    per BatchNorm for the running-statistics replay; bn_stats = "eval" selects eval mode (running statistics, nothing recorded).  Hooks: operands of every conv are bf16 on the HIP path (R.qf / R.qw), the
    pre-BatchNorm maps are fp32 with bf16 gradients (R.qb)."""
    strides = CONVSTEM_STRIDES[tuple(patch)]
    h = x
    for l, st in enumerate(strides):
        # (the single-input-channel first convolution is a direct fp32 kernel on the HIP path: fp32 image, fp32 weight; only its
        # output gradient is consumed as bf16)
        qf, qw = (R.qf, R.qw) if l > 0 else ((lambda t: t), (lambda t: t))
        h = R.qb(F.conv2d(qf(h), qw(p[f"patch_embed.proj.{3 * l}.weight"]), None, stride=st, padding=1))
        bn = f"patch_embed.proj.{3 * l + 1}"
        if bn_stats == "eval":                               # nn.BatchNorm2d in eval mode: the running statistics of the state dict
            h = F.relu(F.batch_norm(h, p[bn + ".running_mean"], p[bn + ".running_var"], p[bn + ".weight"], p[bn + ".bias"], False, 0.1, 1e-5))
            continue
        if bn_stats is not None:
            bn_stats.append((h.mean((0, 2, 3)).detach(), h.var((0, 2, 3), unbiased=False).detach(), h.numel() // h.shape[1]))
        h = F.relu(F.batch_norm(h, None, None, p[bn + ".weight"], p[bn + ".bias"], True, 0.1, 1e-5))
    last = 3 * len(strides)
    h = R.qb(F.conv2d(R.qf(h), R.qw(p[f"patch_embed.proj.{last}.weight"]), p[f"patch_embed.proj.{last}.bias"]))
    return h.flatten(2).transpose(1, 2)


def patch_embed(x, p, patch=None, bn_stats=None):
    """PatchEmbed.forward (models/mae.py:40-43), or ConvStem.forward when the parameters are a conv stem's."""
    if "patch_embed.proj.0.weight" in p:
        return conv_stem(x, p, patch, bn_stats)
    w = p["patch_embed.proj.weight"]
    return F.conv2d(R.qf(x), R.qw(w), p["patch_embed.proj.bias"], stride=w.shape[-2:]).flatten(2).transpose(1, 2)


def masking_from_noise(x, noise=None, mask=None, mask_ratio=0.0):
    """random_masking (models/mae.py:309-347) with the randomness made explicit: either `noise` [N, L]
    (what torch.rand would have produced) or a prefixed `mask` [N, L] (0 keep / 1 remove)."""
    N, L, D = x.shape
    if mask is not None:
        ids_shuffle = torch.argsort(mask.reshape(N, -1), dim=1)
        len_keep = int((mask[0] == 0).sum())
    elif noise is None or mask_ratio == 0:
        return x, torch.zeros(N, L, dtype=x.dtype), torch.arange(L).expand(N, L)
    else:
        len_keep = int(L * (1 - mask_ratio))
        ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :len_keep]
    x_masked = torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))
    m = torch.ones(N, L, dtype=x.dtype)
    m[:, :len_keep] = 0
    return x_masked, torch.gather(m, 1, ids_restore), ids_restore


def learned_pos_table(pos_embed, grid, freq_bins, frames, patch=(16, 16)):
    """interpolate_pos_encoding with `use_learned_pos_embd` (models/mae.py:367-392): the trained table itself only when the patch count
    matches AND the input is square (`w == h`, :372), otherwise its bicubic resampling (scale factors (n + 0.1) / grid) -- in torch, so
    that autograd carries the table's gradient through it."""
    gh, gw = grid
    nf, nt = freq_bins // patch[0], frames // patch[1]
    if nf * nt == gh * gw and freq_bins == frames:
        return pos_embed
    pp = pos_embed[:, 1:].reshape(1, gh, gw, -1).permute(0, 3, 1, 2)
    pp = F.interpolate(pp, scale_factor=((nf + 0.1) / gh, (nt + 0.1) / gw), mode="bicubic")
    assert tuple(pp.shape[-2:]) == (nf, nt)
    return torch.cat((pos_embed[:, :1], pp.permute(0, 2, 3, 1).reshape(1, nf * nt, -1)), dim=1)


def prepare_tokens(x, p, grid, noise=None, mask=None, mask_ratio=0.0, patch=None, bn_stats=None, learned_pos=False):
    """prepare_tokens (models/mae.py:349-365).  `patch` = patch size, required for a conv stem (a PatchEmbed carries it in its kernel)."""
    B, _, Fb, T = x.shape
    if patch is None:
        patch = tuple(p["patch_embed.proj.weight"].shape[-2:])
    tok = patch_embed(x, p, patch, bn_stats)
    if learned_pos:
        pos = learned_pos_table(p["pos_embed"], grid, Fb, T, tuple(patch))
    else:
        pos = torch.from_numpy(interpolate_pos_embed(p["pos_embed"].detach().numpy(), grid, Fb, T, tuple(patch))).to(x.dtype)
    tok = tok + pos[:, 1:]
    tok, m, ids_restore = masking_from_noise(tok, noise, mask, mask_ratio)
    cls = (p["cls_token"] + p["pos_embed"][:, :1]).expand(B, -1, -1)
    return torch.cat((cls, tok), dim=1), m, ids_restore


def forward_encoder(x, p, num_heads, grid, noise=None, mask=None, mask_ratio=0.0, patch=None, bn_stats=None, learned_pos=False):
    """forward_encoder (models/mae.py:394-400)."""
    tok, m, ids_restore = prepare_tokens(x, p, grid, noise, mask, mask_ratio, patch, bn_stats, learned_pos)
    for i in range(infer_arch(p)):
        tok = block(tok, p, f"blocks.{i}.", num_heads)
    C = tok.shape[-1]
    return F.layer_norm(tok, (C,), p["norm.weight"], p["norm.bias"], LN_EPS), m, ids_restore


def patchify(imgs, grid, patch=(16, 16)):
    """patchify (models/mae.py:282-293) for in_chans == 1."""
    h, w = grid
    ph, pw = patch
    x = imgs.reshape(imgs.shape[0], 1, h, ph, w, pw)
    return torch.einsum("nchpwq->nhwpqc", x).reshape(imgs.shape[0], h * w, ph * pw)


def forward_decoder(x, ids_restore, p, dec_heads):
    """forward_decoder (models/mae.py:411-435)."""
    x = R.qb(F.linear(R.qf(x), R.qw(p["decoder_embed.weight"]), p["decoder_embed.bias"]))
    n_mask = ids_restore.shape[1] + 1 - x.shape[1]
    x_ = torch.cat([x[:, 1:], p["mask_token"].expand(x.shape[0], n_mask, -1)], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).expand(-1, -1, x.shape[2]))
    x = torch.cat([x[:, :1], x_], dim=1) + p["decoder_pos_embed"]
    for i in range(infer_arch(p, "decoder_blocks.")):
        x = block(x, p, f"decoder_blocks.{i}.", dec_heads)
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), p["decoder_norm.weight"], p["decoder_norm.bias"], LN_EPS)
    return R.qb(F.linear(R.qf(x), R.qw(p["decoder_pred.weight"]), p["decoder_pred.bias"]))[:, 1:]


def recon_loss(imgs, pred, mask, grid, norm_pix=False):
    """forward_loss (models/mae.py:437-453); norm_pix: targets normalised per patch with the unbiased variance (:443-446)."""
    target = patchify(imgs, grid)
    if norm_pix:
        target = (target - target.mean(dim=-1, keepdim=True)) / (target.var(dim=-1, keepdim=True) + 1.e-6) ** .5
    loss = ((pred - target) ** 2).mean(dim=-1)
    return (loss * mask).sum() / mask.sum()


def forward(x, p, num_heads, grid, mean_pool=False, noise=None, mask=None, mask_ratio=0.0,
            masked_recon=False, dec_heads=6, patch=None, bn_stats=None, learned_pos=False):
    """MaskedAutoencoderViT.forward (models/mae.py:455-469)."""
    enc, m, ids_restore = forward_encoder(x, p, num_heads, grid, noise, mask, mask_ratio, patch, bn_stats, learned_pos)
    latent = enc[:, 1:].mean(dim=1) if mean_pool else enc[:, 0]
    if masked_recon:
        pred = forward_decoder(enc, ids_restore, p, dec_heads)
        return latent, recon_loss(x, pred, m, grid)
    return latent


def encode_vit(x, p, num_heads, grid, unit_frames, split_frames=True, use_cls=True):
    """utils.encode_vit (utils/utils.py:278-314), restated: right-pad to a multiple of unit_frames (a whole extra unit when the
    length already is one), embed unit by unit, then average the CLS embeddings over units, or lay the patch tokens out as
    [time, freq x d], drop the frames that came from padding and average over time."""
    pad = unit_frames - (x.shape[-1] % unit_frames)
    xp = torch.nn.functional.pad(x, (0, pad))
    if not split_frames:
        return forward(xp, p, num_heads, grid)
    n_units = xp.shape[-1] // unit_frames
    gf, gt = grid
    if use_cls:
        return torch.stack([forward(xp[..., i * unit_frames:(i + 1) * unit_frames], p, num_heads, grid) for i in range(n_units)], 1).mean(1)
    rows = []
    for i in range(n_units):
        enc, _, _ = forward_encoder(xp[..., i * unit_frames:(i + 1) * unit_frames], p, num_heads, grid)
        tok = enc[:, 1:]                                            # [B, gf * gt, d], frequency-major
        B, _, d = tok.shape
        rows.append(tok.reshape(B, gf, gt, d).permute(0, 2, 1, 3).reshape(B, gt, gf * d))
    seq = torch.cat(rows, 1)
    drop = int(gt * pad / unit_frames)
    if drop > 0:
        seq = seq[:, :-drop]
    return seq.mean(1)
