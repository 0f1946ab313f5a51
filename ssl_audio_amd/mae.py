"""MaskedAutoencoderViT on the MI355X engine -- same constructor, attribute names, state_dict keys and forward
signature as the reference's models/mae.py:166-469, so checkpoints and call sites carry over unchanged.

The nn.Module tree below only *owns parameters* (identical names / shapes / init families as the reference);
no submodule's forward is ever called.  Compute is the explicit HIP schedule in engine.py / functional.py.
The `vitc_*` variants put the ConvStem (convstem.py) in place of the patch projection.
Options: `norm_pix_loss` (models/mae.py:443-446) and the learned positional table (`--use_learned_pos_embd`, :198-199) are supported -- the
table with its gradient THROUGH the bicubic resampling to other widths (:370-392: `_learned_pos` returns the interpolation as a matrix A,
`TokensFn.backward` / `ConvStemTokensFn.backward` apply A^T to the summed token gradient; pinned by tests/golden/options.npz for the plain
patch projection and tests/golden/convstem_lpe.npz for the ConvStem).  Not provided (out of this path's scope): stochastic depth, forward_attn / forward_viz.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .convstem import ConvStem, ConvStemTokensFn, stem_flat_params
from .engine import encoder_apply
from .pos_embed import get_2d_sincos_pos_embed, get_sinusoid_encoding_table, interpolate_pos_encoding


def to_2tuple(x):
    return tuple(x) if isinstance(x, (list, tuple)) else (x, x)


class PatchEmbed(nn.Module):
    """Parameter holder for the 16x16/16 patch projection (models/mae.py:25-43): runs as patchify + MFMA GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)


class AttentionKBiasZero(nn.Module):
    """Parameter holder (models/mae.py:102-141): qkv without bias + separate q/v biases, k-bias fixed at zero."""

    def __init__(self, dim, num_heads=8, qkv_bias=True):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    """timm.models.vision_transformer.Mlp parameter layout: fc1 -> GELU(erf) -> fc2."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class BlockKBiasZero(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=True, norm_layer=nn.LayerNorm, **_):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = AttentionKBiasZero(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def engine_params(self):
        a, m = self.attn, self.mlp
        return (self.norm1.weight, self.norm1.bias, a.qkv.weight, a.q_bias, a.v_bias, a.proj.weight, a.proj.bias,
                self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)


class MaskedAutoencoderViT(nn.Module):
    def __init__(self, img_size=(64, 96), patch_size=(16, 16), in_chans=1,
                 embed_dim=768, depth=12, num_heads=12, conv_stem=False,
                 use_decoder=False, use_learned_pos_embd=False,
                 decoder_embed_dim=384, decoder_depth=4, decoder_num_heads=6,
                 mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False,
                 block_cls=BlockKBiasZero, use_2d_dec_pos_embd=False,
                 drop_path_rate=0.):
        super().__init__()
        if drop_path_rate:
            raise NotImplementedError("stochastic depth (drop_path_rate > 0) is not on the MI355X hot path (the reference never sets it)")
        if in_chans != 1:
            raise NotImplementedError("audio spectrogram input only (in_chans=1)")
        if (embed_dim // num_heads) != 64 or (use_decoder and decoder_embed_dim // decoder_num_heads != 64):
            raise NotImplementedError("the fused attention kernel is specialised for head_dim 64 (all reference ViT sizes)")
        self.img_size, self.in_chans, self.embed_dim = tuple(img_size), in_chans, embed_dim
        self.conv_stem, self.use_decoder, self.use_learned_pos_embd = conv_stem, use_decoder, use_learned_pos_embd
        self.num_heads, self.decoder_num_heads = num_heads, decoder_num_heads
        self.ln_eps = norm_layer(8).eps

        if conv_stem:                                        # trained, unlike the plain patch projection (models/mae.py:186-192)
            self.patch_embed = ConvStem(img_size, patch_size, in_chans, embed_dim)
        else:
            self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
            for param in self.patch_embed.parameters():      # random patch projection, frozen (models/mae.py:190-192)
                param.requires_grad = False
        total_patches = self.patch_embed.num_patches + 1
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        # fixed sin-cos table, or trained (`--use_learned_pos_embd`, models/mae.py:198-202)
        self.pos_embed = nn.Parameter(torch.zeros(1, total_patches, embed_dim), requires_grad=bool(use_learned_pos_embd))
        self.blocks = nn.ModuleList([block_cls(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        if use_decoder:
            self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
            self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
            self.decoder_pos_embed = nn.Parameter(torch.zeros(1, total_patches, decoder_embed_dim), requires_grad=False)
            self.decoder_blocks = nn.ModuleList([block_cls(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True,
                                                           norm_layer=norm_layer) for _ in range(decoder_depth)])
            self.decoder_norm = norm_layer(decoder_embed_dim)
            self.decoder_pred = nn.Linear(decoder_embed_dim, self.img_patch_dim(), bias=True)
        self.norm_pix_loss = norm_pix_loss
        self._pos_cache = {}
        self.initialize_weights(use_2d_dec_pos_embd)

    # ------------------------------------------------------------------ geometry helpers (models/mae.py:232-240)
    def patch_size(self):
        return self.patch_embed.patch_size

    def grid_size(self):
        return self.patch_embed.grid_size

    def img_patch_dim(self):
        ps = self.patch_size()
        return ps[0] * ps[1] * self.in_chans

    # ------------------------------------------------------------------ init (models/mae.py:242-279)
    def initialize_weights(self, use_2d_dec_pos_embd=False):
        if self.use_learned_pos_embd:
            torch.nn.init.normal_(self.pos_embed, std=.02)              # models/mae.py:244-245
        else:
            pos = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], self.grid_size())
            self.pos_embed.data.copy_(torch.from_numpy(pos).float().unsqueeze(0))
        if self.use_decoder:
            if use_2d_dec_pos_embd:
                dpos = get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], self.grid_size())
            else:
                dpos = get_sinusoid_encoding_table(self.grid_size()[0] * self.grid_size()[1], self.decoder_pos_embed.shape[-1])
            self.decoder_pos_embed.data.copy_(torch.from_numpy(dpos).float().unsqueeze(0))
        if not self.conv_stem:                               # the conv stem keeps nn.Conv2d's default initialisation (models/mae.py:260)
            w = self.patch_embed.proj.weight.data
            torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        torch.nn.init.normal_(self.cls_token, std=.02)
        if self.use_decoder:
            torch.nn.init.normal_(self.mask_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ positional table per input width (cached)
    def interpolate_pos_encoding(self, freq_bins, frames):
        if self.use_learned_pos_embd:
            return self._learned_pos(freq_bins, frames)[0]
        key = (freq_bins, frames, self.pos_embed.device, self.pos_embed._version)
        if key not in self._pos_cache:
            # keep one table per input WIDTH (a step with local crops alternates between two: clearing on every miss rebuilt both on the host
            # every step, and made such a step uncapturable); drop only what a rewritten / moved pos_embed made stale
            for k in [k for k in self._pos_cache if k[0] != "A" and k[2:] != key[2:]]:
                del self._pos_cache[k]
            pos = interpolate_pos_encoding(self.pos_embed.detach().cpu().numpy(), self.grid_size(), freq_bins, frames,
                                           self.patch_size())
            self._pos_cache[key] = torch.from_numpy(pos).float().to(self.pos_embed.device).contiguous()
        return self._pos_cache[key]

    def _learned_pos(self, freq_bins, frames):
        """(`pos` [1, 1 + L, d], A) for the TRAINED table (`--use_learned_pos_embd`): read live every call (the optimiser's launches
        rewrite it without touching `_version`), resampled by the reference's bicubic map whenever the input is not square
        (models/mae.py:370-392: `w == h` is the only shortcut there) -- A [L, gh * gw] on the device, None for the identity."""
        ph, pw = self.patch_size()
        nf, nt = freq_bins // ph, frames // pw
        gh, gw = self.grid_size()
        if nf * nt == gh * gw and freq_bins == frames:
            return self.pos_embed, None
        key = ("A", nf, nt, self.pos_embed.device)
        if key not in self._pos_cache:
            from .pos_embed import interpolation_matrix
            self._pos_cache[key] = torch.from_numpy(interpolation_matrix((gh, gw), nf, nt)).float().to(self.pos_embed.device).contiguous()
        A = self._pos_cache[key]
        table = self.pos_embed.detach()[0]
        pos = torch.empty(1 + nf * nt, table.shape[1], device=table.device)
        pos[0].copy_(table[0])
        ops.matmul_f32(A, table[1:], pos[1:])
        return pos.unsqueeze(0), A

    def patchify(self, imgs):
        ph, pw = self.patch_size()
        h, w = self.grid_size()
        x = imgs.reshape(shape=(imgs.shape[0], self.in_chans, h, ph, w, pw))
        x = torch.einsum('nchpwq->nhwpqc', x)
        return x.reshape(shape=(imgs.shape[0], h * w, self.img_patch_dim()))

    # ------------------------------------------------------------------ masking indices (models/mae.py:309-347)
    def masking_indices(self, N, L, mask_ratio, device, noise=None):
        """Returns (ids_keep int32 [N, keep], mask [N, L], ids_restore [N, L]).  Index bookkeeping only (argsort of the
        per-sample noise); the token gather itself is the HIP kernel behind TokensFn."""
        if isinstance(mask_ratio, (torch.Tensor, np.ndarray, list, tuple)):
            mask_in = torch.as_tensor(mask_ratio, device=device).detach()
            ids_shuffle = torch.argsort(mask_in.reshape(N, -1), dim=1)
            len_keep = int((mask_in[0] == 0).sum())
        else:
            len_keep = int(L * (1 - mask_ratio))
            if noise is None:
                noise = torch.rand(N, L, device=device)
            ids_shuffle = torch.argsort(noise, dim=1)
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        ids_keep = ids_shuffle[:, :len_keep]
        mask = torch.ones([N, L], device=device)
        mask[:, :len_keep] = 0
        mask = torch.gather(mask, dim=1, index=ids_restore)
        return ids_keep.to(torch.int32).contiguous(), mask, ids_restore

    # ------------------------------------------------------------------ forward (models/mae.py:349-469)
    def prepare_tokens(self, x, mask_ratio, noise=None):
        B, nc, w, h = x.shape
        L = (w // self.patch_size()[0]) * (h // self.patch_size()[1])
        pos, pos_A = self._learned_pos(w, h) if self.use_learned_pos_embd else (self.interpolate_pos_encoding(w, h), None)
        no_mask = not isinstance(mask_ratio, (torch.Tensor, np.ndarray, list, tuple)) and mask_ratio == 0
        if no_mask:
            ids_keep, mask = None, torch.zeros([B, L], device=x.device)
            ids_restore = torch.arange(L, device=x.device).to(torch.int)
        else:
            ids_keep, mask, ids_restore = self.masking_indices(B, L, mask_ratio, x.device, noise)
        if self.conv_stem:
            tok = ConvStemTokensFn.apply(x, self.cls_token, pos, ids_keep, self.patch_embed,
                                         self.pos_embed if self.use_learned_pos_embd else None, pos_A, *stem_flat_params(self.patch_embed))
        else:
            tok = Fn.TokensFn.apply(x, self.cls_token, self.patch_embed.proj.weight, self.patch_embed.proj.bias, pos, ids_keep,
                                    self.pos_embed if self.use_learned_pos_embd else None, pos_A)
        return tok, mask, ids_restore

    def _run_blocks(self, tok, blocks, norm, heads, pool):
        return encoder_apply(tok, [b.engine_params() for b in blocks], norm.weight, norm.bias, heads, self.ln_eps, pool)

    def forward_encoder(self, x, mask_ratio, noise=None):
        tok, mask, ids_restore = self.prepare_tokens(x, mask_ratio, noise)
        return self._run_blocks(tok, self.blocks, self.norm, self.num_heads, "all"), mask, ids_restore

    def forward_decoder(self, x, ids_restore):
        x = Fn.LinearFn.apply(x, self.decoder_embed.weight, self.decoder_embed.bias)
        x = Fn.MaeUnshuffleFn.apply(x, self.mask_token, self.decoder_pos_embed, ids_restore)   # mask tokens + un-shuffle + pos
        x = self._run_blocks(x, self.decoder_blocks, self.decoder_norm, self.decoder_num_heads, "all")
        return Fn.LinearFn.apply(x, self.decoder_pred.weight, self.decoder_pred.bias)      # [B, 1 + L, P]: CLS row kept, skipped in place by the loss

    def forward_loss(self, imgs, pred, mask):
        ph, pw = self.patch_size()
        # patchify folded in; row0 = 1 when pred still has its CLS row
        return Fn.MaeReconLossFn.apply(pred, imgs, mask, ph, pw, pred.shape[1] - mask.shape[1], bool(self.norm_pix_loss))

    def forward(self, imgs, mask_ratio=0, mean_pool=False, return_all=False, masked_recon=False, noise=None):
        if masked_recon or return_all:
            x, mask, ids_restore = self.forward_encoder(imgs, mask_ratio, noise)
            if return_all:
                latent = x
            elif mean_pool:
                latent = Fn.MeanTokensFn.apply(x)
            else:
                latent = x[:, 0].contiguous()
            if masked_recon:
                pred = self.forward_decoder(x, ids_restore)
                return latent, self.forward_loss(imgs, pred, mask)
            return latent
        tok, _, _ = self.prepare_tokens(imgs, mask_ratio, noise)
        return self._run_blocks(tok, self.blocks, self.norm, self.num_heads, "mean" if mean_pool else "cls")


def _vit(patch_size, embed_dim, depth, num_heads, **kwargs):
    return MaskedAutoencoderViT(patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_base_patchX(patch_size, **kwargs):
    return _vit(patch_size, 768, 12, 12, decoder_embed_dim=384, decoder_depth=4, decoder_num_heads=6, **kwargs)


def mae_vit_small_patchX(patch_size, **kwargs):
    return _vit(patch_size, 384, 12, 6, **kwargs)


def mae_vit_tiny_patchX(patch_size, **kwargs):
    return _vit(patch_size, 192, 12, 3, **kwargs)


def mae_vit_large_patchX(patch_size, **kwargs):
    """Not in the reference: BASELINE config 5's ViT-L (d=1024, 24 layers, 16 heads)."""
    return _vit(patch_size, 1024, 24, 16, decoder_embed_dim=384, decoder_depth=4, decoder_num_heads=6, **kwargs)


def mae_vitc_base_patchX(patch_size, **kwargs):
    """ViTC-B (models/mae.py:531-537): 11 blocks behind the conv stem."""
    return _vit(patch_size, 768, 11, 12, conv_stem=True, decoder_embed_dim=384, decoder_depth=4, decoder_num_heads=6, **kwargs)


def mae_vitc_small_patchX(patch_size, **kwargs):
    return _vit(patch_size, 384, 11, 6, conv_stem=True, **kwargs)


def mae_vitc_tiny_patchX(patch_size, **kwargs):
    return _vit(patch_size, 192, 11, 3, conv_stem=True, **kwargs)


def get_mae_vit(size='base', patch_size=None, c=False, **kwargs):
    if patch_size is None:
        patch_size = [16, 16]
    if c:
        table = {'base': mae_vitc_base_patchX, 'small': mae_vitc_small_patchX, 'tiny': mae_vitc_tiny_patchX}
        if size not in table:
            raise NotImplementedError(f'Size {size} is not supported')
        return table[size](patch_size, **kwargs)
    table = {'base': mae_vit_base_patchX, 'small': mae_vit_small_patchX, 'tiny': mae_vit_tiny_patchX, 'large': mae_vit_large_patchX}
    if size not in table:
        raise NotImplementedError(f'Size {size} is not supported')
    return table[size](patch_size, **kwargs)
