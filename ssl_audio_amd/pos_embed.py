"""Positional tables of the ViT/MAE encoder (host side, init-time constants, float64 numpy).

Mirrors models/pos_embed.py (get_2d_sincos_pos_embed :16-34, get_sinusoid_encoding_table :97-109) and the
per-forward bicubic resize of MaskedAutoencoderViT.interpolate_pos_encoding (models/mae.py:367-392), which
this package evaluates ONCE per input width and caches (the reference recomputes it every forward).
"""
import math

import numpy as np


def _sincos_1d(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_sizes, cls_token=True):
    gH, gW = grid_sizes
    gw, gh = np.meshgrid(np.arange(gW, dtype=np.float32), np.arange(gH, dtype=np.float32))  # "w goes first"
    emb = np.concatenate([_sincos_1d(embed_dim // 2, gw), _sincos_1d(embed_dim // 2, gh)], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def get_sinusoid_encoding_table(n_position, d_hid, cls_token=True):
    j = np.arange(d_hid)
    ang = np.arange(n_position, dtype=np.float64)[:, None] / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    tab = ang.copy()
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    if cls_token:
        tab = np.concatenate([np.zeros([1, d_hid]), tab], axis=0)
    return tab


def _cubic(t):
    A = -0.75
    c1 = lambda x: ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = lambda x: ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return np.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], 0)


def _taps(n_in, n_out, scale_factor):
    # F.interpolate(scale_factor=..., mode='bicubic', align_corners=False): src = (dst + 0.5) / scale - 0.5,
    # coordinate arithmetic in fp32 as PyTorch does, taps clamped to the grid
    inv = np.float32(1.0 / scale_factor)
    src = (inv * (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5)).astype(np.float64)
    fl = np.floor(src)
    idx = np.stack([np.clip(fl + k, 0, n_in - 1) for k in (-1, 0, 1, 2)], 0).astype(np.int64)
    return idx, _cubic(src - fl)


def interpolate_pos_encoding(pos_embed, grid, freq_bins, frames, patch=(16, 16)):
    """pos_embed [1, 1 + gh*gw, d] -> [1, 1 + nf*nt, d] for a freq_bins x frames input (models/mae.py:367-392)."""
    pos_embed = np.asarray(pos_embed, dtype=np.float64)
    gh, gw = grid
    nf, nt = freq_bins // patch[0], frames // patch[1]
    if nf * nt == gh * gw:
        return pos_embed
    d = pos_embed.shape[-1]
    P = pos_embed[0, 1:].reshape(gh, gw, d)
    sf, st = (nf + 0.1) / gh, (nt + 0.1) / gw
    assert int(math.floor(gh * sf)) == nf and int(math.floor(gw * st)) == nt
    iy, wy = _taps(gh, nf, sf)
    ix, wx = _taps(gw, nt, st)
    rows = sum(P[iy[k]] * wy[k][:, None, None] for k in range(4))
    out = sum(rows[:, ix[k]] * wx[k][None, :, None] for k in range(4))
    return np.concatenate([pos_embed[:, :1], out.reshape(1, nf * nt, d)], axis=1)


def interpolation_matrix(grid, nf, nt):
    """The bicubic resampling of interpolate_pos_encoding as a matrix A [nf * nt, gh * gw] (out = A @ patch table), WITHOUT the
    same-size shortcut: with `--use_learned_pos_embd` the reference resamples its table whenever the input is not square, even at the
    table's own grid (models/mae.py:370-375; the +0.1 makes that resampling differ from the identity), and the table's gradient is
    A^T applied to the summed token gradient."""
    gh, gw = grid
    sf, st = (nf + 0.1) / gh, (nt + 0.1) / gw
    assert int(math.floor(gh * sf)) == nf and int(math.floor(gw * st)) == nt
    iy, wy = _taps(gh, nf, sf)
    ix, wx = _taps(gw, nt, st)
    Ay, Ax = np.zeros((nf, gh)), np.zeros((nt, gw))
    for k in range(4):
        np.add.at(Ay, (np.arange(nf), iy[k]), wy[k])
        np.add.at(Ax, (np.arange(nt), ix[k]), wx[k])
    return np.kron(Ay, Ax)
