"""AudioNTT2022 encoder oracle (TEST INFRASTRUCTURE): restates model.py:130-191 (BYOL-A v2 encoder: two conv blocks, per-frame MLP,
stacking, mean + max pooling over time) with plain PyTorch CPU ops on an explicit state dict; the Dropout keep mask is an input."""
import torch
import torch.nn.functional as F

from . import rounding as R


def encoder_frames(x, p, keep=None, drop_p=0.3, bn_stats=None, conv_layers=2):
    """AudioNTT2022Encoder.forward (model.py:156-165) in train mode -> [B, T', d].  keep: [B, T', mlp_hidden] 0/1 mask of nn.Dropout(0.3)
    (None = no dropout, i.e. eval-mode dropout with train-mode BatchNorm)."""
    h = x
    for l in range(conv_layers):
        qf, qw = (R.qf, R.qw) if l > 0 else ((lambda t: t), (lambda t: t))     # first convolution (C_in = 1): direct fp32 kernel on the HIP path
        h = R.qb(F.conv2d(qf(h), qw(p[f"features.{4 * l}.weight"]), p[f"features.{4 * l}.bias"], stride=1, padding=1))
        if bn_stats is not None:
            bn_stats.append((h.mean((0, 2, 3)).detach(), h.var((0, 2, 3), unbiased=False).detach(), h.numel() // h.shape[1]))
        h = F.relu(F.batch_norm(h, None, None, p[f"features.{4 * l + 1}.weight"], p[f"features.{4 * l + 1}.bias"], True, 0.1, 1e-5))
        h = F.max_pool2d(R.qf(h), 2, 2)
    h = h.permute(0, 3, 2, 1)                                   # (batch, time, mel, ch)
    B, T, D, C = h.shape
    h = h.reshape(B, T, C * D)
    f = F.relu(R.qb(F.linear(R.qf(h), R.qw(p["fc.0.weight"]), p["fc.0.bias"])))
    if keep is not None:
        f = f * keep / (1.0 - drop_p)
    f = F.relu(R.qb(F.linear(R.qf(f), R.qw(p["fc.3.weight"]), p["fc.3.bias"])))
    return torch.cat([h, f], dim=2)                             # hstack of the transposes == concatenation along the feature axis


def forward(x, p, keep=None, bn_stats=None):
    """AudioNTT2022.forward (model.py:174-177): frames -> mean_max_pooling (model.py:185-190)."""
    fr = encoder_frames(x, p, keep, bn_stats=bn_stats)
    return fr.max(dim=1)[0] + fr.mean(dim=1)
