"""AudioNTT2022 encoder oracle (TEST INFRASTRUCTURE): restates model.py:130-191 (BYOL-A v2 encoder: two conv blocks, per-frame MLP,
stacking, mean + max pooling over time) with plain PyTorch CPU ops on an explicit state dict; the Dropout keep mask is an input."""
import torch
import torch.nn.functional as F

from . import rounding as R


def se_block(x, w1, w2):
    """SE_Block.forward (model.py:207-211): squeeze (global average pool), excitation (Linear -> ReLU -> Linear -> Sigmoid, no biases),
    channel-wise scale.  On the HIP path the pooled map arrives as bf16 and the gated map is stored as bf16 (R.qf at both ends); the gate
    itself is fp32."""
    x = R.qf(x)
    y = x.mean((2, 3))
    y = torch.sigmoid(F.linear(F.relu(F.linear(y, w1)), w2))
    return R.qf(x * y[:, :, None, None])


def encoder_frames(x, p, keep=None, drop_p=0.3, bn_stats=None, conv_layers=2):
    """AudioNTT2022Encoder.forward (model.py:156-165) in train mode -> [B, T', d].  keep: [B, T', mlp_hidden] 0/1 mask of nn.Dropout(0.3)
    (None = no dropout, i.e. eval-mode dropout with train-mode BatchNorm).  A state dict with `features.4.excitation.0.weight` is the
    `squeeze_excitation=True` network (model.py:141-151): an SE gate after each MaxPool, the second conv block one index later."""
    se = "features.4.excitation.0.weight" in p
    h = x
    for l in range(conv_layers):
        qf, qw = (R.qf, R.qw) if l > 0 else ((lambda t: t), (lambda t: t))     # first convolution (C_in = 1): direct fp32 kernel on the HIP path
        c = (5 if se else 4) * l                                                # index of this block's convolution in `features`
        h = R.qb(F.conv2d(qf(h), qw(p[f"features.{c}.weight"]), p[f"features.{c}.bias"], stride=1, padding=1))
        if bn_stats is not None:
            bn_stats.append((h.mean((0, 2, 3)).detach(), h.var((0, 2, 3), unbiased=False).detach(), h.numel() // h.shape[1]))
        h = F.relu(F.batch_norm(h, None, None, p[f"features.{c + 1}.weight"], p[f"features.{c + 1}.bias"], True, 0.1, 1e-5))
        h = F.max_pool2d(R.qf(h), 2, 2)
        if se:
            h = se_block(h, p[f"features.{c + 4}.excitation.0.weight"], p[f"features.{c + 4}.excitation.2.weight"])
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
