"""ResNet-18 encoder oracle (TEST INFRASTRUCTURE): restates models/resnet.py (BasicBlock :33-80, ResNet :141-274; the ResNet-C stem,
no ResNet-D, `fc` replaced by Identity as model.py:74-81 does) with plain PyTorch CPU ops on an explicit state dict, in train mode
(BatchNorm2d on batch statistics).  Two variants, as the reference's factories build them:

  resnet18            strides [2, 1, 2, 2, 2], global average pooling          -> [B, 512]     (models/resnet.py:277-292)
  resnet18_ReGP_NRF   strides [1, 1, 2, 2, (1, 2)], ReGP = max + mean over time of the (mel x channel) stack -> [B, 4096]  (:349-360)

Parity of this file with the reference is pinned by tests/test_oracle_golden.py against tests/golden/resnet.npz.  The rounding hooks
(oracle/rounding.py) are the identity unless the bf16 mirror mode is on."""
import torch
import torch.nn.functional as F

from . import rounding as R

VARIANTS = {
    "resnet18": dict(strides=[2, 1, 2, 2, 2], regp=False, embed_dim=512),
    "resnet18_ReGP_NRF": dict(strides=[1, 1, 2, 2, (1, 2)], regp=True, embed_dim=4096),
}
LAYERS, PLANES = [2, 2, 2, 2], [64, 128, 256, 512]


def _pair(s):
    return tuple(s) if isinstance(s, (tuple, list)) else (s, s)


_TRAINING = [True]          # forward(..., training=False): nn.BatchNorm2d in eval mode, i.e. the running statistics of the state dict


def _bn(h, p, name, bn_stats):
    if not _TRAINING[0]:
        return F.batch_norm(h, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"], p[name + ".bias"], False, 0.1, 1e-5)
    if bn_stats is not None:
        bn_stats.append((name, h.mean((0, 2, 3)).detach(), h.var((0, 2, 3), unbiased=False).detach(), h.numel() // h.shape[1]))
    return F.batch_norm(h, None, None, p[name + ".weight"], p[name + ".bias"], True, 0.1, 1e-5)


def _conv(a, p, name, stride, padding):
    # the HIP path stores the convolution's input activation and weight as bf16, and consumes the output gradient as bf16 -- except the
    # single-input-channel first convolution, a direct fp32 kernel (fp32 image, fp32 weight)
    if a.shape[1] == 1:
        return R.qb(F.conv2d(a, p[name + ".weight"], None, stride=_pair(stride), padding=padding))
    return R.qb(F.conv2d(R.qf(a), R.qw(p[name + ".weight"]), None, stride=_pair(stride), padding=padding))


def basic_block(x, p, pre, stride, bn_stats=None):
    """BasicBlock.forward (models/resnet.py:62-80); a `downsample.0.weight` key selects the 1x1-conv + BN identity path (:223-226)."""
    out = F.relu(_bn(_conv(x, p, pre + "conv1", stride, 1), p, pre + "bn1", bn_stats))
    out = _bn(_conv(out, p, pre + "conv2", 1, 1), p, pre + "bn2", bn_stats)
    identity = x
    if pre + "downsample.0.weight" in p:
        identity = _bn(_conv(x, p, pre + "downsample.0", stride, 0), p, pre + "downsample.1", bn_stats)
    return F.relu(out + identity)


def forward(x, p, variant="resnet18", bn_stats=None, training=True, layers=None):
    """ResNet._forward_impl (models/resnet.py:255-274) with fc = Identity.  x [B, 1, F, T] -> [B, embed_dim].  `layers`: blocks per stage
    (default the ResNet-18 [2, 2, 2, 2]; [0, 0, 0, 0] = stem + max-pool + head only, for tests that isolate the stem)."""
    _TRAINING[0] = training
    try:
        return _forward(x, p, variant, bn_stats, layers or LAYERS)
    finally:
        _TRAINING[0] = True


def _forward(x, p, variant, bn_stats, layers):
    cfg = VARIANTS[variant]
    s = cfg["strides"]
    h = x
    for l, st in enumerate([s[0], 1, 1]):                                  # ResNet-C stem: three 3x3 convolutions (:177-188)
        h = F.relu(_bn(_conv(h, p, f"conv1.{3 * l}", st, 1), p, f"conv1.{3 * l + 1}", bn_stats))
    h = F.max_pool2d(R.qf(h), kernel_size=3, stride=2, padding=1)
    for li in range(4):
        for b in range(layers[li]):
            h = basic_block(h, p, f"layer{li + 1}.{b}.", s[li + 1] if b == 0 else 1, bn_stats)
    if cfg["regp"]:
        h = h.permute(0, 3, 2, 1)                                          # (batch, time, mel, ch)
        B, T, D, C = h.shape
        h = h.reshape(B, T, C * D)
        return h.max(dim=1)[0] + h.mean(dim=1)
    return h.mean((2, 3))                                                   # AdaptiveAvgPool2d((1, 1)) + flatten


def init_state(variant="resnet18", seed=0, affine_seed=None):
    """A state dict with the reference's keys, shapes and initialisation scheme (kaiming-normal fan_out convolutions, BN weight 1 / bias 0,
    models/resnet.py:199-204) from an explicit generator -- for tests that need weights without the reference at hand.  affine_seed:
    BatchNorm weights 1 + 0.2 N(0,1), biases 0.1 N(0,1) instead, so that fixtures exercise the scale / shift paths."""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def conv(name, co, ci, k):
        std = (2.0 / (co * k * k)) ** 0.5
        p[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * std

    def bn(name, c):
        p[name + ".weight"], p[name + ".bias"] = torch.ones(c), torch.zeros(c)
        p[name + ".running_mean"], p[name + ".running_var"] = torch.zeros(c), torch.ones(c)

    for l, (ci, co) in enumerate([(1, 32), (32, 32), (32, 64)]):
        conv(f"conv1.{3 * l}", co, ci, 3)
        bn(f"conv1.{3 * l + 1}", co)
    s = VARIANTS[variant]["strides"]
    inpl = 64
    for li in range(4):
        pl = PLANES[li]
        for b in range(LAYERS[li]):
            pre = f"layer{li + 1}.{b}."
            stride = _pair(s[li + 1] if b == 0 else 1)
            conv(pre + "conv1", pl, inpl, 3); bn(pre + "bn1", pl)
            conv(pre + "conv2", pl, pl, 3); bn(pre + "bn2", pl)
            if b == 0 and (stride != (1, 1) or inpl != pl):
                conv(pre + "downsample.0", pl, inpl, 1); bn(pre + "downsample.1", pl)
            inpl = pl
    if affine_seed is not None:
        ga = torch.Generator().manual_seed(affine_seed)
        for k in p:                                  # (insertion order: deterministic)
            if p[k].dim() == 1 and k.endswith(".weight"):
                p[k] = 1.0 + 0.2 * torch.randn(p[k].shape, generator=ga)
            elif p[k].dim() == 1 and k.endswith(".bias"):
                p[k] = 0.1 * torch.randn(p[k].shape, generator=ga)
    return p
