#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING the reference (read-only at /root/reference).

Run in the build container only:  python tests/golden/make_golden.py
The GPU box never sees /root/reference; it only sees the .npz files this script writes.

Nothing from the reference is copied: this script imports its modules, feeds seeded inputs and
stores inputs + outputs (data only).  Three absent third-party names are stubbed exactly as
SURVEY.md §8(c)/F8 records (torchvision: imported-but-unused; timm DropPath/Mlp/to_2tuple:
standard definitions; np.float alias removed in numpy>=1.24).
"""
import os
import sys
import types
import random
from functools import partial

import numpy as np

np.float = float  # models/pos_embed.py:52 uses the removed alias

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SSL_AUDIO_REFERENCE", "/root/reference")


def _install_stubs():
    tv = types.ModuleType("torchvision")
    tvd = types.ModuleType("torchvision.datasets")
    tvt = types.ModuleType("torchvision.transforms")
    tv.datasets, tv.transforms = tvd, tvt
    sys.modules.update({"torchvision": tv, "torchvision.datasets": tvd, "torchvision.transforms": tvt})

    class DropPath(nn.Module):  # rate is always 0 on this path -> identity
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    class Mlp(nn.Module):  # timm.models.vision_transformer.Mlp: fc1 -> act -> fc2
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.drop1 = nn.Dropout(drop)
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop2 = nn.Dropout(drop)

        def forward(self, x):
            return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))

    names = ["timm", "timm.models", "timm.models.vision_transformer", "timm.models.layers",
             "timm.models.layers.helpers"]
    mods = {n: types.ModuleType(n) for n in names}
    mods["timm.models.vision_transformer"].DropPath = DropPath
    mods["timm.models.vision_transformer"].Mlp = Mlp
    mods["timm.models.layers.helpers"].to_2tuple = lambda x: tuple(x) if isinstance(x, (list, tuple)) else (x, x)
    sys.modules.update(mods)


_install_stubs()
sys.path.insert(0, REF)

import augmentations as ref_aug  # noqa: E402
import model as ref_model  # noqa: E402
from utils import loss as ref_loss, utils as ref_utils, transforms as ref_transforms  # noqa: E402
from utils import hyperparameters as ref_hp  # noqa: E402
from models import mae as ref_mae, pos_embed as ref_pos  # noqa: E402


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def t2n(t):
    return t.detach().cpu().clone().numpy()  # clone: later in-place updates must not alias a saved vector


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def cfg_ns(**kw):
    a = ref_hp.get_std_parameters().parse_args([])
    for k, v in kw.items():
        setattr(a, k, v)
    return a


# ----------------------------------------------------------------------------- BT loss
def gen_bt_loss():
    out = {}
    for tag, B, D, hsic in [("anchor", 64, 256, False), ("hsic", 64, 256, True), ("cfg1", 32, 256, False),
                            ("ragged", 7, 24, False)]:
        torch.manual_seed(0)
        z1 = torch.randn(B, D)
        z2 = z1 + 0.1 * torch.randn(B, D)
        z1.requires_grad_(True)
        z2.requires_grad_(True)
        cfg = cfg_ns(projector_out_dim=D, alpha=1.0, lmbda=0.005, HSIC=hsic)
        crit = ref_loss.BarlowTwinsLoss(cfg, ncrops=2)
        l = crit.forward_loss(z1, z2)
        l.backward()
        out.update({f"{tag}_z1": t2n(z1), f"{tag}_z2": t2n(z2), f"{tag}_loss": t2n(l),
                    f"{tag}_dz1": t2n(z1.grad), f"{tag}_dz2": t2n(z2.grad),
                    f"{tag}_running_mean": t2n(crit.bn.running_mean), f"{tag}_running_var": t2n(crit.bn.running_var),
                    f"{tag}_nbt": t2n(crit.bn.num_batches_tracked), f"{tag}_hsic": np.array(int(hsic))})
    # BarlowTwinsLoss.forward crop bookkeeping: (g=1, L=0), (g=1, L=1), (g=2, L=0)
    for tag, g, L in [("g1L0", 1, 0), ("g1L1", 1, 1), ("g2L0", 2, 0)]:
        torch.manual_seed(3)
        B, D = 16, 32
        ncrops = L + 2
        n_student = ncrops - (2 - g)
        student = torch.randn(n_student * B, D, requires_grad=True)
        teacher = torch.randn(g * B, D, requires_grad=True)
        cfg = cfg_ns(projector_out_dim=D, alpha=1.0, lmbda=0.005, HSIC=False)
        crit = ref_loss.BarlowTwinsLoss(cfg, ncrops=ncrops)
        l = crit(student, teacher, ngcrops_each=g)
        l.backward()
        out.update({f"fwd_{tag}_student": t2n(student), f"fwd_{tag}_teacher": t2n(teacher),
                    f"fwd_{tag}_loss": t2n(l), f"fwd_{tag}_dstudent": t2n(student.grad),
                    f"fwd_{tag}_dteacher": t2n(teacher.grad), f"fwd_{tag}_ncrops": np.array(ncrops),
                    f"fwd_{tag}_g": np.array(g),
                    f"fwd_{tag}_running_mean": t2n(crit.bn.running_mean),
                    f"fwd_{tag}_running_var": t2n(crit.bn.running_var)})
    x = torch.arange(25.0).reshape(5, 5)
    out["offdiag_in"] = t2n(x)
    out["offdiag_out"] = t2n(ref_utils.off_diagonal(x))
    save("bt_loss", **out)


# ----------------------------------------------------------------------------- augmentations
class _Recorder:
    """Wraps RandomResizeCrop.get_params to record the (i, j, h, w) actually drawn."""

    def __init__(self):
        self.params = []
        self._orig = ref_aug.RandomResizeCrop.get_params

    def __enter__(self):
        orig = self._orig

        def rec(*a, **k):
            p = orig(*a, **k)
            self.params.append([int(v) for v in p])
            return p

        ref_aug.RandomResizeCrop.get_params = staticmethod(rec)
        return self

    def __exit__(self, *a):
        ref_aug.RandomResizeCrop.get_params = staticmethod(self._orig)


def gen_augment():
    out = {}
    # log_mixup_exp
    torch.manual_seed(1)
    xa, xb = torch.randn(1, 64, 96) * 2 - 1, torch.randn(1, 64, 96) * 2 - 1
    for k, alpha in enumerate([0.0, 0.13, 0.2, 1.0]):
        out[f"lme_{k}_alpha"] = np.array(alpha)
        out[f"lme_{k}_out"] = t2n(ref_aug.log_mixup_exp(xa, xb, alpha))
    out["lme_xa"], out["lme_xb"] = t2n(xa), t2n(xb)

    # RandomResizeCrop with recorded params: T=96 (default) and T=1001 (10 s), plus local-crop geometry
    for tag, F_, T_, out_size, vcs, fs, ts, seed in [
        ("t96", 64, 96, (64, 96), (1.0, 1.5), (0.6, 1.5), (0.6, 1.5), 123),
        ("t96b", 64, 96, (64, 96), (1.0, 1.5), (0.6, 1.5), (0.6, 1.5), 7),
        ("t1001", 64, 1001, (64, 1001), (1.0, 1.5), (0.6, 1.5), (0.6, 1.5), 11),
        ("t1001b", 64, 1001, (64, 1001), (1.0, 1.5), (0.6, 1.5), (0.6, 1.5), 12),
        ("local", 64, 96, (16, 16), (1.0, 1.0), (0.05, 0.6), (0.05, 0.6), 5),
    ]:
        seed_all(seed)
        x = torch.randn(1, F_, T_)
        rrc = ref_aug.RandomResizeCrop(out_size, virtual_crop_scale=vcs, freq_scale=fs, time_scale=ts)
        with _Recorder() as r:
            y = rrc(x)
        out[f"rrc_{tag}_x"] = t2n(x)
        out[f"rrc_{tag}_y"] = t2n(y)
        out[f"rrc_{tag}_params"] = np.array(r.params[0])
        out[f"rrc_{tag}_cfg"] = np.array([out_size[0], out_size[1], vcs[0], vcs[1], fs[0], fs[1], ts[0], ts[1], seed],
                                         dtype=np.float64)

    # RandomLinearFader
    seed_all(9)
    x = torch.randn(1, 64, 96)
    st = np.random.get_state()
    ht = 1.0 * ((2.0 * np.random.rand(2)) - 1.0)
    np.random.set_state(st)
    y = ref_aug.RandomLinearFader()(x)
    out["rlf_x"], out["rlf_y"], out["rlf_head_tail"] = t2n(x), t2n(y), ht

    # NormalizeBatch
    torch.manual_seed(2)
    X = torch.randn(5, 1, 64, 96) * 3 + 1
    out["nb_x"], out["nb_y"] = t2n(X), t2n(ref_aug.NormalizeBatch()(X))

    # Whole AudioPairTransform sequence: 5 consecutive clips (bank evolution, RNG call order)
    for tag, T_, L in [("seq96", 96, 0), ("seq208", 208, 0), ("seq96_local", 96, 2)]:
        seed_all(42)
        args = cfg_ns(crop_frames=T_, local_crops_number=L)
        tfm = ref_transforms.AudioPairTransform(args)
        clips = torch.randn(5, 1, 64, T_) * 1.3 - 0.2
        views, locals_ = [], []
        with _Recorder() as r:
            for c in clips:
                crops = tfm(c)
                views.append(torch.stack(crops[:2]))
                if L:
                    locals_.append(torch.stack(crops[2:]))
        out[f"apt_{tag}_clips"] = t2n(clips)
        out[f"apt_{tag}_views"] = t2n(torch.stack(views))          # [5, 2, 1, 64, T]
        if L:
            out[f"apt_{tag}_locals"] = t2n(torch.stack(locals_))   # [5, L, 1, 16, 16]
        out[f"apt_{tag}_rrc_params"] = np.array(r.params)          # draws in call order
        out[f"apt_{tag}_seed"] = np.array(42)
        out[f"apt_{tag}_L"] = np.array(L)
    save("augment", **out)


# ----------------------------------------------------------------------------- ViT (micro) + pos-embed
def micro_vit(use_decoder=False, img_size=(64, 96), embed_dim=128, depth=2, heads=2):
    return ref_mae.MaskedAutoencoderViT(
        img_size=img_size, patch_size=[16, 16], in_chans=1, embed_dim=embed_dim, depth=depth, num_heads=heads,
        mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), use_decoder=use_decoder,
        decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=1)


def perturb_(m, seed):
    """Make every parameter (biases, LN affine, k-less qkv bias) non-trivial so parity tests see them."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "pos_embed" in n:
                continue
            if p.dim() == 1 or n.endswith("cls_token") or n.endswith("mask_token"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))


def gen_vit():
    out = {}
    torch.manual_seed(0)
    m = micro_vit()
    perturb_(m, 1)
    m.eval()
    for k, v in m.state_dict().items():
        out["sd." + k] = t2n(v)
    for tag, T_ in [("t96", 96), ("t208", 208), ("t1001", 1001)]:
        torch.manual_seed(10)
        x = torch.randn(2, 1, 64, T_)
        x.requires_grad_(True)
        lat = m(x)
        out[f"{tag}_x"], out[f"{tag}_latent"] = t2n(x), t2n(lat)
        tok, _, _ = m.prepare_tokens(x, 0)
        out[f"{tag}_tokens"] = t2n(tok)
        out[f"{tag}_pos"] = t2n(m.interpolate_pos_encoding(m.patch_embed(x), 64, T_))
        out[f"{tag}_latent_meanpool"] = t2n(m(x, mean_pool=True))
        full, _, _ = m.forward_encoder(x, 0)
        out[f"{tag}_encoded"] = t2n(full)
        # backward through the encoder: grads of a few named params for sum(latent * w)
        m.zero_grad()
        w = torch.linspace(-1, 1, lat.numel()).reshape(lat.shape)
        (lat * w).sum().backward()
        for pn in ["cls_token", "blocks.0.attn.qkv.weight", "blocks.0.attn.q_bias", "blocks.0.attn.v_bias",
                   "blocks.1.mlp.fc1.weight", "blocks.1.mlp.fc2.bias", "blocks.0.norm1.weight", "norm.bias",
                   "blocks.1.attn.proj.weight"]:
            out[f"{tag}_grad.{pn}"] = t2n(dict(m.named_parameters())[pn].grad)
    # masked forward with a prefixed mask (models/mae.py:317-323) -> deterministic
    torch.manual_seed(11)
    x = torch.randn(3, 1, 64, 96)
    mask = torch.zeros(3, 24)
    for b in range(3):
        mask[b, torch.randperm(24)[:18]] = 1  # 75 % masked, 6 kept
    lat = m(x, mask_ratio=mask)
    out["mask_x"], out["mask_mask"], out["mask_latent"] = t2n(x), t2n(mask), t2n(lat)
    xm, mk, ids = m.random_masking(m.patch_embed(x), mask)
    out["mask_ids_restore"] = t2n(ids)
    out["mask_out_mask"] = t2n(mk)
    # masked forward with the internal torch.rand noise: capture by seeding
    torch.manual_seed(77)
    noise = torch.rand(3, 24)
    torch.manual_seed(77)
    lat2 = m(x, mask_ratio=0.75)
    out["rand_noise"], out["rand_latent"] = t2n(noise), t2n(lat2)
    # positional tables at real sizes (init-time constants)
    out["sincos_192_4x6"] = ref_pos.get_2d_sincos_pos_embed(192, (4, 6)).astype(np.float32)
    out["sincos_768_4x6"] = ref_pos.get_2d_sincos_pos_embed(768, (4, 6)).astype(np.float32)
    out["sinusoid_24_384"] = ref_pos.get_sinusoid_encoding_table(24, 384).astype(np.float32)
    save("vit_micro", **out)

    # MAE decoder + recon loss (reference path T=96, and cfg-5 style img_size=(64, 992) scaled down to (64,208))
    out = {}
    for tag, img in [("t96", (64, 96)), ("t208", (64, 208))]:
        torch.manual_seed(5)
        md = micro_vit(use_decoder=True, img_size=img)
        perturb_(md, 2)
        for k, v in md.state_dict().items():
            out[f"{tag}_sd." + k] = t2n(v)
        L = (img[0] // 16) * (img[1] // 16)
        torch.manual_seed(6)
        x = torch.randn(2, 1, img[0], img[1])
        mask = torch.zeros(2, L)
        keep = int(L * 0.25)
        for b in range(2):
            mask[b, torch.randperm(L)[:L - keep]] = 1
        lat, rl = md(x, mask_ratio=mask, masked_recon=True)
        md.zero_grad()
        (rl + lat.sum() * 0.01).backward()
        out[f"{tag}_x"], out[f"{tag}_mask"] = t2n(x), t2n(mask)
        out[f"{tag}_latent"], out[f"{tag}_recon_loss"] = t2n(lat), t2n(rl)
        out[f"{tag}_patchify"] = t2n(md.patchify(x))
        for pn in ["mask_token", "decoder_embed.weight", "decoder_pred.bias", "decoder_blocks.0.attn.qkv.weight",
                   "blocks.0.mlp.fc1.weight"]:
            out[f"{tag}_grad.{pn}"] = t2n(dict(md.named_parameters())[pn].grad)
    save("mae_micro", **out)


# ----------------------------------------------------------------------------- head / predictor
def gen_head():
    out = {}
    cfg = cfg_ns(projector_hidden_dim=192, projector_out_dim=64, projector_n_hidden_layers=1)
    torch.manual_seed(0)
    head = ref_model.BarlowTwinsHead(cfg, in_dim=128)
    with torch.no_grad():
        head.projector[1].weight.add_(0.2 * torch.randn(192))
        head.projector[1].bias.add_(0.2 * torch.randn(192))
    for k, v in head.state_dict().items():
        out["head_sd." + k] = t2n(v)
    x = torch.randn(2 * 12, 128, requires_grad=True)
    z = head(x, ncrops=2)
    w = torch.randn_like(z)
    (z * w).sum().backward()
    out.update(head_x=t2n(x), head_z=t2n(z), head_w=t2n(w), head_dx=t2n(x.grad))
    for n, p in head.named_parameters():
        out["head_grad." + n] = t2n(p.grad)
    for k, v in head.state_dict().items():
        out["head_sd_after." + k] = t2n(v)
    torch.manual_seed(1)
    pred = ref_model.BarlowTwinsPredictor(64, use=True)
    for k, v in pred.state_dict().items():
        out["pred_sd." + k] = t2n(v)
    x = torch.randn(2 * 12, 64, requires_grad=True)
    z = pred(x, ncrops=1)
    w = torch.randn_like(z)
    (z * w).sum().backward()
    out.update(pred_x=t2n(x), pred_z=t2n(z), pred_w=t2n(w), pred_dx=t2n(x.grad))
    for n, p in pred.named_parameters():
        out["pred_grad." + n] = t2n(p.grad)
    save("head", **out)


# ----------------------------------------------------------------------------- one full training step
class _MicroBackbone(nn.Module):
    """ModelWrapper-shaped backbone around the micro ViT (ModelWrapper itself hard-codes tiny/small/base)."""

    def __init__(self):
        super().__init__()
        self.encoder = micro_vit()
        self.feature_dim = self.encoder.embed_dim

    def forward(self, x, mask_ratio=0, masked_recon=False):
        return self.encoder(x, mask_ratio=mask_ratio, masked_recon=masked_recon)


def gen_step():
    """main_bt_byol.py:79-135 with --stop_gradient --predictor, and the plain two-net variant, B=8, T=96."""
    for tag, stop_grad, use_pred in [("byol", True, True), ("plain", False, False)]:
        out = {}
        cfg = cfg_ns(projector_hidden_dim=192, projector_out_dim=64, model_type="vit_tiny", batch_size=8)
        ref_hp.setup_hyperparameters(cfg)
        torch.manual_seed(0)
        online = ref_utils.MultiCropWrapper(_MicroBackbone(), ref_model.BarlowTwinsHead(cfg, 128))
        perturb_(online, 3)
        predictor = ref_model.BarlowTwinsPredictor(64, use=use_pred)
        torch.manual_seed(1)
        target = ref_utils.MultiCropWrapper(_MicroBackbone(), ref_model.BarlowTwinsHead(cfg, 128))
        target.load_state_dict(online.state_dict())
        if stop_grad:
            for p in target.parameters():
                p.requires_grad = False
        crit = ref_loss.BarlowTwinsLoss(cfg, ncrops=2)
        groups = ref_utils.get_param_groups(online)
        if use_pred:
            groups += ref_utils.get_param_groups(predictor)
        if not stop_grad:
            groups += ref_utils.get_param_groups(target)
        opt = torch.optim.AdamW(groups, lr=cfg.lr, weight_decay=cfg.wd)
        for k, v in online.state_dict().items():
            out["online_sd." + k] = t2n(v)
        for k, v in predictor.state_dict().items():
            out["pred_sd." + k] = t2n(v)
        torch.manual_seed(4)
        images = [torch.randn(8, 1, 64, 96), torch.randn(8, 1, 64, 96)]
        out["view0"], out["view1"] = t2n(images[0]), t2n(images[1])
        ema = ref_utils.EMA(0.99)
        losses = []
        for it in range(2):
            o = online(images[:2], ncrops=2)
            o = predictor(o, ncrops=1)
            t = target(images, ncrops=2)
            l = crit(o, t, ngcrops_each=2)
            losses.append(float(l))
            if stop_grad:
                ref_utils.update_moving_average(ema, target, online)
            opt.zero_grad()
            l.backward()
            if it == 0:
                for pn in ["backbone.encoder.cls_token", "backbone.encoder.blocks.0.attn.qkv.weight",
                           "backbone.encoder.blocks.1.mlp.fc2.weight", "head.projector.0.weight",
                           "head.projector.1.weight", "head.projector.3.weight",
                           "backbone.encoder.norm.weight"]:
                    out["grad0." + pn] = t2n(dict(online.named_parameters())[pn].grad)
            opt.step()
        out["losses"] = np.array(losses)
        out["lr"], out["wd"] = np.array(cfg.lr), np.array(cfg.wd)
        keep = lambda k: ("blocks.0." in k) or k.startswith("head.") or ("norm." in k and "blocks" not in k) or k.endswith("cls_token")
        for k, v in online.state_dict().items():
            if keep(k):
                out["online_sd_after." + k] = t2n(v)
        for k, v in target.state_dict().items():
            if keep(k):
                out["target_sd_after." + k] = t2n(v)
        for k, v in crit.state_dict().items():
            out["crit_sd_after." + k] = t2n(v)
        save(f"step_{tag}", **out)


# ----------------------------------------------------------------------------- MultiCropWrapper grouping, EMA, param groups
def gen_misc():
    out = {}

    class Bk(nn.Module):
        def forward(self, x):
            return x.mean(dim=(1, 2)).unsqueeze(1) * torch.ones(1, 3) + x.shape[-1]

    class Hd(nn.Module):
        def forward(self, x, ncrops):
            return x * ncrops

    mc = ref_utils.MultiCropWrapper(Bk(), Hd())
    torch.manual_seed(0)
    xs = [torch.randn(2, 4, 8), torch.randn(2, 4, 8), torch.randn(2, 4, 5), torch.randn(2, 4, 5), torch.randn(2, 4, 5)]
    out["mc_out"] = t2n(mc(xs, ncrops=5))
    for i, x in enumerate(xs):
        out[f"mc_x{i}"] = t2n(x)
    a, b = nn.Linear(4, 3), nn.Linear(4, 3)
    out["ema_old_w"], out["ema_new_w"] = t2n(a.weight), t2n(b.weight)
    ref_utils.update_moving_average(ref_utils.EMA(0.99), a, b)
    out["ema_out_w"] = t2n(a.weight)
    torch.manual_seed(0)
    online = ref_utils.MultiCropWrapper(_MicroBackbone(), ref_model.BarlowTwinsHead(
        cfg_ns(projector_hidden_dim=192, projector_out_dim=64), 128))
    groups = ref_utils.get_param_groups(online)
    names = {id(p): n for n, p in online.named_parameters()}
    out["pg_regularized"] = np.array([names[id(p)] for p in groups[0]["params"]])
    out["pg_not_regularized"] = np.array([names[id(p)] for p in groups[1]["params"]])
    save("misc", **out)


def gen_optim():
    """LARS (utils/utils.py:150-189, group layout of main_bt_byol.py:326-345) over three steps, and the three schedule helpers."""
    out = {}
    torch.manual_seed(11)
    w = nn.Parameter(torch.randn(7, 5)); b = nn.Parameter(torch.randn(5)); z = nn.Parameter(torch.zeros(3, 3))
    opt = ref_utils.LARS([{"params": [w, z], "lr": 0.2}, {"params": [b], "lr": 0.0048}], lr=0, weight_decay=1.5e-2,
                         weight_decay_filter=True, lars_adaptation_filter=True)
    out["lars_w0"], out["lars_b0"], out["lars_z0"] = t2n(w).copy(), t2n(b).copy(), t2n(z).copy()
    for it in range(3):
        gw, gb, gz = torch.randn(7, 5), torch.randn(5), torch.randn(3, 3) * (1.0 if it else 0.0)   # step 0: zero p AND zero g on z
        w.grad, b.grad, z.grad = gw.clone(), gb.clone(), gz.clone()
        out[f"lars_gw{it}"], out[f"lars_gb{it}"], out[f"lars_gz{it}"] = t2n(gw), t2n(gb), t2n(gz)
        opt.step()
        out[f"lars_w{it + 1}"], out[f"lars_b{it + 1}"], out[f"lars_z{it + 1}"] = t2n(w).copy(), t2n(b).copy(), t2n(z).copy()
    out["lars_cfg"] = np.array([0.2, 0.0048, 1.5e-2, 0.9, 0.001])       # lr weights, lr biases, wd, momentum, eta
    out["cos_sched"] = ref_utils.cosine_scheduler(0.5, 0.01, 5, 7, warmup_epochs=2, start_warmup_value=0.1)
    out["cos_sched_nowarm"] = ref_utils.cosine_scheduler(1.0, 0.0, 3, 4)
    out["sine_sched"] = ref_utils.sine_scheduler_increase(0.75, 4, 6, warmup_epochs=1, warmup_value=0.2)

    class _Opt:
        def __init__(self, n):
            self.param_groups = [{"lr": -1.0} for _ in range(n)]

    loader = list(range(13))
    for name, optname, ngroups in [("adamw", "AdamW", 3), ("lars", "LARS", 2)]:
        args = cfg_ns(epochs=300, batch_size=256, lr=1e-4, lr_weights=0.2, lr_biases=0.0048, optimizer=optname)
        o = _Opt(ngroups)
        rows = []
        for step in [0, 1, 20, 38, 39, 40, 500, 2000, 4000, 4874]:
            ref_utils.adjust_learning_rate(args, o, loader, step)
            rows.append([step] + [g["lr"] for g in o.param_groups])
        out[f"adjust_lr_{name}"] = np.array(rows, dtype=np.float64)
    save("optim", **out)


def gen_noise_norm():
    """MixGaussianNoise (draws captured by re-seeding torch) and RunningNorm over five samples with a two-sample update budget."""
    out = {}
    torch.manual_seed(3)
    x = torch.randn(1, 64, 96) * 2.0 - 1.0
    np.random.seed(5)
    torch.manual_seed(9)
    y = ref_aug.MixGaussianNoise(ratio=0.2)(x)
    np.random.seed(5)
    lambd = 0.2 * np.random.rand()
    torch.manual_seed(9)
    z = torch.normal(0, lambd, x.shape)
    out["gn_x"], out["gn_y"], out["gn_lambda"], out["gn_normal"] = t2n(x), t2n(y), np.float64(lambd), t2n(z / lambd)
    rn = ref_aug.RunningNorm(epoch_samples=1, max_update_epochs=3)
    torch.manual_seed(4)
    for i in range(5):
        img = torch.randn(1, 64, 40) * (1.0 + i) + 0.5 * i
        out[f"rn_x{i}"] = t2n(img)
        out[f"rn_y{i}"] = t2n(rn(img))
    out["rn_mean"], out["rn_std"] = t2n(rn.mean), t2n(rn.std)
    save("noise_norm", **out)


def gen_eval():
    """utils.encode_vit (utils/utils.py:278-314) on the micro ViT of vit_micro.npz (weights are NOT stored again)."""
    g = np.load(os.path.join(HERE, "vit_micro.npz"))
    m = micro_vit()
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    m.eval()
    out = {}
    torch.manual_seed(21)
    for tag, T_ in [("t222", 222), ("t192", 192), ("t50", 50)]:      # ragged, exact multiple (pads a whole unit), shorter than a unit
        x = torch.randn(2, 1, 64, T_)
        out[f"{tag}_x"] = t2n(x)
        with torch.no_grad():
            out[f"{tag}_cls"] = t2n(ref_utils.encode_vit(m, x, split_frames=True, use_cls=True))
            out[f"{tag}_patch"] = t2n(ref_utils.encode_vit(m, x, split_frames=True, use_cls=False))
            out[f"{tag}_whole"] = t2n(ref_utils.encode_vit(m, x, split_frames=False))
    save("eval", **out)


def gen_convstem():
    """ConvStem ViTC (models/mae.py:46-99, conv_stem=True): micro encoder (d=128, 2 blocks, 2 heads; stem channels 16/32/64/128) with
    16x16 and 16x8 patches, train mode (BatchNorm2d on batch statistics): tokens, latent, gradients of every stem parameter, and the
    BatchNorm running statistics after the forward."""
    out = {}
    for tag, patch, T_ in [("p16x16_t96", [16, 16], 96), ("p16x8_t96", [16, 8], 96), ("p16x16_t208", [16, 16], 208)]:
        torch.manual_seed(0)
        m = ref_mae.MaskedAutoencoderViT(img_size=(64, 96), patch_size=patch, in_chans=1, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4,
                                         norm_layer=partial(nn.LayerNorm, eps=1e-6), conv_stem=True)
        perturb_(m, 4)
        m.train()
        for k, v in m.state_dict().items():
            out[f"{tag}_sd." + k] = t2n(v)
        torch.manual_seed(12)
        x = torch.randn(3, 1, 64, T_)
        tok, _, _ = m.prepare_tokens(x, 0)
        out[f"{tag}_x"], out[f"{tag}_tokens"] = t2n(x), t2n(tok)
        m.load_state_dict({k[len(f"{tag}_sd."):]: torch.from_numpy(v) for k, v in out.items() if k.startswith(f"{tag}_sd.")})   # undo the BN update
        lat = m(x)
        out[f"{tag}_latent"] = t2n(lat)
        m.zero_grad()
        w = torch.linspace(-1, 1, lat.numel()).reshape(lat.shape)
        (lat * w).sum().backward()
        for n, prm in m.named_parameters():
            if n.startswith("patch_embed.") or n in ("cls_token", "blocks.0.attn.qkv.weight", "blocks.1.mlp.fc2.weight"):
                out[f"{tag}_grad.{n}"] = t2n(prm.grad)
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[f"{tag}_after." + k] = t2n(v)
        out[f"{tag}_patch"] = np.array(patch)
    save("convstem", **out)


def gen_convstem_lpe():
    """ConvStem ViTC WITH `--use_learned_pos_embd` (models/mae.py:186-199): the trained table is resampled by the reference's bicubic map for every
    non-square input (:367-392 -- also at the table's own grid, 64 x 96) and its gradient flows back through that map.  Micro encoder in
    train mode, T = 96 (same grid, resampled) and T = 208 (13 columns out of 6): latent, table / CLS / last-conv gradients."""
    out = {}
    for tag, T_ in [("t96", 96), ("t208", 208)]:
        torch.manual_seed(31)
        m = ref_mae.MaskedAutoencoderViT(img_size=(64, 96), patch_size=[16, 16], in_chans=1, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4,
                                         norm_layer=partial(nn.LayerNorm, eps=1e-6), conv_stem=True, use_learned_pos_embd=True)
        perturb_(m, 32)
        with torch.no_grad():
            m.pos_embed.add_(0.05 * torch.randn(m.pos_embed.shape))
        assert m.pos_embed.requires_grad
        m.train()
        if tag == "t96":                                        # (the same seeds give both cases the same weights: stored once)
            for k, v in m.state_dict().items():
                out["sd." + k] = t2n(v)
        torch.manual_seed(33)
        x = torch.randn(3, 1, 64, T_)
        lat = m(x)
        w = torch.linspace(-1, 1, lat.numel()).reshape(lat.shape)
        m.zero_grad()
        (lat * w).sum().backward()
        out.update({f"{tag}_x": t2n(x), f"{tag}_latent": t2n(lat), f"{tag}_dpos": t2n(m.pos_embed.grad), f"{tag}_dcls": t2n(m.cls_token.grad)})
        last = [n for n, _ in m.named_parameters() if n.startswith("patch_embed.proj.") and n.endswith(".weight")][-1]
        out[f"{tag}_dlast"] = t2n(dict(m.named_parameters())[last].grad)
        out[f"{tag}_last_name"] = np.array(last)
    save("convstem_lpe", **out)


def gen_audiontt():
    """AudioNTT2022 (model.py:130-191) in train mode: BatchNorm2d on batch statistics, Dropout(0.3) with its mask recovered from a
    forward hook (input / output of the Dropout module), small MLP widths so the fixture stays small (n_mels 64, d 1280, hidden 256)."""
    out = {}
    torch.manual_seed(0)
    m = ref_model.AudioNTT2022(n_mels=64, d=1280, mlp_hidden_d=256)
    m.train()
    for k, v in m.state_dict().items():
        out["sd." + k] = t2n(v)
    rec = {}
    hook = m.fc[2].register_forward_hook(lambda mod, inp, o: rec.update(inp=inp[0].detach().clone(), out=o.detach().clone()))
    torch.manual_seed(5)
    x = torch.randn(3, 1, 64, 40)
    y = m(x)
    hook.remove()
    keep = ((rec["out"] != 0) | (rec["inp"] == 0)).float()        # where the input is 0 the draw is unobservable (and irrelevant)
    w = torch.linspace(-1, 1, y.numel()).reshape(y.shape)
    m.zero_grad()
    (y * w).sum().backward()
    out.update(x=t2n(x), y=t2n(y), keep=t2n(keep))
    for n, prm in m.named_parameters():
        out["grad." + n] = t2n(prm.grad)
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            out["after." + k] = t2n(v)
    save("audiontt", **out)


def gen_resnet():
    """ResNet-18 encoders (models/resnet.py) in train mode, `fc` replaced by Identity as model.py:74-81 does: `resnet18` and
    `resnet18_ReGP_NRF`.  The 11 M weights are not stored: they come from oracle.resnet.init_state(variant, seed) (the reference's
    initialisation scheme from an explicit generator) and are LOADED into the reference model with strict key checking, so the fixture
    pins key names and shapes too.  Stored: input, embedding, gradients of the small parameters, the gradient norm of every parameter,
    BatchNorm buffers after the step."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import resnet as oresnet
    from models import resnet as ref_resnet
    out = {}
    for variant, ctor in [("resnet18", ref_resnet.resnet18), ("resnet18_ReGP_NRF", ref_resnet.resnet18_ReGP_NRF)]:
        m = ctor()
        m.fc = nn.Identity()
        sd = oresnet.init_state(variant, seed=3, affine_seed=11)
        full = dict(sd)
        for k in list(m.state_dict().keys()):
            if k.endswith("num_batches_tracked"):
                full[k] = torch.zeros((), dtype=torch.long)
        m.load_state_dict(full, strict=True)
        m.train()
        torch.manual_seed(21)
        x = torch.randn(4, 1, 64, 96) * 1.3 + 0.2
        y = m(x)
        w = torch.linspace(-1, 1, y.numel()).reshape(y.shape)
        m.zero_grad()
        (y * w).sum().backward()
        out[f"{variant}.x"], out[f"{variant}.y"] = t2n(x), t2n(y)
        names, norms = [], []
        for n, prm in m.named_parameters():
            names.append(n)
            norms.append(float(prm.grad.double().norm()))
            if prm.numel() <= 40000:
                out[f"{variant}.grad.{n}"] = t2n(prm.grad)
        out[f"{variant}.grad_names"] = np.array(names)
        out[f"{variant}.grad_norms"] = np.array(norms)
        for k, v in m.state_dict().items():
            if ("running" in k or "num_batches" in k) and (k.startswith("conv1.") or k.startswith("layer1.0.") or k.startswith("layer4.1.") or "downsample" in k):
                out[f"{variant}.after.{k}"] = t2n(v)
        out[f"{variant}.affine_seed"] = np.array([3, 11])
    save("resnet", **out)


def gen_bn_eval():
    """Eval mode of the BatchNorm2d encoders (`model.eval()`: running statistics, buffers untouched) -- the HEAR / linear-probe path.
    ConvStem ViTC micro encoder (16x8 patches) and both ResNet-18 variants: one train-mode forward first, so the running buffers are
    not their initial 0 / 1, then eval on a second batch.  Stored: every buffer after the train forward, the eval input and output."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import resnet as oresnet
    from models import resnet as ref_resnet
    out = {}
    torch.manual_seed(0)
    m = ref_mae.MaskedAutoencoderViT(img_size=(64, 96), patch_size=[16, 8], in_chans=1, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4,
                                     norm_layer=partial(nn.LayerNorm, eps=1e-6), conv_stem=True)
    perturb_(m, 4)
    m.train()
    torch.manual_seed(12)
    m(torch.randn(3, 1, 64, 96) * 1.5 + 0.3)
    m.eval()
    for k, v in m.state_dict().items():
        out["vitc_sd." + k] = t2n(v)
    torch.manual_seed(13)
    x = torch.randn(2, 1, 64, 96)
    with torch.no_grad():
        tok, _, _ = m.prepare_tokens(x, 0)
        lat = m(x)
    out["vitc_x"], out["vitc_tokens"], out["vitc_latent"] = t2n(x), t2n(tok), t2n(lat)
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert np.array_equal(t2n(v), out["vitc_sd." + k])            # eval left the buffers alone
    for variant, ctor in [("resnet18", ref_resnet.resnet18), ("resnet18_ReGP_NRF", ref_resnet.resnet18_ReGP_NRF)]:
        m = ctor()
        m.fc = nn.Identity()
        full = dict(oresnet.init_state(variant, seed=3, affine_seed=11))
        for k in list(m.state_dict().keys()):
            if k.endswith("num_batches_tracked"):
                full[k] = torch.zeros((), dtype=torch.long)
        m.load_state_dict(full, strict=True)
        m.train()
        torch.manual_seed(21)
        m(torch.randn(4, 1, 64, 96) * 1.3 + 0.2)
        m.eval()
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[f"{variant}.buf.{k}"] = t2n(v)
        torch.manual_seed(22)
        x = torch.randn(2, 1, 64, 96) * 1.3 + 0.2
        with torch.no_grad():
            y = m(x)
        out[f"{variant}.x"], out[f"{variant}.y"] = t2n(x), t2n(y)
        out[f"{variant}.affine_seed"] = np.array([3, 11])
    save("bn_eval", **out)


def gen_schedule():
    """The per-iteration choices of main.py's loop: `generate_random` (utils/utils.py:30-33) over a seeded sequence of calls (the Python
    and numpy global generators both seeded, as a seeded run has them), and the mask-ratio table main.py:440-448 builds."""
    out = {}
    random.seed(5)
    np.random.seed(5)
    out["rand_mask_ratio"] = np.array([float(ref_utils.generate_random(l=0.05, h=0.3, p=0.5)) for _ in range(64)])
    out["rand_after"] = np.array([random.random(), np.random.uniform()])             # both generators' positions after the calls
    out["mask_table"] = ref_utils.sine_scheduler_increase(final_value=0.3, epochs=10, niter_per_ep=7, warmup_epochs=int(10 / 5), warmup_value=0)
    save("schedule", **out)


def gen_audiontt_se():
    """AudioNTT2022(squeeze_excitation=True) (model.py:141-151,196-213): SE gates after both MaxPools, train mode, Dropout mask recovered
    as in gen_audiontt.  Small widths (d 1280, hidden 256); the gates' Linear weights scaled up so that the sigmoid is not flat."""
    out = {}
    torch.manual_seed(0)
    m = ref_model.AudioNTT2022(n_mels=64, d=1280, mlp_hidden_d=256, squeeze_excitation=True)
    with torch.no_grad():
        for k in (4, 9):
            m.features[k].excitation[0].weight.mul_(4.0)
            m.features[k].excitation[2].weight.mul_(4.0)
    m.train()
    for k, v in m.state_dict().items():
        out["sd." + k] = t2n(v)
    rec = {}
    hook = m.fc[2].register_forward_hook(lambda mod, inp, o: rec.update(inp=inp[0].detach().clone(), out=o.detach().clone()))
    torch.manual_seed(6)
    x = torch.randn(3, 1, 64, 40)
    y = m(x)
    hook.remove()
    keep = ((rec["out"] != 0) | (rec["inp"] == 0)).float()
    w = torch.linspace(-1, 1, y.numel()).reshape(y.shape)
    m.zero_grad()
    (y * w).sum().backward()
    out.update(x=t2n(x), y=t2n(y), keep=t2n(keep))
    for n, prm in m.named_parameters():
        out["grad." + n] = t2n(prm.grad)
    save("audiontt_se", **out)


def gen_hear():
    """HEAR wrapper (hear/sample/vit.py:40-247, hear/utils.py) run as the reference runs it: `ViTModelWrapper` + `get_scene_embeddings` /
    `get_timestamp_embeddings` on two 1.3 s clips.  Two absent third-party names are stood in for: `easydict.EasyDict` (attribute dict)
    and `torchaudio.transforms.MelSpectrogram` -- the latter by oracle/frontend.py's restatement of torchaudio's documented defaults, so
    this fixture pins everything the wrapper does AROUND the mel spectrogram (log + eps, batch statistics incl. compute_timestamp_stats'
    division by the frame count, framing, unit chunking with the extra padded unit, CLS pooling, timestamps) and leaves the mel
    arithmetic itself unpinned, as DESIGN.md section 3 states.  The encoder is the micro ViT of the other fixtures, injected through
    `mae.get_mae_vit` (the wrapper's logic does not depend on the encoder's size)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import frontend as ofe

    class EasyDict(dict):
        __getattr__ = dict.__getitem__

    class MelSpectrogram(nn.Module):
        def __init__(self, sample_rate, n_fft, win_length, hop_length, n_mels, f_min, f_max, power=2):
            super().__init__()
            assert power == 2
            self.kw = dict(n_fft=n_fft, hop=hop_length, n_mels=n_mels, f_min=float(f_min), f_max=float(f_max), sr=sample_rate, win=win_length)

        def forward(self, wave):
            k = self.kw
            mel = ofe.mel_power(wave.detach().cpu().double().numpy(), k["n_fft"], k["hop"], k["n_mels"], k["f_min"], k["f_max"], k["sr"], win_length=k["win"])
            return torch.from_numpy(mel).float()

    ta, tat, ed = types.ModuleType("torchaudio"), types.ModuleType("torchaudio.transforms"), types.ModuleType("easydict")
    tat.MelSpectrogram, ta.transforms, ed.EasyDict = MelSpectrogram, tat, EasyDict
    sys.modules.update({"torchaudio": ta, "torchaudio.transforms": tat, "easydict": ed})
    import importlib
    ref_hear = importlib.import_module("hear.sample.vit")
    torch.manual_seed(0)
    micro = micro_vit()
    orig = ref_mae.get_mae_vit
    ref_mae.get_mae_vit = lambda size, patch_size, c: micro
    try:
        cwd = os.getcwd()
        os.chdir(REF)                                   # the wrapper opens hear/config.yaml relative to the repository root
        model = ref_hear.load_model("", "vit_tiny", "16x16")
        model = model.cpu()
        g = torch.Generator().manual_seed(5)
        t = torch.arange(20800) / 16000.0
        audio = 0.1 * torch.randn(2, 20800, generator=g) + 0.4 * torch.sin(2 * torch.pi * 440.0 * t)[None] * torch.tensor([[1.0], [0.3]])
        orig_cuda = torch.cuda.is_available
        torch.cuda.is_available = lambda: False
        scene = ref_hear.get_scene_embeddings(audio, model)
        emb, ts = ref_hear.get_timestamp_embeddings(audio, model, hop_size=100)
        spec = model._to_normalized_spec(audio)
        torch.cuda.is_available = orig_cuda
    finally:
        ref_mae.get_mae_vit = orig
        os.chdir(cwd)
    out = {"audio": t2n(audio), "scene": t2n(scene), "ts_emb": t2n(emb), "ts": t2n(ts), "norm_spec": t2n(spec),
           "unit_frames": np.array(micro.img_size[1]), "timestamp_embedding_size": np.array(model.timestamp_embedding_size)}
    for k, v in micro.state_dict().items():
        out["sd." + k] = t2n(v)
    save("hear", **out)


# ----------------------------------------------------------------------------- non-default options of the path (round 4)
def gen_options():
    """Options the reference exposes that are off by default: `--projector_n_hidden_layers` != 1 (model.py:16-22), the learned
    positional embedding (`--use_learned_pos_embd`, models/mae.py:196-199) and `norm_pix_loss` (models/mae.py:443-446)."""
    out = {}
    # ---- projector depth 2 and 0
    for tag, nh in (("h2", 2), ("h0", 0)):
        cfg = cfg_ns(projector_hidden_dim=192, projector_out_dim=64, projector_n_hidden_layers=nh)
        torch.manual_seed(10 + nh)
        head = ref_model.BarlowTwinsHead(cfg, in_dim=128)
        with torch.no_grad():
            for m in head.projector:
                if isinstance(m, nn.BatchNorm1d):
                    m.weight.add_(0.2 * torch.randn(m.weight.shape))
                    m.bias.add_(0.2 * torch.randn(m.bias.shape))
        for k, v in head.state_dict().items():
            out[f"{tag}_sd." + k] = t2n(v)
        x = torch.randn(2 * 12, 128, requires_grad=True)
        z = head(x, ncrops=2)
        w = torch.randn_like(z)
        (z * w).sum().backward()
        out.update({f"{tag}_x": t2n(x), f"{tag}_z": t2n(z), f"{tag}_w": t2n(w), f"{tag}_dx": t2n(x.grad)})
        for n, p in head.named_parameters():
            out[f"{tag}_grad." + n] = t2n(p.grad)
        for k, v in head.state_dict().items():
            out[f"{tag}_sd_after." + k] = t2n(v)
    # ---- learned positional embedding: micro ViT, T = 96 (the table's own grid: no interpolation), CLS latent, linear loss
    torch.manual_seed(21)
    vit = ref_mae.MaskedAutoencoderViT(img_size=(64, 96), patch_size=[16, 16], in_chans=1, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4,
                                       norm_layer=partial(nn.LayerNorm, eps=1e-6), use_learned_pos_embd=True)
    perturb_(vit, 22)
    with torch.no_grad():
        vit.pos_embed.add_(0.05 * torch.randn(vit.pos_embed.shape))
    assert vit.pos_embed.requires_grad
    for k, v in vit.state_dict().items():
        out["lpe_sd." + k] = t2n(v)
    x = torch.randn(3, 1, 64, 96)
    lat = vit(x)
    w = torch.randn_like(lat)
    (lat * w).sum().backward()
    out.update(lpe_x=t2n(x), lpe_latent=t2n(lat), lpe_w=t2n(w), lpe_dpos=t2n(vit.pos_embed.grad), lpe_dcls=t2n(vit.cls_token.grad),
               lpe_dqkv0=t2n(vit.blocks[0].attn.qkv.weight.grad))
    # ---- norm_pix_loss: decoder + reconstruction loss with per-patch normalised targets
    torch.manual_seed(23)
    mvit = ref_mae.MaskedAutoencoderViT(img_size=(64, 96), patch_size=[16, 16], in_chans=1, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6), use_decoder=True, decoder_embed_dim=64, decoder_depth=1,
                                        decoder_num_heads=1, norm_pix_loss=True)
    imgs = torch.randn(3, 1, 64, 96) * 1.7 + 0.4
    pred = torch.randn(3, 24, 256, requires_grad=True)
    mask = (torch.rand(3, 24) < 0.6).float()
    mask[0, 0] = 1.0
    loss = mvit.forward_loss(imgs, pred, mask)
    loss.backward()
    out.update(npl_imgs=t2n(imgs), npl_pred=t2n(pred), npl_mask=t2n(mask), npl_loss=t2n(loss), npl_dpred=t2n(pred.grad))
    save("options", **out)


if __name__ == "__main__":
    gen_options() if "options" in sys.argv[1:] else None
    gen_hear() if "hear" in sys.argv[1:] else None
    gen_resnet() if "resnet" in sys.argv[1:] else None
    gen_convstem() if "convstem" in sys.argv[1:] else None
    gen_convstem_lpe() if "convstem_lpe" in sys.argv[1:] else None
    gen_audiontt() if "audiontt" in sys.argv[1:] else None
    gen_bn_eval() if "bn_eval" in sys.argv[1:] else None
    gen_schedule() if "schedule" in sys.argv[1:] else None
    gen_audiontt_se() if "audiontt_se" in sys.argv[1:] else None
    if len(sys.argv) > 1:
        sys.exit(0)
    gen_bt_loss()
    gen_augment()
    gen_vit()
    gen_head()
    gen_step()
    gen_misc()
    gen_optim()
    gen_eval()
    gen_noise_norm()
