"""Pins the CPU oracle (oracle/) against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import random

import numpy as np
import pytest
import torch

from oracle import augment as oaug
from oracle import frontend as ofe
from oracle import heads as oh
from oracle import step as ostep
from oracle import vit as ovit


def T(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype)            # (a copy: the cached fixtures must never be written through a tensor)


# ----------------------------------------------------------------------------- BT loss
@pytest.mark.parametrize("tag", ["anchor", "hsic", "cfg1", "ragged"])
def test_bt_forward_loss(golden, tag):
    g = golden("bt_loss")
    z1, z2 = T(g[f"{tag}_z1"]).requires_grad_(True), T(g[f"{tag}_z2"]).requires_grad_(True)
    hsic = bool(g[f"{tag}_hsic"])
    loss, stats = oh.bt_forward_loss(z1, z2, 1.0, 0.005, hsic)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{tag}_loss"], rtol=2e-6)
    np.testing.assert_allclose(z1.grad.numpy(), g[f"{tag}_dz1"], rtol=1e-4, atol=2e-8)
    np.testing.assert_allclose(z2.grad.numpy(), g[f"{tag}_dz2"], rtol=1e-4, atol=2e-8)
    # analytic backward agrees with the reference's autograd
    l2, dz1, dz2 = oh.bt_forward_loss_backward(z1.detach().double(), z2.detach().double(), 1.0, 0.005, hsic)
    np.testing.assert_allclose(l2.item(), g[f"{tag}_loss"], rtol=2e-6)
    np.testing.assert_allclose(dz1.numpy(), g[f"{tag}_dz1"], rtol=2e-4, atol=5e-8)
    np.testing.assert_allclose(dz2.numpy(), g[f"{tag}_dz2"], rtol=2e-4, atol=5e-8)
    rm, rv, nbt = oh.bt_running_stats([(tuple(s.detach() for s in stats), z1.shape[0])], z1.shape[1])
    np.testing.assert_allclose(rm.numpy(), g[f"{tag}_running_mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rv.numpy(), g[f"{tag}_running_var"], rtol=1e-5)
    assert nbt == int(g[f"{tag}_nbt"])


def test_bt_anchor_value(golden):
    """Known-answer anchor recorded in SURVEY.md §8(c)."""
    assert abs(float(golden("bt_loss")["anchor_loss"]) - 5.219377040863037) < 1e-5


@pytest.mark.parametrize("tag", ["g1L0", "g1L1", "g2L0"])
def test_bt_forward_crops(golden, tag):
    g = golden("bt_loss")
    st, te = T(g[f"fwd_{tag}_student"]).requires_grad_(True), T(g[f"fwd_{tag}_teacher"]).requires_grad_(True)
    loss, stats = oh.bt_forward(st, te, int(g[f"fwd_{tag}_ncrops"]), int(g[f"fwd_{tag}_g"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"fwd_{tag}_loss"], rtol=3e-6)
    np.testing.assert_allclose(st.grad.numpy(), g[f"fwd_{tag}_dstudent"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(te.grad.numpy(), g[f"fwd_{tag}_dteacher"], rtol=1e-4, atol=1e-7)
    rm, rv, _ = oh.bt_running_stats([(tuple(s.detach() for s in s4), n) for s4, n in stats], st.shape[1])
    np.testing.assert_allclose(rm.numpy(), g[f"fwd_{tag}_running_mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rv.numpy(), g[f"fwd_{tag}_running_var"], rtol=1e-5)


def test_off_diagonal(golden):
    g = golden("bt_loss")
    np.testing.assert_array_equal(oh.off_diagonal(T(g["offdiag_in"])).numpy(), g["offdiag_out"])


# ----------------------------------------------------------------------------- augmentations
def test_log_mixup_exp(golden):
    g = golden("augment")
    for k in range(4):
        y = oaug.log_mixup_exp(g["lme_xa"].astype(np.float64), g["lme_xb"].astype(np.float64), float(g[f"lme_{k}_alpha"]))
        np.testing.assert_allclose(y, g[f"lme_{k}_out"], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("tag", ["t96", "t96b", "t1001", "t1001b", "local"])
def test_rrc(golden, tag):
    g = golden("augment")
    cfg = g[f"rrc_{tag}_cfg"]
    out_size, vcs = (int(cfg[0]), int(cfg[1])), (cfg[2], cfg[3])
    x = g[f"rrc_{tag}_x"]
    # sampler: same seeds -> same (i, j, h, w)  (x itself consumed torch RNG only)
    seed = int(cfg[8])
    np_rng, py_rng = np.random.RandomState(seed), random.Random(seed)
    canvas = oaug.canvas_size(x.shape[-2:], vcs)
    p = oaug.draw_rrc_params(np_rng, py_rng, canvas, x.shape[-2:], (cfg[6], cfg[7]), (cfg[4], cfg[5]))
    assert list(p) == list(g[f"rrc_{tag}_params"])
    y = oaug.rrc_apply(x, p, out_size, vcs)
    np.testing.assert_allclose(y, g[f"rrc_{tag}_y"], rtol=0, atol=3e-5)


def test_rrc_anchor(golden):
    """SURVEY.md §8(c) anchor: seeds 123, 64x96 in, canvas 64x144 -> (0, 3, 64, 82)."""
    assert list(golden("augment")["rrc_t96_params"]) == [0, 3, 64, 82]


def test_linear_fader(golden):
    g = golden("augment")
    y = oaug.linear_fader_apply(g["rlf_x"].astype(np.float64), *g["rlf_head_tail"])
    np.testing.assert_allclose(y, g["rlf_y"], atol=1e-6)


def test_normalize_batch(golden):
    g = golden("augment")
    np.testing.assert_allclose(oaug.normalize_batch(g["nb_x"]), g["nb_y"], atol=2e-6)


@pytest.mark.parametrize("tag", ["seq96", "seq208", "seq96_local"])
def test_audio_pair_transform_sequence(golden, tag):
    """Bank evolution + RNG call order over 5 consecutive clips."""
    g = golden("augment")
    clips = g[f"apt_{tag}_clips"]
    L = int(g[f"apt_{tag}_L"])
    tfm = oaug.PairTransformOracle(crop_frames=clips.shape[-1], local_crops_number=L, seed=int(g[f"apt_{tag}_seed"]))
    rrc_seen = []
    for k, c in enumerate(clips):
        crops = tfm(c)
        for v in range(2):
            np.testing.assert_allclose(crops[v], g[f"apt_{tag}_views"][k, v], atol=5e-5)
        for l in range(L):
            np.testing.assert_allclose(crops[2 + l], g[f"apt_{tag}_locals"][k, l], atol=5e-5)
    rrc_seen = [list(r["rrc"]) for r in tfm.records]
    assert rrc_seen == g[f"apt_{tag}_rrc_params"].tolist()
    assert tfm.records[0]["bank_index"] == -1 and tfm.records[1]["bank_index"] == 0
    assert len(tfm.bank) == 10


# ----------------------------------------------------------------------------- frontend (parity unpinned: self-consistency only)
def test_frontend_matches_torch_stft():
    rng = np.random.RandomState(0)
    wave = (0.1 * rng.randn(2, 16000)).astype(np.float32)
    p = ofe.power_spectrogram(wave)
    st = torch.stft(torch.from_numpy(wave).double(), 1024, 160, 1024, torch.hann_window(1024, periodic=True, dtype=torch.float64),
                    center=True, pad_mode="reflect", return_complex=True)
    np.testing.assert_allclose(p, (st.abs() ** 2).numpy(), rtol=1e-9, atol=1e-12)
    assert ofe.logmel(wave).shape == (2, 64, 101)
    assert ofe.n_frames(160000) == 1001 and ofe.n_frames(15200) == 96


def test_mel_filterbank_shape_and_partition():
    fb = ofe.mel_filterbank()
    assert fb.shape == (513, 64) and fb.min() >= 0
    assert (fb > 0).sum(1).max() <= 2          # each bin feeds at most two triangles
    np.testing.assert_allclose(fb[40:480].sum(1), 1.0, atol=1e-9)  # interior bins: the two slopes sum to 1


def test_crop_pad_normalize():
    x = np.arange(2 * 5, dtype=np.float64).reshape(1, 2, 5)
    np.testing.assert_allclose(ofe.crop_pad_normalize(x, 3, 1, 1.0, 2.0), (x[..., 1:4] - 1) / 2)
    y = ofe.crop_pad_normalize(x, 7, 0, 1.0, 2.0)
    assert y.shape[-1] == 7 and np.allclose(y[..., 5:], -0.5)


# ----------------------------------------------------------------------------- ViT micro
def _vit_params(g, prefix="sd."):
    return {k[len(prefix):]: T(v) for k, v in g.items() if k.startswith(prefix)}


def test_pos_tables(golden):
    g = golden("vit_micro")
    np.testing.assert_allclose(ovit.sincos_2d(192, (4, 6)), g["sincos_192_4x6"], atol=1e-6)
    np.testing.assert_allclose(ovit.sincos_2d(768, (4, 6)), g["sincos_768_4x6"], atol=1e-6)
    np.testing.assert_allclose(ovit.sinusoid_table(24, 384), g["sinusoid_24_384"], atol=1e-6)


@pytest.mark.parametrize("tag,T_", [("t96", 96), ("t208", 208), ("t1001", 1001)])
def test_vit_forward_backward(golden, tag, T_):
    g = golden("vit_micro")
    p = _vit_params(g)
    names = [k[len(f"{tag}_grad."):] for k in g if k.startswith(f"{tag}_grad.")]
    for n in names:
        p[n].requires_grad_(True)
    x = T(g[f"{tag}_x"])
    pos = ovit.interpolate_pos_embed(p["pos_embed"].numpy(), (4, 6), 64, T_)
    np.testing.assert_allclose(pos, g[f"{tag}_pos"], atol=2e-6)
    tok, _, _ = ovit.prepare_tokens(x, p, (4, 6))
    np.testing.assert_allclose(tok.detach().numpy(), g[f"{tag}_tokens"], atol=1e-5)
    enc, _, _ = ovit.forward_encoder(x, p, 2, (4, 6))
    np.testing.assert_allclose(enc.detach().numpy(), g[f"{tag}_encoded"], atol=5e-5)
    lat = ovit.forward(x, p, 2, (4, 6))
    np.testing.assert_allclose(lat.detach().numpy(), g[f"{tag}_latent"], atol=5e-5)
    np.testing.assert_allclose(ovit.forward(x, p, 2, (4, 6), mean_pool=True).detach().numpy(), g[f"{tag}_latent_meanpool"], atol=5e-5)
    w = torch.linspace(-1, 1, lat.numel()).reshape(lat.shape)
    (lat * w).sum().backward()
    for n in names:
        np.testing.assert_allclose(p[n].grad.numpy(), g[f"{tag}_grad.{n}"], rtol=2e-3, atol=2e-5, err_msg=n)


def test_vit_masking(golden):
    g = golden("vit_micro")
    p = _vit_params(g)
    x = T(g["mask_x"])
    lat = ovit.forward(x, p, 2, (4, 6), mask=T(g["mask_mask"]))
    np.testing.assert_allclose(lat.numpy(), g["mask_latent"], atol=5e-5)
    _, m, ids = ovit.masking_from_noise(ovit.patch_embed(x, p), mask=T(g["mask_mask"]))
    np.testing.assert_array_equal(ids.numpy(), g["mask_ids_restore"])
    np.testing.assert_array_equal(m.numpy(), g["mask_out_mask"])
    lat2 = ovit.forward(x, p, 2, (4, 6), noise=T(g["rand_noise"]), mask_ratio=0.75)
    np.testing.assert_allclose(lat2.numpy(), g["rand_latent"], atol=5e-5)


@pytest.mark.parametrize("tag,grid", [("t96", (4, 6)), ("t208", (4, 13))])
def test_mae_decoder(golden, tag, grid):
    g = golden("mae_micro")
    p = _vit_params(g, f"{tag}_sd.")
    names = [k[len(f"{tag}_grad."):] for k in g if k.startswith(f"{tag}_grad.")]
    for n in names:
        p[n].requires_grad_(True)
    x = T(g[f"{tag}_x"])
    np.testing.assert_array_equal(ovit.patchify(x, grid).numpy(), g[f"{tag}_patchify"])
    lat, rl = ovit.forward(x, p, 2, grid, mask=T(g[f"{tag}_mask"]), masked_recon=True, dec_heads=1)
    np.testing.assert_allclose(lat.detach().numpy(), g[f"{tag}_latent"], atol=5e-5)
    np.testing.assert_allclose(rl.item(), g[f"{tag}_recon_loss"], rtol=1e-5)
    (rl + lat.sum() * 0.01).backward()
    for n in names:
        np.testing.assert_allclose(p[n].grad.numpy(), g[f"{tag}_grad.{n}"], rtol=2e-3, atol=2e-6, err_msg=n)


# ----------------------------------------------------------------------------- head / predictor
def test_head_and_predictor(golden):
    g = golden("head")
    for pre, fwd, nc in [("head", oh.head_forward, 2), ("pred", oh.predictor_forward, 1)]:
        sd = {k[len(pre + "_sd."):]: T(v) for k, v in g.items() if k.startswith(pre + "_sd.")}
        leaves = {k: v.requires_grad_(True) for k, v in sd.items() if "running" not in k and "num_batches" not in k}
        x = T(g[pre + "_x"]).requires_grad_(True)
        z, stats = fwd(x, sd, ncrops=nc)
        np.testing.assert_allclose(z.detach().numpy(), g[pre + "_z"], atol=2e-5)
        (z * T(g[pre + "_w"])).sum().backward()
        np.testing.assert_allclose(x.grad.numpy(), g[pre + "_dx"], rtol=1e-3, atol=2e-6)
        for k, v in leaves.items():
            np.testing.assert_allclose(v.grad.numpy(), g[f"{pre}_grad.{k}"], rtol=2e-3, atol=5e-6, err_msg=k)
    # running stats after two chunk calls
    sd = {k[len("head_sd."):]: T(v) for k, v in g.items() if k.startswith("head_sd.")}
    x = T(g["head_x"])
    _, stats = oh.head_forward(x, sd, ncrops=2)
    st = {"projector.1.running_mean": sd["projector.1.running_mean"].clone(), "projector.1.running_var": sd["projector.1.running_var"].clone(),
          "projector.1.num_batches_tracked": sd["projector.1.num_batches_tracked"].clone()}
    ostep.apply_bn_buffers(st, "projector.1.", stats)
    np.testing.assert_allclose(st["projector.1.running_mean"].numpy(), g["head_sd_after.projector.1.running_mean"], atol=1e-6)
    np.testing.assert_allclose(st["projector.1.running_var"].numpy(), g["head_sd_after.projector.1.running_var"], rtol=1e-5)
    assert int(st["projector.1.num_batches_tracked"]) == int(g["head_sd_after.projector.1.num_batches_tracked"])


# ----------------------------------------------------------------------------- non-default options (round 4)
def test_options_projector_depth_learned_pos_norm_pix(golden):
    """`--projector_n_hidden_layers` 2 and 0 (model.py:16-22), `--use_learned_pos_embd` (models/mae.py:198-199, :370-392: a trained table,
    resampled for non-square inputs, gradient through the resampling) and `norm_pix_loss` (models/mae.py:443-446) against the reference's
    own outputs and gradients (tests/golden/options.npz)."""
    g = golden("options")
    for tag in ("h2", "h0"):
        sd = {k[len(tag + "_sd."):]: T(v) for k, v in g.items() if k.startswith(tag + "_sd.")}
        leaves = {k: v.requires_grad_(True) for k, v in sd.items() if "running" not in k and "num_batches" not in k}
        x = T(g[tag + "_x"]).requires_grad_(True)
        z, stats = oh.head_forward(x, sd, ncrops=2)
        np.testing.assert_allclose(z.detach().numpy(), g[tag + "_z"], atol=2e-5)
        (z * T(g[tag + "_w"])).sum().backward()
        np.testing.assert_allclose(x.grad.numpy(), g[tag + "_dx"], rtol=1e-3, atol=2e-6)
        for k, v in leaves.items():
            np.testing.assert_allclose(v.grad.numpy(), g[f"{tag}_grad.{k}"], rtol=2e-3, atol=5e-6, err_msg=k)
        assert len(stats) == {"h2": 4, "h0": 0}[tag]                       # two BatchNorms x two crop chunks / none
    p = {k[len("lpe_sd."):]: T(v) for k, v in g.items() if k.startswith("lpe_sd.")}
    for k in ("pos_embed", "cls_token", "blocks.0.attn.qkv.weight"):
        p[k].requires_grad_(True)
    lat = ovit.forward(T(g["lpe_x"]), p, 2, (4, 6), learned_pos=True)
    np.testing.assert_allclose(lat.detach().numpy(), g["lpe_latent"], atol=3e-5)
    (lat * T(g["lpe_w"])).sum().backward()
    np.testing.assert_allclose(p["pos_embed"].grad.numpy(), g["lpe_dpos"], rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(p["cls_token"].grad.numpy(), g["lpe_dcls"], rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(p["blocks.0.attn.qkv.weight"].grad.numpy(), g["lpe_dqkv0"], rtol=2e-3, atol=5e-6)
    # the resampling of the trained table as a matrix (what the HIP path multiplies by) == torch's bicubic interpolate
    from ssl_audio_amd.pos_embed import interpolation_matrix
    A = torch.from_numpy(interpolation_matrix((4, 6), 4, 6)).float()
    ref = ovit.learned_pos_table(T(g["lpe_sd.pos_embed"]), (4, 6), 64, 96)[0, 1:]
    np.testing.assert_allclose((A @ T(g["lpe_sd.pos_embed"])[0, 1:]).numpy(), ref.numpy(), atol=2e-6)
    pred = T(g["npl_pred"]).requires_grad_(True)
    loss = ovit.recon_loss(T(g["npl_imgs"]), pred, T(g["npl_mask"]), (4, 6), norm_pix=True)
    np.testing.assert_allclose(float(loss), float(g["npl_loss"]), rtol=1e-6)
    loss.backward()
    np.testing.assert_allclose(pred.grad.numpy(), g["npl_dpred"], rtol=1e-4, atol=1e-8)


# ----------------------------------------------------------------------------- full step
@pytest.mark.parametrize("tag,stop_grad,use_pred", [("byol", True, True), ("plain", False, False)])
def test_full_step(golden, tag, stop_grad, use_pred):
    g = golden(f"step_{tag}")

    def load(prefix):
        out = {}
        for k, v in g.items():
            if k.startswith(prefix):
                t = torch.from_numpy(np.asarray(v))
                out[k[len(prefix):]] = t.clone()
        return out

    online, pred = load("online_sd."), load("pred_sd.")
    target = {k: v.clone() for k, v in online.items()}
    views = [T(g["view0"]), T(g["view1"])]
    opt = ostep.AdamW(float(g["lr"]), float(g["wd"]))
    losses = []
    for it in range(2):
        l, grads = ostep.bt_byol_step(online, target, pred, views, 2, (4, 6), opt, stop_grad, use_pred)
        losses.append(l)
        if it == 0:
            for k in [k for k in g if k.startswith("grad0.")]:
                np.testing.assert_allclose(grads[k[len("grad0."):]].numpy(), g[k], rtol=5e-3, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-4)
    for k, v in g.items():
        if k.startswith("online_sd_after."):
            np.testing.assert_allclose(online[k[len("online_sd_after."):]].numpy(), v, rtol=1e-3, atol=4e-5, err_msg=k)  # Adam normalises near-zero grads to +-lr: 2 steps * 2 * lr = 2.5e-5
        if k.startswith("target_sd_after."):
            np.testing.assert_allclose(target[k[len("target_sd_after."):]].numpy(), v, rtol=1e-3, atol=4e-5, err_msg=k)  # Adam normalises near-zero grads to +-lr: 2 steps * 2 * lr = 2.5e-5


def test_misc(golden):
    g = golden("misc")
    assert ostep.multicrop_groups([8, 8, 5, 5, 5]) == [(0, 2), (2, 5)]
    np.testing.assert_allclose(0.99 * g["ema_old_w"] + 0.01 * g["ema_new_w"], g["ema_out_w"], atol=1e-7)
    reg, noreg = ostep.split_param_groups(
        [(str(n), (torch.zeros(2, 2) if str(n) in set(g["pg_regularized"].tolist()) else torch.zeros(2)).requires_grad_(True))
         for n in list(g["pg_regularized"]) + list(g["pg_not_regularized"])])
    assert reg == g["pg_regularized"].tolist() and noreg == g["pg_not_regularized"].tolist()


# ------------------------------------------------------------------------------------------------ optimiser family
def test_lars_oracle_golden(golden):
    """oracle.step.lars_step vs three steps of the reference's LARS with the filters main_bt_byol.py:344-345 sets."""
    import torch
    from oracle import step as ostep
    g = golden("optim")
    lr_w, lr_b, wd, mom, eta = [float(x) for x in g["lars_cfg"]]
    st = {k: (torch.from_numpy(g[f"lars_{k}0"].copy()), torch.zeros_like(torch.from_numpy(g[f"lars_{k}0"].copy()))) for k in "wbz"}
    for it in range(3):
        for k, lr in (("w", lr_w), ("z", lr_w), ("b", lr_b)):
            p, mu = st[k]
            st[k] = ostep.lars_step(p, torch.from_numpy(g[f"lars_g{k}{it}"]), mu, lr, wd, mom, eta, weight_decay_filter=True, lars_adaptation_filter=True)
            np.testing.assert_allclose(st[k][0].numpy(), g[f"lars_{k}{it + 1}"], rtol=2e-6, atol=1e-7, err_msg=f"{k} step {it}")


def test_schedules_golden(golden):
    """The product's host-side schedule helpers (ssl_audio_amd.utils) against the reference's tables / learning rates."""
    import types
    from ssl_audio_amd import utils
    g = golden("optim")
    np.testing.assert_allclose(utils.cosine_scheduler(0.5, 0.01, 5, 7, warmup_epochs=2, start_warmup_value=0.1), g["cos_sched"], rtol=1e-12)
    np.testing.assert_allclose(utils.cosine_scheduler(1.0, 0.0, 3, 4), g["cos_sched_nowarm"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(utils.sine_scheduler_increase(0.75, 4, 6, warmup_epochs=1, warmup_value=0.2), g["sine_sched"], rtol=1e-12, atol=1e-15)

    class Opt:
        def __init__(self, n):
            self.param_groups = [{"lr": -1.0} for _ in range(n)]

    loader = list(range(13))
    for name, optname, n in [("adamw", "AdamW", 3), ("lars", "LARS", 2)]:
        args = types.SimpleNamespace(epochs=300, batch_size=256, lr=1e-4, lr_weights=0.2, lr_biases=0.0048, optimizer=optname)
        o = Opt(n)
        for row in g[f"adjust_lr_{name}"]:
            utils.adjust_learning_rate(args, o, loader, int(row[0]))
            np.testing.assert_allclose([gr["lr"] for gr in o.param_groups], row[1:], rtol=1e-12, atol=1e-18)


@pytest.mark.parametrize("tag", ["t222", "t192", "t50"])
def test_encode_vit_oracle_golden(golden, tag):
    """oracle.vit.encode_vit vs the reference's utils.encode_vit (eval path) on the micro ViT."""
    p = _vit_params(golden("vit_micro"))
    g = golden("eval")
    x = T(g[f"{tag}_x"])
    np.testing.assert_allclose(ovit.encode_vit(x, p, 2, (4, 6), 96).numpy(), g[f"{tag}_cls"], atol=5e-5)
    np.testing.assert_allclose(ovit.encode_vit(x, p, 2, (4, 6), 96, use_cls=False).numpy(), g[f"{tag}_patch"], atol=5e-5)
    np.testing.assert_allclose(ovit.encode_vit(x, p, 2, (4, 6), 96, split_frames=False).numpy(), g[f"{tag}_whole"], atol=5e-5)


def test_gaussian_noise_and_running_norm_oracle_golden(golden):
    """MixGaussianNoise (recorded lambda and normal draws) and RunningNorm (3 updating samples, then frozen) vs the reference."""
    g = golden("noise_norm")
    np.testing.assert_allclose(oaug.mix_gaussian_noise(g["gn_x"], float(g["gn_lambda"]), g["gn_normal"]), g["gn_y"], rtol=2e-5, atol=2e-6)
    rn = oaug.RunningNormOracle(epoch_samples=1, max_update_epochs=3)
    for i in range(5):
        np.testing.assert_allclose(rn(g[f"rn_x{i}"]), g[f"rn_y{i}"], rtol=2e-5, atol=2e-6, err_msg=f"sample {i}")
    np.testing.assert_allclose(rn.mu, g["rn_mean"], rtol=1e-5)


@pytest.mark.parametrize("tag", ["p16x16_t96", "p16x8_t96", "p16x16_t208"])
def test_convstem_oracle_golden(golden, tag):
    """oracle.vit.conv_stem (models/mae.py:46-99) inside the encoder: tokens, latent and every captured gradient of the reference's
    micro ViTC, fp32: 1e-5 (bit-exact at the constructor width, interpolation rounding at T = 208)."""
    from oracle import vit as ovit
    g = golden("convstem")
    sd = {k[len(tag) + 4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(tag + "_sd.")}
    patch = tuple(int(v) for v in g[tag + "_patch"])
    x = torch.from_numpy(g[tag + "_x"])
    grid = (4, 96 // patch[1])
    names = [k for k in sd if not ("running" in k or "num_batches" in k or k == "pos_embed")]
    leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    tok, _, _ = ovit.prepare_tokens(x, leaf, grid, patch=patch)
    st = []
    lat = ovit.forward(x, leaf, 2, grid, patch=patch, bn_stats=st)
    assert float((tok.detach() - torch.from_numpy(g[tag + "_tokens"])).abs().max()) < 1e-5
    assert float((lat.detach() - torch.from_numpy(g[tag + "_latent"])).abs().max()) < 1e-5
    w = torch.linspace(-1, 1, lat.numel()).reshape(lat.shape)
    gs = dict(zip(names, torch.autograd.grad((lat * w).sum(), [leaf[k] for k in names], allow_unused=True)))
    for k, v in g.items():
        if k.startswith(tag + "_grad."):
            ref = torch.from_numpy(v)
            assert float((gs[k[len(tag) + 6:]] - ref).norm() / (ref.norm() + 1e-30)) < 1e-5, k
    # BatchNorm running statistics replayed from the recorded batch statistics (momentum 0.1, unbiased variance)
    for l, (mu, var, n) in enumerate(st):
        rm = 0.9 * sd[f"patch_embed.proj.{3 * l + 1}.running_mean"] + 0.1 * mu
        rv = 0.9 * sd[f"patch_embed.proj.{3 * l + 1}.running_var"] + 0.1 * var * n / (n - 1)
        np.testing.assert_allclose(rm.numpy(), g[f"{tag}_after.patch_embed.proj.{3 * l + 1}.running_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(rv.numpy(), g[f"{tag}_after.patch_embed.proj.{3 * l + 1}.running_var"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag,T_", [("t96", 96), ("t208", 208)])
def test_convstem_learned_pos_oracle_golden(golden, tag, T_):
    """ConvStem ViTC with `--use_learned_pos_embd` (models/mae.py:186-199, :367-392): the oracle's latent and the gradients of the trained
    positional table (through the bicubic resampling: 64 x 96 is not square, so even the table's own grid is resampled), the CLS token and
    the stem's last convolution against the reference's (tests/golden/convstem_lpe.npz), fp32."""
    from oracle import vit as ovit
    g = golden("convstem_lpe")
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("sd.")}
    last = str(g[tag + "_last_name"])
    names = ["pos_embed", "cls_token", last]
    leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    x = torch.from_numpy(np.array(g[tag + "_x"]))
    lat = ovit.forward(x, leaf, 2, (4, 6), patch=(16, 16), learned_pos=True)
    assert float((lat.detach() - torch.from_numpy(np.array(g[tag + "_latent"]))).abs().max()) < 3e-5
    w = torch.linspace(-1, 1, lat.numel()).reshape(lat.shape)
    gs = dict(zip(names, torch.autograd.grad((lat * w).sum(), [leaf[k] for k in names])))
    for k, key in (("pos_embed", "_dpos"), ("cls_token", "_dcls"), (last, "_dlast")):
        ref = torch.from_numpy(np.array(g[tag + key]))
        assert float((gs[k] - ref).norm() / (ref.norm() + 1e-30)) < 1e-4, k


def test_hear_frame_audio_matches_the_reference_loop():
    """ssl_audio_amd.hear.utils.frame_audio (vectorised) == the literal restatement of hear/utils.py:56-106 (oracle/hear.py)."""
    from oracle import hear as ohear
    from ssl_audio_amd.hear import utils as hutils
    g = torch.Generator().manual_seed(0)
    for n_samples, frame, hop in [(16000, 15200, 50.0), (27211, 15200, 50.0), (5000, 800, 12.5)]:
        a = torch.randn(2, n_samples, generator=g)
        fr, ts = hutils.frame_audio(a, frame, hop, 16000)
        rf, rt = ohear.frame_audio(a.numpy(), frame, hop, 16000)
        assert fr.shape == rf.shape and np.array_equal(fr.numpy(), rf) and np.allclose(ts.numpy(), rt)


def test_audiontt_oracle_golden(golden):
    """oracle.audiontt (model.py:130-191) against the reference's AudioNTT2022 run in train mode with its own Dropout mask: output 1e-5,
    gradients 1e-5 (the conv biases sit ahead of a BatchNorm: true gradient 0, rounding noise only -- skipped)."""
    from oracle import audiontt as oa
    g = golden("audiontt")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    names = [k for k in sd if not ("running" in k or "num_batches" in k)]
    leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    st = []
    y = oa.forward(torch.from_numpy(g["x"]), leaf, torch.from_numpy(g["keep"]), bn_stats=st)
    assert float((y.detach() - torch.from_numpy(g["y"])).abs().max()) < 1e-5
    w = torch.linspace(-1, 1, y.numel()).reshape(y.shape)
    gs = dict(zip(names, torch.autograd.grad((y * w).sum(), [leaf[k] for k in names])))
    for k, v in g.items():
        if k.startswith("grad.") and float(np.linalg.norm(v)) > 1e-2:
            ref = torch.from_numpy(v)
            assert float((gs[k[5:]] - ref).norm() / ref.norm()) < 1e-5, k
    for l, (mu, var, n) in enumerate(st):
        rv = 0.9 * sd[f"features.{4 * l + 1}.running_var"] + 0.1 * var * n / (n - 1)
        np.testing.assert_allclose(rv.numpy(), g[f"after.features.{4 * l + 1}.running_var"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("variant", ["resnet18", "resnet18_ReGP_NRF"])
def test_resnet_oracle_matches_reference(golden, variant):
    """oracle/resnet.py == models/resnet.py (`fc` = Identity, train mode) on the fixture the reference produced: embedding, gradients of
    the small parameters, gradient norms of all 62, BatchNorm buffers after the step (tests/golden/resnet.npz)."""
    from oracle import resnet as oresnet
    g = golden("resnet")
    seed, aseed = [int(v) for v in g[f"{variant}.affine_seed"]]
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in oresnet.init_state(variant, seed, aseed).items()}
    x = torch.from_numpy(g[f"{variant}.x"])
    stats = []
    y = oresnet.forward(x, p, variant, bn_stats=stats)
    ref = torch.from_numpy(g[f"{variant}.y"])
    assert y.shape == ref.shape and float((y.detach() - ref).norm() / ref.norm()) < 2e-5
    w = torch.linspace(-1, 1, y.numel()).reshape(y.shape)
    (y * w).sum().backward()
    names = [str(n) for n in g[f"{variant}.grad_names"]]
    assert names == [k for k in p if p[k].requires_grad]                       # parameter order == named_parameters() of the reference
    for n, nrm in zip(names, g[f"{variant}.grad_norms"]):
        assert abs(float(p[n].grad.double().norm()) - nrm) <= 2e-4 * max(nrm, 1e-6) + 1e-7, n
        key = f"{variant}.grad.{n}"
        if key in g:
            assert float((p[n].grad - T(g[key])).norm() / T(g[key]).norm()) < 5e-4, n
    for name, mean, var, cnt in stats:                                          # running buffers: momentum 0.1, unbiased variance
        key = f"{variant}.after.{name}.running_mean"
        if key in g:
            assert np.allclose(0.1 * mean.numpy(), g[key], rtol=1e-4, atol=1e-6), name
            assert np.allclose(0.9 + 0.1 * var.numpy() * cnt / (cnt - 1), g[f"{variant}.after.{name}.running_var"], rtol=1e-4, atol=1e-6), name


def test_audiontt_se_oracle_golden(golden):
    """oracle.audiontt with SE gates (model.py:141-151,196-213) against the reference's AudioNTT2022(squeeze_excitation=True): output and
    every parameter gradient, fp32."""
    from oracle import audiontt as oa
    g = golden("audiontt_se")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    names = [k for k in sd if not ("running" in k or "num_batches" in k)]
    leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    y = oa.forward(torch.from_numpy(g["x"]), leaf, torch.from_numpy(g["keep"]))
    assert float((y.detach() - torch.from_numpy(g["y"])).abs().max()) < 1e-5
    w = torch.linspace(-1, 1, y.numel()).reshape(y.shape)
    gs = dict(zip(names, torch.autograd.grad((y * w).sum(), [leaf[k] for k in names])))
    assert "features.4.excitation.0.weight" in gs and "features.9.excitation.2.weight" in gs
    for k in names:
        ref = torch.from_numpy(g["grad." + k])
        if float(ref.norm()) > 1e-3:
            assert float((gs[k] - ref).norm() / ref.norm()) < 2e-4, k


def test_schedule_choices_golden(golden):
    """main.py:51-57,71-81: utils.generate_random consumes the two global generators exactly like the reference (values AND generator
    positions afterwards), and the trainer's mask_ratio_for / apply_schedules make the loop's choices (table entry, random draw,
    constant; learning rates written into param_groups by utils.adjust_learning_rate with the trainer in the optimiser's place)."""
    import random
    import types
    from ssl_audio_amd import utils
    from ssl_audio_amd.train import BarlowTwinsTrainer
    g = golden("schedule")
    random.seed(5)
    np.random.seed(5)
    got = np.array([float(utils.generate_random(l=0.05, h=0.3, p=0.5)) for _ in range(64)])
    np.testing.assert_array_equal(got, g["rand_mask_ratio"])
    np.testing.assert_array_equal(np.array([random.random(), np.random.uniform()]), g["rand_after"])
    assert (got == 0).any() and (got > 0.05).any()
    table = utils.sine_scheduler_increase(final_value=0.3, epochs=10, niter_per_ep=7, warmup_epochs=2, warmup_value=0)
    np.testing.assert_allclose(table, g["mask_table"], rtol=1e-12, atol=1e-15)
    # the trainer's choices, without building a trainer (no GPU here): the methods only read cfg / param_groups
    t = types.SimpleNamespace(cfg=types.SimpleNamespace(mask=True, random_mask_ratio=False, mask_ratio=0.75, mask_beta=0.3, lr_schedule=True,
                                                        epochs=300, batch_size=256, lr=1e-4, optimizer="AdamW"),
                              mask_ratio_schedule=None, param_groups=[{"lr": -1.0} for _ in range(4)])
    assert BarlowTwinsTrainer.mask_ratio_for(t, 3) == 0.75
    t.mask_ratio_schedule = table
    assert BarlowTwinsTrainer.mask_ratio_for(t, 20) == float(g["mask_table"][20])
    t.mask_ratio_schedule, t.cfg.random_mask_ratio = None, True
    random.seed(5)
    np.random.seed(5)
    assert [float(BarlowTwinsTrainer.mask_ratio_for(t, i)) for i in range(8)] == list(g["rand_mask_ratio"][:8])
    t.cfg.mask = False
    assert BarlowTwinsTrainer.mask_ratio_for(t, 0) == 0
    lr_rows = golden("optim")["adjust_lr_adamw"]                       # reference's adjust_learning_rate, 13 iterations per epoch
    for row in lr_rows:
        BarlowTwinsTrainer.apply_schedules(t, int(row[0]), 13)
        assert all(abs(grp["lr"] - row[1]) <= 1e-12 * max(abs(row[1]), 1e-30) for grp in t.param_groups), row


def test_bn_eval_oracle_matches_reference(golden):
    """Eval mode of the BatchNorm2d encoders (running statistics): oracle == the reference's `model.eval()` forward on
    tests/golden/bn_eval.npz (ConvStem ViTC micro encoder, both ResNet-18 variants)."""
    from oracle import resnet as oresnet, vit as ovit
    g = golden("bn_eval")
    sd = {k[len("vitc_sd."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("vitc_sd.")}
    x = torch.from_numpy(g["vitc_x"])
    tok, _, _ = ovit.prepare_tokens(x, sd, (4, 12), patch=(16, 8), bn_stats="eval")
    lat = ovit.forward(x, sd, 2, (4, 12), patch=(16, 8), bn_stats="eval")
    assert float((tok - torch.from_numpy(g["vitc_tokens"])).abs().max()) < 1e-5
    assert float((lat - torch.from_numpy(g["vitc_latent"])).abs().max()) < 1e-5
    for variant in ("resnet18", "resnet18_ReGP_NRF"):
        seed, aseed = [int(v) for v in g[f"{variant}.affine_seed"]]
        p = oresnet.init_state(variant, seed, aseed)
        p.update({k[len(variant) + 5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(f"{variant}.buf.") and "running" in k})
        y = oresnet.forward(torch.from_numpy(g[f"{variant}.x"]), p, variant, training=False)
        ref = torch.from_numpy(g[f"{variant}.y"])
        assert float((y - ref).norm() / ref.norm()) < 2e-5, variant


def test_hear_oracle_matches_the_reference_wrapper(golden):
    """oracle/hear.py == hear/sample/vit.py + hear/utils.py as the reference ran them (tests/golden/hear.npz: the reference's wrapper with
    `torchaudio.transforms.MelSpectrogram` stood in for by oracle/frontend.py, so the mel arithmetic itself stays unpinned): the normalised
    log-mel, scene embeddings (mean over units incl. the extra padded unit), timestamp embeddings with compute_timestamp_stats' division
    by the frame count, and the timestamps."""
    from oracle import hear as ohear
    g = golden("hear")
    sd = {k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")}
    audio = g["audio"]
    unit = int(g["unit_frames"])
    x = ohear.to_feature(audio)
    x = (x - x.mean()) / x.std()
    np.testing.assert_allclose(x.numpy(), g["norm_spec"], rtol=0, atol=2e-5)
    scene = ohear.scene_embeddings(audio, sd, 2, (4, 6), unit)
    assert scene.shape == g["scene"].shape
    np.testing.assert_allclose(scene.numpy(), g["scene"], rtol=0, atol=2e-4)
    emb, ts = ohear.timestamp_embeddings(audio, sd, 2, (4, 6), unit, hop_size=100)
    assert emb.shape == g["ts_emb"].shape and int(g["timestamp_embedding_size"]) == 128 * 4
    np.testing.assert_allclose(ts, g["ts"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(emb.numpy(), g["ts_emb"], rtol=0, atol=5e-4)
