"""The BASELINE.json encoders under a check (round-1 verdict: configs 3/4/5 never ran in a test).

  cfg 3  ViT-B, 10 s clips (T = 1001, N = 249), trainer 'bt'  -- B = 24 -> S*N = 11 952 rows, so qkv / proj / fc1 / fc2 forward,
         dgrad and wgrad all dispatch to the 256^2 persistent / ring / split-K kernels (>= 128 tiles of 256 x 256)
  cfg 4  trainer 'byol' (EMA target + predictor, two AdamW states), two steps
  cfg 5  trainer 'mae', ViT-L, T = 992, 75 % masking + reconstruction
  full-size property runs of every config (128 / 256 clips per GPU): finite, loss falls, every trainable gradient non-zero

Expected values come from the CPU oracle on the same weights and views.  Gradients are compared against the oracle in
its bf16-mirror mode (oracle/rounding.py: the same restatement, rounding where the HIP path stores bf16), bound 2e-2.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gradcheck import check_step_gradients  # noqa: E402  (tests/gradcheck.py)
from ssl_audio_amd import engine, hyperparameters as hp, ops  # noqa: E402
from ssl_audio_amd.train import BarlowTwinsTrainer  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    ops.lib()
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def flat_grads(tr, names=None, flat=None, module=None):
    """Gradients the step left in the flat buffer (before AdamW consumed them they are not modified by it)."""
    module = module or tr.online
    out = {}
    for k, p in module.named_parameters():
        ent = engine.GRAD_SINK.get(id(p))
        if ent is not None and ent[0]() is p and (names is None or k in names):
            out[k] = ent[1].detach().float().cpu().clone()
    return out


def grad_report(got, ref, floor=1e-6):
    errs = {}
    for k, g in ref.items():
        if k in got and float(g.norm()) > floor:
            errs[k] = rel(got[k], g)
    return errs


def cosine_rows(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return torch.nn.functional.cosine_similarity(a, b, dim=1)


def correlated_views(B, T, seed):
    """Two views that share content (like two augmentations of one clip), so the BT loss is in its working regime."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(B, 1, 64, T, generator=g)
    return [base + 0.3 * torch.randn(B, 1, 64, T, generator=g), base + 0.3 * torch.randn(B, 1, 64, T, generator=g)]


def cpu_state(module):
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


# ------------------------------------------------------------------------------------------------ cfg 3: ViT-B, 10 s
def test_cfg3_vit_base_10s_step_vs_oracle(dev):
    """BASELINE config 3's per-GPU step at B = 24 (main.py:86-119 / train_one_epoch's single-network form): embeddings by
    cosine >= 0.999 and loss rel <= 3e-2 against the fp32 oracle; first-step gradients rel <= 2e-2 against the bf16-mirror
    oracle.  Every encoder GEMM of this shape takes the 256^2 kernels (asserted through ops.gemm_kernel_family)."""
    from oracle import rounding as R, step as ostep
    B, T = 24, 1001
    cfg = hp.make_args(model_type="vit_base", batch_size=B, crop_frames=T, dataset="audioset")
    tr = BarlowTwinsTrainer(cfg, dev, mode="bt", batch_per_rank=B, clip_samples=160000, seed=0, from_waveform=False)
    M = 2 * B * 249
    for (n, k) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        assert ops.gemm_kernel_family(M, n, k, True, True, 1, False) == "gemm256_persist_kernel"
        # data gradients read the transposed weight copy in the forward layout (engine._dgrad_w); the k-strided read stays available
        assert ops.gemm_kernel_family(M, k, n, True, True, 1, False, epi1=True) in ("gemm256_persist_kernel", "gemm256_phase_kernel")
        assert ops.gemm_kernel_family(M, k, n, True, False, 1, False) == "gemm256_ring_kernel"
    sd0 = cpu_state(tr.online)
    views = correlated_views(B, T, seed=11)
    dviews = [v.to(dev) for v in views]
    with torch.no_grad():
        z_hip = tr.online(dviews, ncrops=2).float().cpu()
    bn = tr.online.head.projector[1]                                  # undo the probe's running-statistics update
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        getattr(bn, k).copy_(sd0["head.projector.1." + k])
    loss = float(tr.step_views(dviews))
    torch.cuda.synchronize()
    got = flat_grads(tr)
    # ---- fp32 oracle: embeddings + loss
    sd = {k: v.clone() for k, v in sd0.items()}
    z_ref, _ = ostep.network_forward(sd, views, 2, 12, (4, 6))
    cos = cosine_rows(z_hip, z_ref)
    assert float(cos.min()) >= 0.999, float(cos.min())
    ref_loss, fgrads = ostep.bt_step(sd, views, 12, (4, 6), ostep.AdamW(cfg.lr, cfg.wd))
    assert np.isfinite(loss) and abs(loss - ref_loss) / abs(ref_loss) <= 3e-2, (loss, ref_loss)
    # ---- bf16-mirror oracle: gradients
    sd = {k: v.clone() for k, v in sd0.items()}
    with R.mirror_hip_bf16():
        mloss, mgrads = ostep.bt_step(sd, views, 12, (4, 6), ostep.AdamW(cfg.lr, cfg.wd))
    assert abs(loss - mloss) / abs(mloss) <= 1e-2, (loss, mloss)
    print("cfg3 ViT-B 10s: loss", loss, "oracle", ref_loss, "mirror", mloss, "min cos", float(cos.min()))
    check_step_gradients("cfg3 step", got, mgrads, fgrads, 150)


def test_cfg3_encoder_gradients_linear_loss(dev):
    """The discriminating gradient check at BASELINE config 3's shapes: ViT-B, T = 1001, 48 sequences (11 952 rows: every encoder
    GEMM forward / dgrad / wgrad on the 256^2 kernels), loss = sum(latent * w) with a fixed w -- no BatchNorm, no batch-mean
    cancellation, so the gradients are well conditioned.  Every encoder parameter's gradient against the oracle in bf16-mirror
    mode: rel <= 2e-2 (a sign or operand error in any kernel is O(1)); latent rel <= 1e-2 against the same oracle."""
    from oracle import rounding as R, vit as ovit
    from ssl_audio_amd import mae
    S, T = 48, 1001
    torch.manual_seed(0)
    m = mae.get_mae_vit("base").to(dev)
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for n, p in m.named_parameters():                               # biases / LN affine / q,v biases away from their trivial init
            if p.requires_grad and p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g).to(dev))
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = torch.randn(S, 1, 64, T, generator=g)
    w = torch.randn(S, 768, generator=g)
    lat = m(x.to(dev))
    (lat * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    names = [k for k, p in m.named_parameters() if p.requires_grad]
    got = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters() if p.requires_grad and p.grad is not None}
    leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    with R.mirror_hip_bf16():
        ref = ovit.forward(x, leaf, 12, (4, 6))
        mg = torch.autograd.grad((ref * w).sum(), [leaf[k] for k in names], allow_unused=True)
    assert rel(lat, ref) <= 1e-2, rel(lat, ref)
    mgrads = {k: g_ for k, g_ in zip(names, mg) if g_ is not None}
    errs = grad_report(got, mgrads)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print("cfg3 encoder, linear loss: latent rel", rel(lat, ref), "grads", len(errs), "median", float(np.median(list(errs.values()))), "worst", worst)
    # q_bias = column sums of dQ, and sum_j dS_ij = 0 for every query row: with near-identical keys (random init, late blocks) the
    # sum cancels almost completely, so the bf16 storage of dqkv shows in it (7e-2 at block 11); every other gradient is <= 2e-2
    rest = {k: v for k, v in errs.items() if not k.endswith("attn.q_bias")}
    qb = {k: v for k, v in errs.items() if k.endswith("attn.q_bias")}
    assert len(errs) == len(names) == 12 * 13 + 3 and max(rest.values()) <= 2e-2 and max(qb.values()) <= 0.15, worst


# ------------------------------------------------------------------------------------------------ cfg 4: trainer mode 'byol'
def test_cfg4_byol_trainer_two_steps_vs_oracle(dev):
    """main_bt_byol.py:79-135 with --stop_gradient --predictor as train.BarlowTwinsTrainer(mode='byol') runs it (flat EMA target,
    second FlatState + GradSync for the predictor): two steps on ViT-T against oracle.step.bt_byol_step -- losses, online and
    predictor gradients (mirror oracle, 2e-2), the EMA'd target and that both AdamW states advanced."""
    from oracle import rounding as R, step as ostep
    B, T = 16, 96
    cfg = hp.make_args(model_type="vit_tiny", batch_size=B, crop_frames=T, projector_hidden_dim=512, projector_out_dim=128,
                       stop_gradient=True, predictor=True)
    tr = BarlowTwinsTrainer(cfg, dev, mode="byol", batch_per_rank=B, clip_samples=15200, seed=0, from_waveform=False)
    on0, tg0, pr0 = cpu_state(tr.online), cpu_state(tr.target), cpu_state(tr.predictor)
    for k in on0:
        assert torch.equal(on0[k], tg0[k]), k                       # target starts as a copy (main_bt_byol.py:428)
    on_f, tg_f, pr_f = ({k: v.clone() for k, v in s.items()} for s in (on0, tg0, pr0))
    on_m, tg_m, pr_m = ({k: v.clone() for k, v in s.items()} for s in (on0, tg0, pr0))
    opt_f, opt_m = ostep.AdamW(cfg.lr, cfg.wd), ostep.AdamW(cfg.lr, cfg.wd)
    for it in range(2):
        views = correlated_views(B, T, seed=21 + it)
        loss = float(tr.step_views([v.to(dev) for v in views]))
        torch.cuda.synchronize()
        got_on, got_pr = flat_grads(tr), flat_grads(tr, module=tr.predictor)
        ref_loss, fgrads = ostep.bt_byol_step(on_f, tg_f, pr_f, views, 3, (4, 6), opt_f, True, True)
        assert abs(loss - ref_loss) / abs(ref_loss) <= 3e-2, (it, loss, ref_loss)
        if it == 0:
            with R.mirror_hip_bf16():
                mloss, mgrads = ostep.bt_byol_step(on_m, tg_m, pr_m, views, 3, (4, 6), opt_m, True, True)
            print("byol step 0: loss", loss, "oracle", ref_loss, "mirror", mloss)
            check_step_gradients("byol step 0", got_on, mgrads, fgrads, 150)
            assert all(float(g.abs().max()) > 0 for g in got_pr.values()) and len(got_pr) == 4
        # EMA'd target (parameters only, before the optimiser step; buffers are the target's own forward statistics)
        tg = cpu_state(tr.target)
        for k in ["backbone.encoder.encoder.blocks.0.attn.qkv.weight", "backbone.encoder.encoder.blocks.11.mlp.fc2.bias",
                  "head.projector.0.weight", "head.projector.1.weight", "backbone.encoder.encoder.cls_token"]:
            # step 1 mixes in the online weights after their first AdamW step (|dp| <= lr each): agreement to 0.01 * 2 * lr
            np.testing.assert_allclose(tg[k].numpy(), tg_f[k].numpy(), rtol=0, atol=2.5e-2 * cfg.lr + 1e-7, err_msg=k)
        np.testing.assert_allclose(tg["head.projector.1.running_var"].numpy(), tg_f["head.projector.1.running_var"].numpy(), rtol=3e-2, atol=1e-3)
    assert tr.flat.step_count == 2 and tr.flat_pred.step_count == 2
    pr = cpu_state(tr.predictor)
    moved = float((pr["predictor.0.weight"] - pr0["predictor.0.weight"]).abs().max())
    assert 0 < moved <= 2.02 * cfg.lr + 1e-9, moved
    # the bf16 weight copy the target's GEMMs read follows the EMA (one cast launch per step)
    p = dict(tr.target.named_parameters())["head.projector.3.weight"]
    assert torch.equal(engine.BF16_WEIGHTS.get(p).float().cpu(), p.detach().cpu().to(torch.bfloat16).float())


# ------------------------------------------------------------------------------------------------ cfg 5: ViT-L MAE, T = 992
def test_cfg5_vit_large_mae_step_vs_oracle(dev):
    """BASELINE config 5's step (main.py:69-125 with --mask --masked_recon, constructor grid 4 x 62) on ViT-L
    at T = 992, B = 4, a fixed 75 % mask: BT + reconstruction loss rel <= 3e-2 against the fp32 oracle, gradients rel <= 2e-2
    against the mirror oracle (encoder visited by two passes: masked view 1, unmasked view 2)."""
    from oracle import heads as oheads, rounding as R, step as ostep, vit as ovit
    B, T, L = 4, 992, 248
    g = torch.Generator().manual_seed(3)
    mask = torch.zeros(B, L)
    for b in range(B):
        mask[b, torch.randperm(L, generator=g)[:186]] = 1                        # keep 62 of 248
    cfg = hp.make_args(model_type="vit_large", batch_size=B, crop_frames=T, dataset="audioset", projector_hidden_dim=2048,
                       projector_out_dim=128, masked_recon=True, mask=True, mask_ratio=mask.to(dev))
    tr = BarlowTwinsTrainer(cfg, dev, mode="mae", batch_per_rank=B, clip_samples=158720, seed=0, from_waveform=False)
    sd0 = cpu_state(tr.online)
    views = correlated_views(B, T, seed=5)
    loss = float(tr.step_views([v.to(dev) for v in views]))
    torch.cuda.synchronize()
    got = flat_grads(tr)

    def oracle_loss(sd):
        enc = {k[len("backbone.encoder.encoder."):]: v for k, v in sd.items() if k.startswith("backbone.encoder.encoder.")}
        head = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
        lat_t, recon = ovit.forward(views[0], enc, 16, (4, 62), mask=mask, masked_recon=True, dec_heads=6)
        lat_s = ovit.forward(views[1], enc, 16, (4, 62))
        zt, _ = oheads.head_forward(lat_t, head, 1)
        zs, _ = oheads.head_forward(lat_s, head, 1)
        bt, _ = oheads.bt_forward(zs, zt, 2, ngcrops_each=1)
        return bt + recon, recon

    names = [k for k in ostep.trainable_names(sd0)]

    def oracle_grads(mirror):
        leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd0.items()}
        with R.mirror_hip_bf16(mirror):
            val, rec = oracle_loss(leaf)
            gs = torch.autograd.grad(val, [leaf[k] for k in names], allow_unused=True)
        return float(val.detach()), float(rec.detach()), {k: g_ for k, g_ in zip(names, gs) if g_ is not None}

    ref, recon, fgrads = oracle_grads(False)
    assert recon > 0 and abs(loss - ref) / abs(ref) <= 3e-2, (loss, ref, recon)
    mref, _, mgrads = oracle_grads(True)
    print("cfg5 ViT-L MAE: loss", loss, "oracle", ref, "mirror", mref)
    check_step_gradients("cfg5 step", got, mgrads, fgrads, 330)


# ------------------------------------------------------------------------------------------------ full-size property runs
@pytest.mark.parametrize("name,model_type,frames,B,mode", [
    ("cfg2", "vit_tiny", 1001, 256, "bt"),
    ("cfg3", "vit_base", 1001, 128, "bt"),
    ("cfg4", "vit_base", 1001, 128, "byol"),
    ("cfg5", "vit_large", 992, 256, "mae"),
])
def test_full_size_properties(dev, name, model_type, frames, B, mode):
    """BASELINE.json's per-GPU shapes, waveform in -> AdamW out, three steps on one batch: every loss finite, the loss falls,
    every trainable parameter received a non-zero finite gradient, and the flat bf16 weight copy tracks the fp32 master."""
    extra = dict(masked_recon=True, mask=True, mask_ratio=0.75) if mode == "mae" else {}
    cfg = hp.make_args(model_type=model_type, batch_size=B, crop_frames=frames, dataset="audioset", stop_gradient=(mode == "byol"),
                       predictor=(mode == "byol"), **extra)
    n_samples = 160000 if frames == 1001 else 158720
    tr = BarlowTwinsTrainer(cfg, dev, mode=mode, batch_per_rank=B, clip_samples=n_samples, seed=0)
    g = torch.Generator(device=dev).manual_seed(1234)
    t = torch.arange(n_samples, device=dev, dtype=torch.float32) / 16000.0
    w = 0.1 * torch.randn(B, n_samples, device=dev, generator=g)
    for _ in range(3):
        w += (0.05 + 0.45 * torch.rand(B, 1, device=dev, generator=g)) * torch.sin(2 * torch.pi * (100.0 + 6900.0 * torch.rand(B, 1, device=dev, generator=g)) * t)
    views = tr.make_views(w)
    if mode == "mae":
        torch.manual_seed(0)                                              # the random mask is drawn per step: fix its stream
    losses = []
    for it in range(3):
        if mode == "mae":
            torch.manual_seed(7)                                          # same mask every step, so the loss is comparable
        losses.append(float(tr.step_views(views)))
        if it == 0:
            bad = [k for k, gr in flat_grads(tr).items() if not (torch.isfinite(gr).all() and float(gr.abs().max()) > 0)]
            assert not bad, bad[:5]
            n_train = sum(1 for p in tr.online.parameters() if p.requires_grad)
            assert len(flat_grads(tr)) == n_train
    torch.cuda.synchronize()
    print(name, "losses", losses)
    assert all(np.isfinite(losses)) and losses[2] < losses[0], losses
    assert torch.equal(tr.flat.params_bf16.float(), tr.flat.params.to(torch.bfloat16).float())
    del tr
    torch.cuda.empty_cache()


def test_trainer_local_crops_step_vs_oracle(dev):
    """BarlowTwinsTrainer(mode='bt') with cfg.local_crops_number = 2 runs main.py:86-119's step on the batched device
    path -- teacher = global view 1, student = global view 2 + two 16 x 16 local crops (their own width group through the encoder:
    N = 2 tokens), BarlowTwinsLoss(ncrops = L + 2) averaging L + 1 terms -- against the CPU oracle on the same weights and crops; then
    the whole path from log-mels (augmentation launches included) for three steps."""
    from oracle import step as ostep, heads as oheads
    L, B = 2, 8
    cfg = hp.make_args(model_type="vit_tiny", batch_size=B, crop_frames=96, projector_hidden_dim=256, projector_out_dim=64, local_crops_number=L)
    tr = BarlowTwinsTrainer(cfg, dev, mode="bt", batch_per_rank=B, clip_samples=15200, seed=0, from_waveform=False)
    assert tr.criterion.ncrops == L + 2
    g = torch.Generator().manual_seed(8)
    base = torch.randn(B, 1, 64, 96, generator=g)
    crops = [base + 0.3 * torch.randn(B, 1, 64, 96, generator=g), base + 0.3 * torch.randn(B, 1, 64, 96, generator=g)]
    crops += [base[:, :, 20:36, 16 * (l + 1):16 * (l + 2)].clone() + 0.1 * torch.randn(B, 1, 16, 16, generator=g) for l in range(L)]
    sd = {k: v.detach().float().cpu().clone() for k, v in tr.online.state_dict().items() if "num_batches" not in k}
    loss = float(tr.step_views([c.to(dev) for c in crops]))
    zt, _ = ostep.network_forward(sd, crops[:1], 1, 3, (4, 6))
    zs, _ = ostep.network_forward(sd, crops[1:], L + 1, 3, (4, 6))
    ref, _ = oheads.bt_forward(zs, zt, L + 2, ngcrops_each=1)
    print("trainer, local crops: loss", loss, "oracle", float(ref))
    assert abs(loss - float(ref)) <= 3e-2 * abs(float(ref)), (loss, float(ref))
    grads = flat_grads(tr)
    bad = [k for k, gr in grads.items() if not (torch.isfinite(gr).all() and float(gr.abs().max()) > 0)]
    assert not bad, bad[:5]
    # from log-mels: the augmentation's two launches (globals, locals) feed the same step; the loss falls on a fixed batch
    lms = torch.randn(B, 1, 64, 96, generator=g).to(dev)
    losses = [float(tr.step(lms)) for _ in range(3)]
    assert all(np.isfinite(losses)), losses
    with pytest.raises(NotImplementedError):
        BarlowTwinsTrainer(cfg, dev, mode="byol", batch_per_rank=B, clip_samples=15200, seed=0, from_waveform=False)


def test_trainer_honours_mask_flags_of_both_drivers(dev):
    """`--mask [--masked_recon]` as the two drivers apply them: main.py:69-125 masks (and reconstructs) the TEACHER's view only -- the
    trainer's mode 'bt' with those flags is exactly its mode 'mae' (BASELINE config 5), bit for bit on the same generator state;
    main_bt_byol.py:79-114 masks both views of the ONLINE encoder, the target sees them whole -- with mask_ratio 0 that is the unmasked
    step bit for bit, with 0.25 a different, finite loss with every gradient alive (reconstruction loss added when asked)."""
    g = torch.Generator().manual_seed(9)
    views = [torch.randn(8, 1, 64, 96, generator=g).to(dev), torch.randn(8, 1, 64, 96, generator=g).to(dev)]
    kw = dict(model_type="vit_tiny", batch_size=8, crop_frames=96, projector_hidden_dim=256, projector_out_dim=64)

    def step(mode, seed=123, mask_ratio=None, **flags):
        cfg = hp.make_args(**kw, **flags)
        tr = BarlowTwinsTrainer(cfg, dev, mode=mode, batch_per_rank=8, clip_samples=15200, seed=0, from_waveform=False)
        torch.manual_seed(seed)                      # the masking noise comes from torch's generator (models/mae.py:332)
        loss = float(tr.step_views(views, mask_ratio=mask_ratio))
        torch.cuda.synchronize()
        return loss, tr
    l_mae, t_mae = step("mae", masked_recon=True, mask=True, mask_ratio=0.5)
    l_bt, t_bt = step("bt", masked_recon=True, mask=True, mask_ratio=0.5)
    assert l_mae == l_bt and torch.equal(t_mae.flat.grads, t_bt.flat.grads) and torch.equal(t_mae.flat.params, t_bt.flat.params)
    l_plain, t_plain = step("bt")
    l_mask, t_mask = step("bt", mask=True, mask_ratio=0.5)        # masking without the decoder
    assert np.isfinite(l_mask) and l_mask != l_plain and l_mask != l_bt
    byol = dict(stop_gradient=True, predictor=True)
    l0, t0 = step("byol", **byol)
    l0m, t0m = step("byol", mask=True, mask_ratio=0.0, **byol)
    assert l0 == l0m and torch.equal(t0.flat.grads, t0m.flat.grads)
    l25, t25 = step("byol", mask=True, mask_ratio=0.25, **byol)
    assert np.isfinite(l25) and l25 != l0
    bad = [k for k, gr in flat_grads(t25).items() if not (torch.isfinite(gr).all() and float(gr.abs().max()) > 0)]
    assert not bad, bad[:5]
    l25r, t25r = step("byol", mask=True, mask_ratio=0.25, masked_recon=True, **byol)
    # + reconstruction loss (main_bt_byol.py:112-114): the decoder takes part (a different initialisation sequence: losses not comparable)
    gr = flat_grads(t25r)
    assert np.isfinite(l25r) and float(gr["backbone.encoder.encoder.mask_token"].abs().max()) > 0
    assert float(gr["backbone.encoder.encoder.decoder_pred.weight"].abs().max()) > 0
    # graph capture: a FIXED mask ratio is capturable since round 5 (test_graph_replay_equals_eager[mae]); a ratio that changes from step to
    # step is refused (the kept-token count is baked into the captured launches), and a captured step refuses another ratio
    t25.mask_ratio_schedule = [0.1, 0.2, 0.3]
    with pytest.raises(NotImplementedError):
        t25.enable_graph()
    t25.mask_ratio_schedule = None
    t25.enable_graph()
    with pytest.raises(RuntimeError, match="mask_ratio"):
        t25.step_views(views, mask_ratio=0.5)
    assert np.isfinite(float(t25.step_views(views)))


def test_step_through_the_dispatcher_equals_direct(dev):
    """the same three steps with every kernel call routed through `torch.ops.ssl_audio.*` (torch.library custom
    operators: ops.route_through_dispatcher) and with the direct ctypes calls -- same kernels, so bit-identical losses, gradients and
    weights (the path holds no float atomics); and the routed step really passes the dispatcher (the operators' call counts move)."""
    cfg = hp.make_args(model_type="vit_tiny", batch_size=8, crop_frames=96, projector_hidden_dim=512, projector_out_dim=128)
    g = torch.Generator().manual_seed(6)
    batches = [[torch.randn(8, 1, 64, 96, generator=g).to(dev), torch.randn(8, 1, 64, 96, generator=g).to(dev)] for _ in range(3)]

    def run():
        tr = BarlowTwinsTrainer(cfg, dev, mode="bt", batch_per_rank=8, clip_samples=15200, seed=0, from_waveform=False)
        losses = [float(tr.step_views(v)) for v in batches]
        torch.cuda.synchronize()
        return losses, tr.flat.params.clone(), tr.flat.grads.clone(), tr.flat.m.clone()

    direct = run()
    import ssl_audio_amd.custom_ops as co
    calls = {"n": 0}
    orig = ops._DIRECT_FNS["gemm"] if hasattr(ops, "_DIRECT_FNS") and "gemm" in ops._DIRECT_FNS else ops.gemm
    try:
        ops.route_through_dispatcher(True)
        assert ops.DISPATCH == "torch_ops" and ops.gemm is torch.ops.ssl_audio.gemm
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            routed = run()
        names = {e.key for e in prof.key_averages()}
        assert {"ssl_audio::gemm", "ssl_audio::layernorm_fwd", "ssl_audio::attention_bwd", "ssl_audio::adamw_step_dev"} <= names, sorted(n for n in names if "ssl_audio" in n)
    finally:
        ops.route_through_dispatcher(False)
    assert ops.DISPATCH == "direct" and ops.gemm is orig
    assert direct[0] == routed[0], (direct[0], routed[0])
    for a, b, what in zip(direct[1:], routed[1:], ("params", "grads", "m")):
        assert torch.equal(a, b), what
    del co, calls


@pytest.mark.parametrize("mode", ["bt", "byol", "mae", "bt_local"])
def test_graph_replay_equals_eager(dev, mode):
    """the device part of the step captured into ONE HIP graph (BarlowTwinsTrainer.enable_graph) and replayed takes
    the same steps as the eager path, and two eager runs take the same steps as each other -- BIT FOR BIT over five steps: losses,
    gradients, weights, Adam moments, EMA target and predictor, with the learning rate CHANGING between steps (it reaches the captured
    AdamW launches through device memory).  No kernel on this path sums floats in an order that depends on scheduling (SA_DETERMINISTIC,
    the default: split-K slices, bias column sums, the CLS-token gradient and the loss scalar all add their partials in a fixed order),
    so any difference here is a race, not rounding."""
    # round 5 (VERDICT r4 missing #5): "mae" -- view 1 through the 75 %-masked encoder + MAE decoder: the masking indices are device-side
    # bookkeeping (torch.rand / argsort) INSIDE the captured step, a replay advances the generator like an eager step -- and "bt_local":
    # two 16 x 16 local crops through their own width group, static input buffers for them too.
    assert ops.DETERMINISTIC_WGRAD, "run with SA_DETERMINISTIC unset or 1"
    tmode = "bt" if mode == "bt_local" else mode
    L = 2 if mode == "bt_local" else 0
    extra = dict(masked_recon=True, mask=True, mask_ratio=0.75) if mode == "mae" else {}
    cfg = hp.make_args(model_type="vit_tiny", batch_size=8, crop_frames=96, projector_hidden_dim=512, projector_out_dim=128,
                       stop_gradient=(mode == "byol"), predictor=(mode == "byol"), local_crops_number=L, **extra)
    g = torch.Generator().manual_seed(4)
    base = [torch.randn(8, 1, 64, 96, generator=g) for _ in range(5)]
    batches = [[(b + 0.3 * torch.randn(8, 1, 64, 96, generator=g)).to(dev), (b + 0.3 * torch.randn(8, 1, 64, 96, generator=g)).to(dev)] +
               [torch.randn(8, 1, 16, 16, generator=g).to(dev) for _ in range(L)] for b in base]
    lrs = [1e-4, 3e-4, 2e-4, 5e-5, 1e-4]

    def flats(tr):
        out = {"online": tr.flat}
        if mode == "byol":
            out["target"], out["pred"] = tr.flat_target, tr.flat_pred
        return out

    def run(graph):
        torch.manual_seed(77); torch.cuda.manual_seed(77)           # (the masking noise comes from the device generator)
        tr = BarlowTwinsTrainer(cfg, dev, mode=tmode, batch_per_rank=8, clip_samples=15200, seed=0, from_waveform=False)
        torch.cuda.manual_seed(78)
        losses, trace = [], []
        for i, v in enumerate(batches):
            for grp in tr.param_groups:
                grp["lr"] = lrs[i]
            losses.append(float(tr.step_views(v)))
            snap = {}
            for fname, fl in flats(tr).items():
                for t in ("grads", "params", "m", "v"):
                    if getattr(fl, t, None) is not None:
                        snap[f"{fname}.{t}"] = getattr(fl, t).clone()
            trace.append(snap)
            if i == 0 and graph:
                tr.enable_graph()
        torch.cuda.synchronize()
        assert (tr._graph is not None) == graph
        return tr, losses, trace

    def first_difference(ta, tb, what):
        """First step / buffer / parameter at which two runs differ (the failure message names the tensor to look at)."""
        for i, (sa, sb) in enumerate(zip(ta, tb)):
            for k in sa:                                       # insertion order: grads before params before moments
                if not torch.equal(sa[k], sb[k]):
                    d = (sa[k] - sb[k]).abs()
                    idx = int(d.argmax())
                    return f"{what}: step {i}, buffer {k}, first/worst element {idx}: {float(sa[k].view(-1)[idx])!r} vs {float(sb[k].view(-1)[idx])!r} (max |diff| {float(d.max()):.3e}, {int((d > 0).sum())} of {d.numel()} differ)"
        return None

    a, la, ta = run(False)
    a2, la2, ta2 = run(False)
    b, lb, tb = run(True)
    print(mode, "eager losses", la, "eager again", la2, "graph losses", lb)
    assert a.flat.step_count == b.flat.step_count == 5
    diff = first_difference(ta, ta2, "eager vs eager")
    assert diff is None, diff
    assert la == la2, (la, la2)
    diff = first_difference(ta, tb, "eager vs graph replay")
    assert diff is None, diff
    assert la == lb, (la, lb)
    del a2
    # a non-finite loss leaves weights and moments untouched (device-side gate), and the host notices on its next look
    before = b.flat.params.clone(), b.flat.m.clone()
    bad = [torch.full_like(batches[0][0], float("nan")), batches[0][1]]
    b.step_views(bad)
    torch.cuda.synchronize()
    assert torch.equal(b.flat.params, before[0]) and torch.equal(b.flat.m, before[1])
    with pytest.raises(FloatingPointError):
        b.assert_finite()
