"""HEAR wrappers and checkpoint-key compatibility (hear/sample/vit.py:64-77,129-247, linear.py:114-133)."""
import os
import tempfile
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ssl_audio_amd import engine, hyperparameters as hp, ops  # noqa: E402
from ssl_audio_amd.hear import utils as hutils, vit as hvit  # noqa: E402
from ssl_audio_amd.train import BarlowTwinsTrainer  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    ops.lib()
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_hear_scene_and_timestamp_embeddings_vs_oracle(dev):
    """load_model(vit_tiny, 16x16) -> get_scene_embeddings / get_timestamp_embeddings on 1.7 s clips against oracle/hear.py with the
    same weights: bf16 encoder tolerance 2e-2 (the reference's statistics quirk of hear/utils.py:47-50 included).  Mel arithmetic:
    vs our restatement of torchaudio's spec (reference parity unpinned, as for the training frontend)."""
    from oracle import hear as ohear
    torch.manual_seed(0)
    model = hvit.load_model("", "vit_tiny", "16x16")
    assert model.scene_embedding_size == 192 and model.timestamp_embedding_size == 192 * 4 and model.sample_rate == 16000
    g = torch.Generator().manual_seed(1)
    audio = 0.3 * torch.randn(3, 27200, generator=g)
    sd = {k: v.detach().cpu() for k, v in model.model.state_dict().items()}
    scene = hvit.get_scene_embeddings(audio.to(dev), model)
    ref = ohear.scene_embeddings(audio.numpy(), sd, 3, (4, 6), 96)
    assert scene.shape == (3, 192) and rel(scene, ref) < 2e-2, rel(scene, ref)
    emb, ts = hvit.get_timestamp_embeddings(audio.to(dev), model)
    ref_e, ref_t = ohear.timestamp_embeddings(audio.numpy(), sd, 3, (4, 6), 96)
    assert emb.shape == ref_e.shape and np.allclose(ts.numpy(), ref_t) and rel(emb, ref_e) < 2e-2, rel(emb, ref_e)
    tsx = model._get_timestamps(audio, emb)
    assert tsx.shape == (3, emb.shape[1])


def test_checkpoint_keys_round_trip(dev):
    """What main_bt_byol.py:492-503 saves ({'model': online.state_dict()}, DDP-prefixed or not) loads into (i) the HEAR wrapper
    (hear/sample/vit.py:64-77), (ii) the linear-probe encoder (linear.py:114-133) and (iii) a FlatState-backed trainer, whose pinned
    bf16 weight copies must follow the load (engine._Bf16Cache refreshes on the version bump): identical embeddings afterwards."""
    from ssl_audio_amd import model as smodel
    cfg = hp.make_args(model_type="vit_tiny", batch_size=8, crop_frames=96, projector_hidden_dim=512, projector_out_dim=128)
    tr1 = BarlowTwinsTrainer(cfg, dev, mode="bt", batch_per_rank=8, clip_samples=15200, seed=0, from_waveform=False)
    g = torch.Generator().manual_seed(2)
    views = [torch.randn(8, 1, 64, 96, generator=g).to(dev), torch.randn(8, 1, 64, 96, generator=g).to(dev)]
    tr1.step_views(views)                                                 # weights now differ from any seed's initialisation
    sd = tr1.online.state_dict()
    with tempfile.TemporaryDirectory() as d:
        for prefix in ("", "module."):
            path = os.path.join(d, f"ckpt{len(prefix)}.pth")
            torch.save({"model": {prefix + k: v for k, v in sd.items()}, "epoch": 1}, path)
            wrapper = hvit.load_model(path, "vit_tiny", "16x16")
            enc = smodel.ModelWrapper(cfg).encoder.to(dev)
            hvit.load_encoder_state_dict(enc, torch.load(path, map_location="cpu"))
            x = views[0]
            with torch.no_grad():
                a = tr1.online.backbone(x)
                assert torch.equal(wrapper.model(x), a) and torch.equal(enc(x), a)
    tr2 = BarlowTwinsTrainer(cfg, dev, mode="bt", batch_per_rank=8, clip_samples=15200, seed=5, from_waveform=False)
    with torch.no_grad():
        before = tr2.online(views, ncrops=2)
    tr2.online.load_state_dict(sd)
    with torch.no_grad():
        z1, z2 = tr1.online(views, ncrops=2), tr2.online(views, ncrops=2)
    assert not torch.equal(before, z2) and torch.equal(z1, z2)
    p = dict(tr2.online.named_parameters())["backbone.encoder.encoder.blocks.3.mlp.fc1.weight"]
    assert torch.equal(engine.BF16_WEIGHTS.get(p).float(), p.detach().to(torch.bfloat16).float())
    l1, l2 = float(tr1.step_views(views)), float(tr2.step_views(views))   # and a step on the loaded weights runs
    assert np.isfinite(l2) and abs(l1 - l2) <= 1e-5 * abs(l1)


def test_hear_wrapper_vs_reference_fixture(dev):
    """The HEAR wrapper against what the REFERENCE's wrapper produced (tests/golden/hear.npz: hear/sample/vit.py run with the micro ViT and
    torchaudio's MelSpectrogram stood in for by our restatement): normalised log-mel 1e-3, scene and timestamp embeddings 2e-2 (bf16
    encoder), timestamps exact."""
    from ssl_audio_amd import mae
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hear.npz"))
    model = hvit.load_model("", "vit_tiny", "16x16")
    micro = mae.MaskedAutoencoderViT(img_size=(64, 96), patch_size=[16, 16], in_chans=1, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4,
                                     norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), use_decoder=False, decoder_embed_dim=64,
                                     decoder_depth=1, decoder_num_heads=1).to(dev)
    micro.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}, strict=True)
    model.model = micro
    model.scene_embedding_size = micro.embed_dim
    model.timestamp_embedding_size = micro.embed_dim * micro.grid_size()[0]
    assert model.timestamp_embedding_size == int(g["timestamp_embedding_size"])
    audio = torch.from_numpy(g["audio"]).to(dev)
    spec = model._to_normalized_spec(audio)
    assert rel(spec, g["norm_spec"]) < 1e-3, rel(spec, g["norm_spec"])
    scene = hvit.get_scene_embeddings(audio, model)
    assert scene.shape == g["scene"].shape and rel(scene, g["scene"]) < 2e-2, rel(scene, g["scene"])
    emb, ts = hvit.get_timestamp_embeddings(audio, model, hop_size=100)
    assert emb.shape == g["ts_emb"].shape and np.allclose(ts.cpu().numpy(), g["ts"], atol=1e-3)
    assert rel(emb, g["ts_emb"]) < 2e-2, rel(emb, g["ts_emb"])


@pytest.mark.parametrize("mode", ["bt", "byol"])
def test_trainer_checkpoint_resume(dev, mode):
    """save -> load -> continue (main_bt_byol.py:492-503, utils/utils.py:37-46): a trainer that loads state_dict() of another after two steps
    takes the same third step (loss and weights; the fused AdamW's moments and step count travel in torch.optim.AdamW's state_dict form),
    and that 'optimizer' entry loads into the reference driver's own torch.optim.AdamW over utils.get_param_groups."""
    from ssl_audio_amd import utils
    cfg = hp.make_args(model_type="vit_tiny", batch_size=8, crop_frames=96, projector_hidden_dim=512, projector_out_dim=128,
                       stop_gradient=(mode == "byol"), predictor=(mode == "byol"))
    g = torch.Generator().manual_seed(4)
    batches = [[torch.randn(8, 1, 64, 96, generator=g).to(dev), torch.randn(8, 1, 64, 96, generator=g).to(dev)] for _ in range(3)]
    a = BarlowTwinsTrainer(cfg, dev, mode=mode, batch_per_rank=8, clip_samples=15200, seed=0, from_waveform=False)
    for v in batches[:2]:
        a.step_views(v)
    a_main_form = a.state_dict(driver="main")["optimizer"] if mode != "byol" else None
    a_m_before, a_v_before = a.flat.m.clone(), a.flat.v.clone()
    a_pm_before = a.flat_pred.m.clone() if mode == "byol" else None
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ckpt.pth")
        utils.save_on_master(a.state_dict(epoch=3), path)
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    b = BarlowTwinsTrainer(cfg, dev, mode=mode, batch_per_rank=8, clip_samples=15200, seed=9, from_waveform=False)
    assert b.load_state_dict(ckpt) == 3 and b.flat.step_count == 2
    assert torch.equal(a.flat.params, b.flat.params) and torch.equal(a.flat.m, b.flat.m) and torch.equal(a.flat.v, b.flat.v)
    assert torch.equal(a.flat.params_bf16, b.flat.params_bf16)
    la, lb = float(a.step_views(batches[2])), float(b.step_views(batches[2]))
    assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
    assert rel(b.flat.params, a.flat.params.cpu()) < 1e-6
    # the 'optimizer' entry in the reference driver's OWN optimiser, built the way get_optimizer builds it (main_bt_byol.py:301-305):
    # one AdamW over get_param_groups(encoder) + get_param_groups(predictor) -- four groups, the last two empty without a predictor
    from ssl_audio_amd.model import BarlowTwinsPredictor
    pred_mod = b.predictor if mode == "byol" else BarlowTwinsPredictor(cfg.projector_out_dim, use=False)
    groups = utils.get_param_groups(b.online)
    groups.extend(utils.get_param_groups(pred_mod))
    opt = torch.optim.AdamW(groups, lr=cfg.lr, weight_decay=cfg.wd)
    opt.load_state_dict(ckpt["optimizer"])
    assert len(opt.param_groups) == 4 and opt.param_groups[1]["weight_decay"] == 0.0 and opt.param_groups[3]["weight_decay"] == 0.0
    p0 = opt.param_groups[0]["params"][0]
    assert opt.state[p0]["exp_avg"].shape == p0.shape and int(float(opt.state[p0]["step"])) == 2
    name0 = [n for n, p in b.online.named_parameters() if p is p0][0]
    off, cnt = a.flat.offsets[name0]
    assert torch.equal(opt.state[p0]["exp_avg"].reshape(-1).to(dev), a_m_before[off:off + cnt])
    if mode == "byol":
        q0 = opt.param_groups[2]["params"][0]
        qn = [n for n, p in b.predictor.named_parameters() if p is q0][0]
        off, cnt = a.flat_pred.offsets[qn]
        assert torch.equal(opt.state[q0]["exp_avg"].reshape(-1).to(dev), a_pm_before[off:off + cnt])
    # ... and the reverse: what that optimiser saves (a real main_bt_byol.py checkpoint) resumes a trainer, predictor moments included
    c = BarlowTwinsTrainer(cfg, dev, mode=mode, batch_per_rank=8, clip_samples=15200, seed=11, from_waveform=False)
    ref_ckpt = {"model": ckpt["model"], "optimizer": opt.state_dict(), "epoch": 5, "barlow_twins_loss": ckpt["barlow_twins_loss"]}
    if mode == "byol":
        ref_ckpt.update(predictor=ckpt["predictor"], target=ckpt["target"])
    assert c.load_state_dict(ref_ckpt) == 5 and c.flat.step_count == 2
    assert torch.equal(c.flat.m, a_m_before) and torch.equal(c.flat.v, a_v_before)
    if mode == "byol":
        assert torch.equal(c.flat_pred.m, a_pm_before) and c.flat_pred.step_count == 2
    else:                                                   # main.py's two-group optimiser state loads as well
        opt2 = torch.optim.AdamW(utils.get_param_groups(b.online), lr=cfg.lr, weight_decay=cfg.wd)
        opt2.load_state_dict(a_main_form)
        c.load_state_dict({"model": ckpt["model"], "optimizer": opt2.state_dict()})
        assert torch.equal(c.flat.m, a_m_before)
