"""smoke(): one tiny Barlow Twins pre-training step of the HIP hot path on cuda:0, checked against the CPU oracle.

Only __graft_entry__.smoke() and the tests call this; it is the one place inside the package that imports
`oracle` (as the checker, lazily, never on the product path).
"""
import numpy as np
import torch

from . import hyperparameters as hp
from .train import AUDIOSET_STATS, BarlowTwinsTrainer


def synthetic_waveforms(n, n_samples, first_clip=0, device="cpu"):
    """Clip k: torch.Generator seeded 1234+k; 0.1*N(0,1) + 3 sinusoids (f ~ U[100,7000] Hz, amp ~ U[0.05,0.5])  (BASELINE.md §3)."""
    out = torch.empty(n, n_samples)
    t = torch.arange(n_samples, dtype=torch.float64) / 16000.0
    for k in range(n):
        g = torch.Generator().manual_seed(1234 + first_clip + k)
        w = 0.1 * torch.randn(n_samples, generator=g, dtype=torch.float64)
        for _ in range(3):
            f = 100.0 + 6900.0 * torch.rand(1, generator=g, dtype=torch.float64)
            a = 0.05 + 0.45 * torch.rand(1, generator=g, dtype=torch.float64)
            w = w + a * torch.sin(2 * np.pi * f * t)
        out[k] = w.float()
    return out.to(device)


def oracle_step_from_trainer(trainer, waves, lengths=None):
    """Re-run the trainer's step on the CPU oracle: same waveforms, same parameters, same augmentation draws (incl. each clip's dataset
    crop start, datasets.py:342-345; `lengths`: per-clip sample counts when the rows are padded)."""
    from oracle import augment as oaug, frontend as ofe, step as ostep
    cfg = trainer.cfg
    w = waves.cpu().numpy()
    starts = trainer.augment.starts or [0] * w.shape[0]
    lms = np.stack([ofe.crop_pad_normalize(ofe.logmel(w[b, :(lengths[b] if lengths is not None else w.shape[1])]), cfg.crop_frames, starts[b],
                                           *AUDIOSET_STATS) for b in range(w.shape[0])])                  # [B, 64, T]
    B = lms.shape[0]
    views = [np.zeros((B, 1, cfg.n_mels, cfg.crop_frames)), np.zeros((B, 1, cfg.n_mels, cfg.crop_frames))]
    canvas = (cfg.n_mels, int(cfg.crop_frames * cfg.virtual_crop_scale[1]))
    for r in trainer.augment.records:
        b = r["clip"] % B
        x = lms[b][None]
        k = r["bank_index"]
        if k >= 0:
            c = min(2 * r["clip"] + r["view"], trainer.augment.n)
            g = 2 * r["clip"] + r["view"] - c + k
            x = oaug.mixup_apply(x, lms[(g // 2) % B][None], r["alpha"])
        y = oaug.rrc_apply(x, r["rrc"], (cfg.n_mels, cfg.crop_frames), tuple(cfg.virtual_crop_scale))
        y = oaug.linear_fader_apply(y, *r["head_tail"])
        views[r["view"]][b] = y
    tv = [torch.from_numpy(v).float() for v in views]
    sd = {k: v.detach().cpu().clone() for k, v in trainer._initial_state.items()}
    heads = {"tiny": 3, "small": 6, "base": 12, "large": 16}[cfg.model_type.split("_")[-1]]
    opt = ostep.AdamW(cfg.lr, cfg.wd)
    loss, grads = ostep.bt_step(sd, tv, heads, (4, 6), opt)
    return loss, grads, sd, tv


def smoke(n_clips=8, seconds=1.0, verbose=True):
    dev = torch.device("cuda:0")
    cfg = hp.make_args(model_type="vit_tiny", batch_size=n_clips, crop_frames=int(seconds * 16000) // 160 + 1,
                       projector_hidden_dim=512, projector_out_dim=128)
    n_samples = int(seconds * 16000)
    trainer = BarlowTwinsTrainer(cfg, dev, mode="bt", batch_per_rank=n_clips, clip_samples=n_samples, seed=0)
    trainer._initial_state = {k: v.detach().clone() for k, v in trainer.online.state_dict().items()}
    waves = synthetic_waveforms(n_clips, n_samples, device=dev)
    loss = float(trainer.step(waves))
    torch.cuda.synchronize()
    ref_loss, _, sd_after, _ = oracle_step_from_trainer(trainer, waves)
    rel = abs(loss - ref_loss) / abs(ref_loss)
    # parameters after one AdamW step: every element moved by at most lr (Adam), compare a dense weight
    key = "backbone.encoder.encoder.blocks.0.mlp.fc1.weight"
    got = trainer.online.state_dict()[key].cpu()
    moved = float((got - trainer._initial_state[key].cpu()).abs().max())
    diff = float((got - sd_after[key]).abs().max())
    if verbose:
        print(f"smoke: HIP loss {loss:.5f}  oracle loss {ref_loss:.5f}  rel diff {rel:.2e}; fc1 max|dp| {moved:.2e}, vs oracle {diff:.2e}")
    if not np.isfinite(loss) or rel > 5e-2:
        raise AssertionError(f"smoke: loss mismatch vs CPU oracle: {loss} vs {ref_loss}")
    if not (0 < moved <= 1.01 * cfg.lr + 1e-9):
        raise AssertionError(f"smoke: optimiser step moved fc1 by {moved}, expected (0, lr={cfg.lr}]")
    return loss, ref_loss
