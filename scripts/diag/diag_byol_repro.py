"""Diagnostic: two identically seeded eager BYOL trainers stepped side by side on the same views; prints, per step, the losses and the
largest difference of the flat weights / Adam moments / gradients between the two.  Run-to-run differences start at fp32
summation-order noise (atomic bias-gradient column sums); the question is how fast a step amplifies them (AdamW normalises
noise-level gradients to +-lr steps) as opposed to a sudden jump (which would be a race)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssl_audio_amd import hyperparameters as hp
from ssl_audio_amd.train import BarlowTwinsTrainer

dev = torch.device("cuda:0")
if os.environ.get("SA_DIAG_NANFILL") == "1":      # torch.empty() returns NaN-filled memory: any read of uninitialised memory becomes visible
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
mode = sys.argv[1] if len(sys.argv) > 1 else "byol"
cfg = hp.make_args(model_type="vit_tiny", batch_size=8, crop_frames=96, projector_hidden_dim=512, projector_out_dim=128,
                   stop_gradient=(mode == "byol"), predictor=(mode == "byol"))
g = torch.Generator().manual_seed(4)
base = [torch.randn(8, 1, 64, 96, generator=g) for _ in range(5)]
batches = [[(b + 0.3 * torch.randn(8, 1, 64, 96, generator=g)).to(dev), (b + 0.3 * torch.randn(8, 1, 64, 96, generator=g)).to(dev)] for b in base]
lrs = [1e-4, 3e-4, 2e-4, 5e-5, 1e-4]
A = BarlowTwinsTrainer(cfg, dev, mode=mode, batch_per_rank=8, clip_samples=15200, seed=0, from_waveform=False)
B = BarlowTwinsTrainer(cfg, dev, mode=mode, batch_per_rank=8, clip_samples=15200, seed=0, from_waveform=False)
print("initial weight difference", float((A.flat.params - B.flat.params).abs().max()))
for i, v in enumerate(batches):
    for tr in (A, B):
        for grp in tr.param_groups:
            grp["lr"] = lrs[i]
    la = float(A.step_views(v)); ga = A.flat.grads.clone()
    lb = float(B.step_views(v)); gb = B.flat.grads.clone()
    dg = (ga - gb).abs()
    k = int(dg.argmax())
    name = next((n for n, (off, cnt) in A.flat.offsets.items() if off <= k < off + cnt), "?")
    print(f"step {i}: loss {la:.6f} {lb:.6f} | max|dgrad| {float(dg.max()):.3e} (|g| there {float(ga[k].abs()):.3e}, {name}) rel-norm {float(dg.norm() / ga.norm()):.2e}"
          f" | max|dparam| {float((A.flat.params - B.flat.params).abs().max()):.3e} | max|dm| {float((A.flat.m - B.flat.m).abs().max()):.3e}")
    if float(dg.norm() / ga.norm()) > 1e-6:
        print("  per-parameter relative gradient difference (model order):")
        for n, (off, cnt) in A.flat.offsets.items():
            a, b = ga[off:off + cnt], gb[off:off + cnt]
            r = float((a - b).norm() / (a.norm() + 1e-30))
            if r > 1e-7:
                print(f"    {n:60s} {r:.2e}")
        break
