"""Does it train?  N steps of the full pipeline (log-mel frontend -> augmentation -> encoder + projector -> BT loss -> backward -> AdamW) on a
fixed pool of synthetic clips (noise + 3 sinusoids each), printing the loss.  With a fixed pool the Barlow Twins loss
must fall steadily (the two views of a clip share its sinusoids); a NaN, a plateau at the initial value or an explosion shows up here.
   python scripts/train_sanity.py [tiny|base] [steps] [clips] [crop_frames]
crop_frames < 1001 (e.g. the reference's default 96): every clip is cropped at its own random start per step and a quarter of the clips are
given shorter lengths (datasets.py:342-351 through the batched frontend's per-clip vectors)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import hyperparameters as hp
from ssl_audio_amd.train import BarlowTwinsTrainer

size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
crop = int(sys.argv[4]) if len(sys.argv) > 4 else 1001
dev = torch.device("cuda:0")
n = 160000
cfg = hp.make_args(model_type="vit_" + size, batch_size=B, crop_frames=crop, dataset="audioset")
lengths = None if crop == 1001 else [n if b % 4 else 160 * (crop // 2 + 7 * (b % 5)) for b in range(B)]   # every 4th clip shorter than the crop: right pad
tr = BarlowTwinsTrainer(cfg, dev, mode="bt", batch_per_rank=B, clip_samples=n, seed=0)
g = torch.Generator(device=dev).manual_seed(1234)
t = torch.arange(n, device=dev, dtype=torch.float32) / 16000.0
pool = []
for _ in range(4):
    w = 0.1 * torch.randn(B, n, device=dev, generator=g)
    for _ in range(3):
        f = 100.0 + 6900.0 * torch.rand(B, 1, device=dev, generator=g)
        a = 0.05 + 0.45 * torch.rand(B, 1, device=dev, generator=g)
        w += a * torch.sin(2 * torch.pi * f * t)
    pool.append(w)
losses = []
for s in range(steps):
    losses.append(tr.step(pool[s % len(pool)], lengths=lengths))
    if (s + 1) % max(steps // 10, 1) == 0:
        torch.cuda.synchronize()
        recent = torch.stack([l.detach().float() for l in losses[-max(steps // 10, 1):]]).mean().item()
        print(f"step {s + 1:4d}  mean loss of the last {max(steps // 10, 1)} steps: {recent:9.3f}", flush=True)
tr.assert_finite()
first = torch.stack([l.detach().float() for l in losses[:10]]).mean().item()
last = torch.stack([l.detach().float() for l in losses[-10:]]).mean().item()
print(f"vit_{size}, {B} clips/step, {steps} steps, crop {crop}: loss {first:.2f} -> {last:.2f}" + ("" if lengths is None else f"; last batch's crop starts {tr.augment.starts[:8]} ..."))
assert last < 0.7 * first, "the loss did not fall"
