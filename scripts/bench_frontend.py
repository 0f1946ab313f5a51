"""Stand-alone timing of the frontend / augmentation launches at the headline shape (128 clips of 10 s): logmel_kernel, augment_kernel.
   python scripts/bench_frontend.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops
from ssl_audio_amd.frontend import MelSpectrogram
from ssl_audio_amd.augmentations import BatchedPairAugment
dev = torch.device("cuda:0")
B, L, T = 128, 160000, 1001
g = torch.Generator(device=dev).manual_seed(0)
wave = 0.1 * torch.randn(B, L, device=dev, generator=g)
fe = MelSpectrogram()
aug = BatchedPairAugment(dev, 64, T, T, True, True, True, 0.2, seed=0)


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def frontend():
    slots = aug.next_slots(B)
    fe(wave, crop_frames=T, start=0, norm_stats=(-0.8294, 4.6230), out=slots.view(B, 1, 64, T))
    aug.clips += B          # (advance the ring as a real step does)


us = timed(frontend)
nb = 4.0 * B * L + 4.0 * B * 64 * T
print(f"logmel_kernel   {us:7.1f} us  {nb / us / 1e3:7.1f} GB/s algorithmic ({nb / 1e6:.0f} MB)")
out = torch.empty(2, B, 1, 64, T, device=dev)


def augment():
    aug.clips -= B if aug.clips >= B else 0
    aug(B, out=out)


# host-side sampling + H2D of the parameters is part of the call: time the launch alone through HIP events inside ops via STREAM_PROFILE
ops.STREAM_PROFILE = {}
for _ in range(12): augment()
torch.cuda.synchronize()
recs = ops.STREAM_PROFILE.get("augment_kernel", [])[2:]
ops.STREAM_PROFILE = None
us = sum(r[0].elapsed_time(r[1]) for r in recs) / max(len(recs), 1) * 1e3
nb = recs[0][2] if recs else 0.0
print(f"augment_kernel  {us:7.1f} us  {nb / us / 1e3:7.1f} GB/s algorithmic ({nb / 1e6:.0f} MB)")
