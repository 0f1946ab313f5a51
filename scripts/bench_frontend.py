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
# the kernel alone: one batch's parameters are drawn and uploaded ONCE (host-side sampling and the three small H2D copies of a real
# step are not part of the launch), then the same launch is repeated
aug.clips -= B
src, mix, par, canvas = aug.draw(B)
aug.clips += B
p_dev = torch.tensor(par, dtype=torch.float32).reshape(-1, 8).to(dev)
s_dev = torch.tensor(src, dtype=torch.int32).to(dev)
m_dev = torch.tensor(mix, dtype=torch.int32).to(dev)
ratio = canvas[1] / max(T - 1, 1)
us = timed(lambda: ops.augment_views(aug.store, 64 * T, s_dev, m_dev, p_dev, out.view(2 * B, 1, 64, T), 64, T, canvas, ratio, True))
nb = 2 * B * 4.0 * (2 * 64 * T + 64 * T)
print(f"augment_kernel  {us:7.1f} us  {nb / us / 1e3:7.1f} GB/s algorithmic ({nb / 1e6:.0f} MB)")
