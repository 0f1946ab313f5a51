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
# the per-clip instantiation (ABI v6: starts / lengths / offsets as device vectors) on the same work, then the reference's default crop
# (96 frames at each clip's own random start out of the 1001) and 1 s clips right-padded to 1001 frames
zeros = torch.zeros(B, dtype=torch.int32, device=dev)
full = torch.full((B,), L, dtype=torch.int32, device=dev)
o1 = torch.empty(B, 1, 64, T, device=dev)
us = timed(lambda: fe(wave, crop_frames=T, start=zeros, lengths=full, offsets=zeros, norm_stats=(-0.8294, 4.6230), out=o1))
print(f"  per-clip geometry, same work   {us:7.1f} us")
st = torch.randint(0, T - 96, (B,), device=dev).int()
o2 = torch.empty(B, 1, 64, 96, device=dev)
us = timed(lambda: fe(wave, crop_frames=96, start=st, norm_stats=(-0.8294, 4.6230), out=o2))
print(f"  96-frame crops at per-clip starts   {us:7.1f} us  ({4.0 * B * (95 * 160 + 1024) / us / 1e3:.0f} GB/s of samples touched)")
short = torch.full((B,), 16000, dtype=torch.int32, device=dev)
us = timed(lambda: fe(wave, crop_frames=T, lengths=short, norm_stats=(-0.8294, 4.6230), out=o1))
print(f"  1 s clips padded to {T} frames   {us:7.1f} us")
import time
augh = BatchedPairAugment(dev, 64, T, T, True, True, True, 0.2, seed=1)
augh.next_slots(256)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); augh.draw(256, src_frames=T); ts.append((time.perf_counter() - t0) * 1e3); augh.clips += 256
print(f"host sampler, 256 clips (512 views): {min(ts):.2f} ms (min of 5; {os.cpu_count()} host threads visible)")
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
