"""Cycle stamps of the log-mel kernel's phases (a library built with -DSA_LOGMEL_DBG=4 writes them over its outputs):
   SA_HIP_LIB=<dbg lib> python scripts/diag/logmel_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssl_audio_amd.frontend import MelSpectrogram
dev = torch.device("cuda:0")
B, L, T = 128, 160000, 1001
wave = 0.1 * torch.randn(B, L, device=dev)
fe = MelSpectrogram()
for _ in range(3):
    out = fe(wave, crop_frames=T, start=0, norm_stats=(-0.8294, 4.6230))
torch.cuda.synchronize()
o = out[:, 0].cpu()                      # [B, 64, T]
# the kernel is persistent: workgroup i (of 3 x CUs) starts with entry i of the (16-frame group, clip) list -- group-major since round 5 -- and stamps that group's outputs
import ctypes
from ssl_audio_amd import ops
n_wg = 3 * torch.cuda.get_device_properties(0).multi_processor_count
gpc = (T + 15) // 16
names = ["prologue(tables)", "phase1(4 frames)", "weights", "barrier1", "mfma", "partials+barriers", "log+store", "whole workgroup"]
for w in range(4):
    rows = []
    for i in range(n_wg):
        b, c = divmod(i, B)
        if 16 * b + 8 <= T:
            rows.append(o[c, 16 * w, 16 * b: 16 * b + 8])
    st = torch.stack(rows)
    print("wave", w, " ".join(f"{n}={st[:, i].mean():.0f}" for i, n in enumerate(names)), f"({len(rows)} workgroups, {B * gpc / n_wg:.1f} groups each)")
