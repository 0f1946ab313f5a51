"""Where does the log-mel kernel differ from the oracle?  (diagnostic)  python scripts/diag/logmel_err.py [L] [T] [start]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssl_audio_amd import frontend as fe
from oracle import frontend as ofe
L, T, start = (int(x) for x in (sys.argv[1:4] + ["160000", "1001", "0"][len(sys.argv) - 1:]))
g = torch.Generator().manual_seed(29)
wave = 0.1 * torch.randn(3, L, generator=g)
out = fe.MelSpectrogram()(wave.cuda(), crop_frames=T, start=start, norm_stats=(-0.8294, 4.6230)).cpu().numpy()[:, 0]
ref = ofe.crop_pad_normalize(ofe.logmel(wave.numpy()), T, start, -0.8294, 4.6230)
d = np.abs(out - ref)
print("max", d.max(), "mean", d.mean())
bad = np.argwhere(d > 2e-3)
print(len(bad), "bad of", d.size)
if len(bad):
    print("clips", np.unique(bad[:, 0]), "bands", np.unique(bad[:, 1])[:70], "frames", np.unique(bad[:, 2])[:80])
