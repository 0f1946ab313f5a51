"""Diagnostic: does the device-side non-finite gate keep AdamW from touching the weights?  (python scripts/diag_nan_gate.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssl_audio_amd import ops, hyperparameters as hp
from ssl_audio_amd.train import BarlowTwinsTrainer
dev = torch.device("cuda:0")
n = 4096
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
hyper = torch.tensor([1e-3, 10.0, 31.6, 0.0], device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
p0 = p.clone()
ops.adamw_step_dev(p, g, m, v, hyper, 0.9, 0.999, 1e-8, 0.05, skip_flag=flag)
print("flag 0: changed", bool((p != p0).any()))
p1 = p.clone(); flag.fill_(1)
ops.adamw_step_dev(p, g, m, v, hyper, 0.9, 0.999, 1e-8, 0.05, skip_flag=flag)
print("flag 1: changed", bool((p != p1).any()))
x = torch.tensor([float("nan")], device=dev); f2 = torch.zeros(1, dtype=torch.int32, device=dev)
ops.count_nonfinite(x, f2); print("count_nonfinite(nan) ->", int(f2))
cfg = hp.make_args(model_type="vit_tiny", batch_size=8, crop_frames=96, projector_hidden_dim=512, projector_out_dim=128)
tr = BarlowTwinsTrainer(cfg, dev, mode="bt", batch_per_rank=8, clip_samples=15200, seed=0, from_waveform=False)
gen = torch.Generator().manual_seed(1)
v1, v2 = torch.randn(8, 1, 64, 96, generator=gen).to(dev), torch.randn(8, 1, 64, 96, generator=gen).to(dev)
print("loss", float(tr.step_views([v1, v2])))
before = tr.flat.params.clone()
l = tr.step_views([torch.full_like(v1, float("nan")), v2])
torch.cuda.synchronize()
print("nan-step loss", float(l), "flag", int(tr._nonfinite), "params changed", bool((tr.flat.params != before).any()), "nan in params", bool(torch.isnan(tr.flat.params).any()))
bad = torch.isnan(tr.flat.params).nonzero().flatten()
if bad.numel():
    lo, hi = int(bad.min()), int(bad.max())
    names = [k for k, (off, cnt) in tr.flat.offsets.items() if off <= hi and off + cnt > lo]
    print("nan range", lo, hi, "n_decay", tr.flat.n_decay, "n_train", tr.flat.n_train, names[:6])
