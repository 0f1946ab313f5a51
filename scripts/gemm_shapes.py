"""Which GEMM shapes does one training step launch, on which kernel family, and how long does each take?
usage: python scripts/gemm_shapes.py [workload]   (one GPU; prints a table sorted by total time)"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops, hyperparameters as hp
from ssl_audio_amd.train import BarlowTwinsTrainer

dev = torch.device("cuda:0")
size = sys.argv[1] if len(sys.argv) > 1 else "base"
B, T = (256 if size == "tiny" else 128), 1001
cfg = hp.make_args(model_type="vit_" + size, batch_size=B, crop_frames=T, dataset="audioset")
tr = BarlowTwinsTrainer(cfg, dev, mode="bt", batch_per_rank=B, clip_samples=160000, seed=0, from_waveform=False)
g = torch.Generator().manual_seed(0)
views = [torch.randn(B, 1, 64, T, generator=g).to(dev) for _ in range(2)]
for _ in range(2):
    tr.step_views(views)
torch.cuda.synchronize()
log = []
orig = ops.gemm
def logged(A, Bm, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(A, Bm, **kw); e1.record()
    ak, bk = kw.get("a_kmajor", True), kw.get("b_kmajor", True)
    M, K = (A.shape if ak else A.shape[::-1])
    N = Bm.shape[0] if bk else Bm.shape[1]
    epi = "+".join(k for k in ("bias", "residual", "aux_in", "aux_out", "colsum_out", "accumulate") if kw.get(k) is not None and kw.get(k) is not False)
    log.append((e0, e1, (M, N, K, ("N" if ak else "T") + ("T" if bk else "N"), kw.get("split_k", 1), bool(kw.get("tile256")), kw.get("act", 0), epi,
                         "f32" if kw.get("out_f32") is not None else "bf16")))
ops.gemm = logged
import ssl_audio_amd.engine as eng, ssl_audio_amd.functional as fn
tr.step_views(views)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for e0, e1, key in log:
    t = e0.elapsed_time(e1) * 1e3
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += t
tot = sum(v[1] for v in agg.values())
print(f"{len(log)} launches, {tot/1e3:.2f} ms")
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, lay, split, t256, act, epi, out = key
    fam = ops.gemm_kernel_family(M, N, K, lay[0] == "N", lay[1] == "T", split, t256, out == "f32" and "residual" in epi and act == 0, out == "bf16" and act == 0 and "aux" not in epi)
    print(f"{lay} M={M:6d} N={N:5d} K={K:6d} split={split:2d}{'*' if t256 else ' '} act={act} {epi:28s} {out:4s} x{n:3d}  {t/n:8.1f} us  {2*M*N*K/(t/n)/1e6:7.1f} TF/s  {100*t/tot:5.1f}%  {fam}")
