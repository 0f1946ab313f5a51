"""Sweep the split-K count of the four weight-gradient GEMMs of a block (256^2 tiles, deterministic slice reduce included) against the
count `ops.pick_split_k` chooses.   python scripts/sweep_splitk.py            (SA_BENCH_D / SA_BENCH_SEQS as scripts/bench_gemm.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops
dev = torch.device("cuda:0")
d = int(os.environ.get("SA_BENCH_D", "768"))
M = int(os.environ.get("SA_BENCH_SEQS", "256")) * 249
g = torch.Generator(device=dev).manual_seed(0)
rb = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
x16, h4, q16 = rb(M, d), rb(M, 4 * d), rb(M, 3 * d)


def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, N, K, dy, xx in [("qkv", 3 * d, d, q16, x16), ("proj", d, d, x16, x16), ("fc1", 4 * d, d, h4, x16), ("fc2", d, 4 * d, x16, h4)]:
    out = torch.zeros(N, K, device=dev)
    tile256 = ((N + 255) // 256) * ((K + 255) // 256) >= 9
    chosen = ops.pick_split_k(N, K, M, tile=256 if tile256 else 128)
    tiles = ((N + 255) // 256) * ((K + 255) // 256) if tile256 else ((N + 127) // 128) * ((K + 127) // 128)
    res = {}
    for sk in sorted(set(list(range(2, 41)) + [chosen])):
        try:
            res[sk] = timed(lambda: ops.gemm(dy, xx, a_kmajor=False, b_kmajor=False, out_f32=out, split_k=sk, tile256=tile256))
        except Exception as e:  # noqa: BLE001
            res[sk] = float("nan")
    best = min(res, key=lambda k: res[k] if res[k] == res[k] else 1e9)
    print(f"wgrad {name:4s} [{N} x {K}], {tiles} tiles of {'256' if tile256 else '128'}^2: chosen split {chosen} = {res[chosen]:.1f} us; best split {best} = {res[best]:.1f} us "
          f"({100 * (res[chosen] - res[best]) / res[chosen]:.1f} % faster); workgroups chosen / best: {tiles * chosen} / {tiles * best}")
    print("   " + "  ".join(f"{k}:{v:.0f}" for k, v in res.items()))
