"""GEMM micro-benchmark at the ViT-B step's shapes (M = 256 sequences x 249 tokens).  Random operands (rule 25)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops
dev = torch.device("cuda:0")
M = 63744
d = 768
g = torch.Generator(device=dev).manual_seed(0)
def rb(*s): return torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
cases = []
for name, N, K in [("qkv", 3 * d, d), ("proj", d, d), ("fc1", 4 * d, d), ("fc2", d, 4 * d)]:
    cases.append((f"fwd  NT {name}", dict(A=rb(M, K), B=rb(N, K), a_kmajor=True, b_kmajor=True, out_bf16=torch.empty(M, N, device=dev, dtype=torch.bfloat16)), 2.0 * M * N * K))
    cases.append((f"dgrad NN {name}", dict(A=rb(M, N), B=rb(N, K), a_kmajor=True, b_kmajor=False, out_bf16=torch.empty(M, K, device=dev, dtype=torch.bfloat16)), 2.0 * M * N * K))
    out = torch.zeros(N, K, device=dev)
    cases.append((f"wgrad TN {name}", dict(A=rb(M, N), B=rb(M, K), a_kmajor=False, b_kmajor=False, out_f32=out, split_k=ops.pick_split_k(N, K, M)), 2.0 * M * N * K))
    cases.append((f"wgrad256 {name}", dict(A=cases[-1][1]["A"], B=cases[-1][1]["B"], a_kmajor=False, b_kmajor=False, out_f32=out,
                                          split_k=ops.pick_split_k(N, K, M, tile=256), tile256=True), 2.0 * M * N * K))
def run(kw):
    kw = dict(kw); A = kw.pop("A"); B = kw.pop("B")
    ops.gemm(A, B, **kw)
only = sys.argv[1] if len(sys.argv) > 1 else None
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tot_t = tot_f = 0.0
for name, kw, fl in cases:
    if only and only not in name:
        continue
    for _ in range(3): run(kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = reps
    for _ in range(n): run(kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tot_t += ms; tot_f += fl
    print(f"{name:18s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s" + (f"  split_k={kw.get('split_k')}" if 'split_k' in kw else ""))
print(f"ALL (one layer fwd+bwd GEMMs): {tot_t:.3f} ms  {tot_f/tot_t/1e9:.1f} TFLOP/s")
