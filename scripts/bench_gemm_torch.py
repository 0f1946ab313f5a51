"""The bar SURVEY.md §2.3 names: hipBLASLt (through torch.matmul, bf16) on the same 12 products of one ViT-B block's forward + backward
(M = 63 744 rows), plain products without the fused epilogues, random operands.  Scripts only: the product never calls torch.matmul."""
import sys, os
import torch
dev = torch.device("cuda:0")
M, d = 63744, 768
g = torch.Generator(device=dev).manual_seed(0)
def rb(*s): return torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
rows = []
for name, N, K in [("qkv", 3 * d, d), ("proj", d, d), ("fc1", 4 * d, d), ("fc2", d, 4 * d)]:
    X, W, dY = rb(M, K), rb(N, K), rb(M, N)
    rows.append((f"fwd   {name}  X W^T", lambda X=X, W=W: torch.matmul(X, W.t()), 2.0 * M * N * K))
    rows.append((f"dgrad {name}  dY W", lambda dY=dY, W=W: torch.matmul(dY, W), 2.0 * M * N * K))
    rows.append((f"wgrad {name}  dY^T X", lambda dY=dY, X=X: torch.matmul(dY.t(), X), 2.0 * M * N * K))
tot_t = tot_f = 0.0
for name, fn, fl in rows:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tot_t += ms; tot_f += fl
    print(f"{name:24s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")
print(f"ALL hipBLASLt (torch.matmul bf16, plain products): {tot_t:.3f} ms  {tot_f/tot_t/1e9:.1f} TFLOP/s")
