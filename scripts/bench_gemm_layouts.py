"""Same 4096^3 problem through the four operand layouts (isolates layout effects from shape effects)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for n in (4096, 8192):
    A = torch.randn(n, n, device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn(n, n, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    for akm, bkm in [(True, True), (True, False), (False, True), (False, False)]:
        for _ in range(2): ops.gemm(A, B, a_kmajor=akm, b_kmajor=bkm, out_bf16=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.gemm(A, B, a_kmajor=akm, b_kmajor=bkm, out_bf16=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"n={n} a_kmajor={akm!s:5} b_kmajor={bkm!s:5}  {ms*1e3:8.1f} us  {2*n**3/ms/1e9:7.1f} TFLOP/s")
