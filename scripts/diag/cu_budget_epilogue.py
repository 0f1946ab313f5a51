import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ssl_audio_amd import ops
dev=torch.device("cuda:0")
d=768; M=256*249
g=torch.Generator(device=dev).manual_seed(0)
rb=lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
x16, h4 = rb(M,d), rb(M,4*d)
res=torch.randn(M,d,device=dev,generator=g)
Wp, W2 = rb(d,d)*0.05, rb(d,4*d)*0.05
bp=torch.randn(d,device=dev,generator=g)
o=torch.empty(M,d,device=dev); o16=torch.empty(M,d,device=dev,dtype=torch.bfloat16)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for cus in (256, 192, 128, 64):
    ops.set_cu_budget(cus if cus<256 else 0)
    a=t(lambda: ops.gemm(x16, Wp, bias=bp, residual=res, out_f32=o))
    b=t(lambda: ops.gemm(x16, Wp, bias=bp, out_bf16=o16))
    c=t(lambda: ops.gemm(h4, W2, bias=bp, residual=res, out_f32=o))
    e=t(lambda: ops.gemm(h4, W2, bias=bp, out_bf16=o16))
    print(f"CUs {cus}: proj fwd +res->f32 {a:.1f} us, ->bf16 {b:.1f} us | fc2 fwd +res->f32 {c:.1f} us, ->bf16 {e:.1f} us")
