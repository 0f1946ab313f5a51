"""Stand-alone timing of the LayerNorm kernels at the ViT-B bench shape (M = 128 * 251 rows, D = 768)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops

M, D = int(os.environ.get("SA_LN_M", 128 * 251)), int(os.environ.get("SA_LN_D", 768))
dev = "cuda"
x = torch.randn(M, D, device=dev)
dyb = torch.randn(M, D, device=dev).bfloat16()
dres = torch.randn(M, D, device=dev)
g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
y = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
dx = torch.empty(M, D, device=dev)
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev); dxs = torch.zeros(D, device=dev)


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, y_bf16=y, mean=mean, rstd=rstd))
print(f"ln_fwd            {t:7.1f} us  {(M*D*6)/t/1e6:6.2f} TB/s")
t = timeit(lambda: ops.layernorm_bwd(dyb, x, g, mean, rstd, dres=dres, dx_f32=dx, dgamma=dg, dbeta=db, dxsum=dxs))
print(f"ln_bwd full       {t:7.1f} us  {(M*D*14)/t/1e6:6.2f} TB/s")
t = timeit(lambda: ops.layernorm_bwd(dyb, x, g, mean, rstd, dres=dres, dx_f32=dx, dgamma=dg, dbeta=db))
print(f"ln_bwd no dxsum   {t:7.1f} us")
t = timeit(lambda: ops.layernorm_bwd(dyb, x, g, mean, rstd, dres=dres, dx_f32=dx))
print(f"ln_bwd no colsums {t:7.1f} us  {(M*D*14)/t/1e6:6.2f} TB/s")
t = timeit(lambda: dx.copy_(dres))
print(f"copy fp32 (ref)   {t:7.1f} us  {(M*D*8)/t/1e6:6.2f} TB/s")
