"""Diagnostic: dump sa_layernorm_bwd outputs for fixed inputs (compare two builds of the library bit for bit, and against fp64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops
dev = "cuda"
torch.manual_seed(0)
M, D = 400, 192
x = torch.randn(M, D, device=dev); dy = torch.randn(M, D, device=dev).bfloat16(); dres = torch.randn(M, D, device=dev)
g = torch.randn(D, device=dev); mean = x.mean(1); rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt()
dx = torch.empty(M, D, device=dev); dx16 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev); dxs = torch.zeros(D, device=dev)
ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dx_f32=dx, dx_bf16=dx16, dgamma=dg, dbeta=db, dxsum=dxs)
xd, dyd, gd_, md, rd = x.double(), dy.double(), g.double(), mean.double(), rstd.double()
xh = (xd - md[:, None]) * rd[:, None]; gdy = dyd * gd_
ref = rd[:, None] * (gdy - gdy.mean(1, keepdim=True) - xh * (gdy * xh).mean(1, keepdim=True)) + dres.double()
print("max |dx - fp64|", float((dx.double() - ref).abs().max()), "ulp-ish", float(((dx.double() - ref).abs() / ref.abs().clamp_min(1e-3)).max()))
torch.save({"dx": dx.cpu(), "dx16": dx16.cpu(), "dg": dg.cpu(), "db": db.cpu(), "dxs": dxs.cpu()}, sys.argv[1])
if len(sys.argv) > 2:
    o = torch.load(sys.argv[2])
    for k, v in {"dx": dx, "dx16": dx16, "dg": dg, "db": db, "dxs": dxs}.items():
        a, b = v.cpu().float(), o[k].float()
        print(k, "elements differing", int((a != b).sum()), "of", a.numel(), "max abs diff", float((a - b).abs().max()))
