"""Diagnostic: is sa_layernorm_bwd bit-reproducible launch to launch (same inputs, fresh outputs)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssl_audio_amd import ops
dev = "cuda"
torch.manual_seed(0)
for (M, D) in [(400, 192), (16, 192), (63744, 768), (25 * 16, 192)]:
    x = torch.randn(M, D, device=dev); dy = torch.randn(M, D, device=dev).bfloat16(); dres = torch.randn(M, D, device=dev)
    g = torch.randn(D, device=dev); mean = x.mean(1); rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt()
    ref = None; bad = 0
    for it in range(100 if M < 10000 else 10):
        dx = torch.full((M, D), float("nan"), device=dev); dx16 = torch.zeros(M, D, device=dev, dtype=torch.bfloat16)
        dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev); dxs = torch.zeros(D, device=dev)
        ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dx_f32=dx, dx_bf16=dx16, dgamma=dg, dbeta=db, dxsum=dxs)
        cur = (dx.clone(), dx16.clone(), dg.clone(), db.clone(), dxs.clone())
        if ref is None: ref = cur
        else:
            for a, b in zip(ref, cur):
                if not torch.equal(a, b): bad += 1; break
    print(M, D, "mismatching launches:", bad, "nan in dx:", bool(torch.isnan(ref[0]).any()))
