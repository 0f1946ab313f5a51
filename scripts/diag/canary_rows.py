"""Do the compact epilogues of the tiled GEMM kernels write rows past M?  (Their bf16 stores carry the row block in the SCALAR offset of a
buffer store, which is outside the resource's bounds check.)  Output canvases with 512 guard rows behind the M rows handed to sa_gemm_bf16.
   python scripts/diag/canary_rows.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssl_audio_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for (M, N, K) in [(33000, 768, 256), (33000, 768, 1536), (33000, 2304, 768), (4133, 768, 192), (33000 + 77, 3072, 768)]:
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g)
    aux = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    def canvas(dt):
        t = torch.full((M + 512, N), 777.0, device=dev, dtype=dt)
        return t, t[:M]
    for tag, kw, dt in [("bias->bf16", dict(bias=bias), torch.bfloat16), ("gelu pair", dict(bias=bias, act=3), torch.bfloat16), ("gelu", dict(bias=bias, act=1), torch.bfloat16),
                        ("bias+res->f32", dict(bias=bias, residual=res), torch.float32), ("x aux + colsum", dict(act=4, aux_in=aux), torch.bfloat16)]:
        full, out = canvas(dt)
        kw = dict(kw)
        extra = None
        if tag == "gelu pair":
            extra, ao = canvas(torch.bfloat16); kw["aux_out"] = ao
        if tag.startswith("x aux"):
            kw["colsum_out"] = torch.zeros(N, device=dev)
        if dt == torch.float32: ops.gemm(A, W, out_f32=out, **kw)
        else: ops.gemm(A, W, out_bf16=out, **kw)
        torch.cuda.synchronize()
        touched = int((full[M:] != 777.0).any(dim=1).sum()) + (int((extra[M:] != 777.0).any(dim=1).sum()) if extra is not None else 0)
        bad += touched
        print(f"M={M} N={N} K={K} {tag:16s} guard rows written: {touched}")
print("TOTAL guard rows written:", bad)
