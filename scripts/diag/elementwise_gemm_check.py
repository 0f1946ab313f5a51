"""ELEMENTWISE check of the tiled GEMM kernels' compact epilogues at step-sized shapes (norm-relative tests can hide a handful of
corrupted elements): every output element against an fp64 reference, bound = bf16 / fp32 rounding of that element + accumulation noise.
Motivation: the store-data hazard found in gemm_wreg.hip (scripts/diag/scan_store_hazard.py lists
the same instruction pattern in the tiled kernels' ISA).   python scripts/diag/elementwise_gemm_check.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssl_audio_amd import ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator(device=dev).manual_seed(0)
worst = 0
for (M, N, K) in [(63744, 768, 768), (63744, 768, 3072), (63744, 3072, 768), (63744, 2304, 768), (127488, 192, 768), (127488, 768, 192)]:
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g)
    aux = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    acc = (A.double() @ W.double().t())                       # fp64 on the GPU (stock matmul): the reference
    scale = float(acc.abs().max())
    for tag, kw, ref, is32 in [("bias->bf16", dict(bias=bias), acc + bias.double(), False),
                               ("bias+res->f32", dict(bias=bias, residual=res), acc + bias.double() + res.double(), True),
                               ("x aux + colsum", dict(act=4, aux_in=aux), acc * aux.double(), False),
                               ("gelu pair", dict(bias=bias, act=3), torch.nn.functional.gelu(acc + bias.double()), False)]:
        nbad_tot = 0
        for r in range(reps):
            if is32:
                out = torch.empty(M, N, device=dev); ops.gemm(A, W, out_f32=out, **kw)
            else:
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); kw2 = dict(kw)
                if tag == "gelu pair": kw2["aux_out"] = torch.empty_like(out)
                if tag.startswith("x aux"): kw2["colsum_out"] = torch.zeros(N, device=dev)
                ops.gemm(A, W, out_bf16=out, **kw2)
            torch.cuda.synchronize()
            d = (out.double() - ref).abs()
            bound = (2.0 ** -8 if not is32 else 2.0 ** -20) * ref.abs() + 2e-5 * scale * (K / 768) ** 0.5 + (2e-3 if tag == "gelu pair" else 0)
            nbad = int((d > bound).sum())
            nbad_tot += nbad
            if nbad:
                idx = (d > bound).nonzero()[:5].tolist()
                print("   BAD", tag, (M, N, K), nbad, idx, [float(out[i, j]) for i, j in idx], [float(ref[i, j]) for i, j in idx])
        worst += nbad_tot
        print(f"M={M} N={N} K={K} {tag:16s} elements out of bound over {reps} runs: {nbad_tot}")
    del acc, res, aux
print("TOTAL out of bound:", worst)
