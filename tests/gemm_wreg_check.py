"""Parity of the weights-in-registers streaming kernel (csrc/gemm_wreg.hip), run by tests/test_kernels_gpu.py::test_gemm_weights_in_registers in a
subprocess with SA_GEMM_WREG=2 (the library reads the variable once; 2 = a K = 192 forward-layout launch the kernel does not cover is an
error).  fp64 references; bf16 outputs within one rounding (2^-8 relative per element + accumulation noise), the fp32 residual output to
1e-5; a one-hot product pins the column-to-fragment map exactly; rows past M of the output allocations stay untouched (the kernel clips
rows through buffer bounds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops

assert os.environ.get("SA_GEMM_WREG") == "2"
dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale).to(dtype)


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(M, N):
    K = 192
    A = bf(rnd((M, K), 71)); W = bf(rnd((N, K), 72, 0.1)); bias = rnd((N,), 73)
    acc = A.double() @ W.double().t()
    pad = 96

    def canvas(dtype):
        t = torch.full((M + pad, N), 777.0, device=dev, dtype=dtype)
        return t, t[:M]

    def check_bf16(out, ref, tag):
        d = (out.double().cpu() - ref).abs()
        assert rel_err(out, ref) < 4e-3, tag
        assert float((d - 2.0 ** -8 * ref.abs()).max()) < 1e-3 * float(ref.abs().max()), tag

    # kind 1: bias -> bf16, alpha
    full, out = canvas(BF16)
    ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), alpha=0.5, out_bf16=out)
    check_bf16(out, 0.5 * acc + bias.double(), "bias -> bf16")
    assert bool((full[M:] == 777.0).all())
    # one-hot operands: every output element is one exact product
    A1 = torch.zeros(M, K); W1 = torch.zeros(N, K)
    rows = torch.arange(M); A1[rows, rows % K] = 1.0
    cols = torch.arange(N); W1[cols, (cols * 7) % K] = (cols % 13 + 1).float()
    ops.gemm(bf(A1).to(dev), bf(W1).to(dev), out_bf16=out)
    assert torch.equal(out.float().cpu(), (A1.double() @ W1.double().t()).float())
    if N == 192:
        # kind 3: bias + residual -> fp32
        res = rnd((M, N), 74)
        full32, out32 = canvas(torch.float32)
        ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), residual=res.to(dev), out_f32=out32)
        assert rel_err(out32, acc + bias.double() + res.double()) < 1e-5
        assert bool((full32[M:] == 777.0).all())
    if N == 768:
        # kind 6: GELU pair (fc1 forward, act 3) and kind 8: GELU only
        h = (acc + bias.double()).requires_grad_(True)
        y = torch.nn.functional.gelu(h)
        y.sum().backward()
        fulld, dg = canvas(BF16)
        ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), act=3, aux_out=dg, out_bf16=out)
        check_bf16(out, y.detach(), "GELU")
        check_bf16(dg, h.grad, "GELU'")
        assert bool((full[M:] == 777.0).all()) and bool((fulld[M:] == 777.0).all())
        out8 = torch.empty_like(out)
        ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), act=1, out_bf16=out8)
        assert torch.equal(out8, out)
        # kind 5: x aux (fc2 data gradient, act 4) with the fused column sums (fc1's bias gradient)
        aux = bf(rnd((M, N), 75))
        cs = torch.full((N,), 2.0, device=dev)
        ops.gemm(A.to(dev), W.to(dev), act=4, aux_in=aux.to(dev), out_bf16=out, colsum_out=cs)
        ref5 = acc * aux.double()
        check_bf16(out, ref5, "x aux")
        assert rel_err(cs, 2.0 + ref5.sum(0)) < 1e-4
        assert bool((full[M:] == 777.0).all())
        ops.gemm(A.to(dev), W.to(dev), act=4, aux_in=aux.to(dev), out_bf16=out)          # (without the column sums)
        check_bf16(out, ref5, "x aux, no column sums")




for M in (4096 + 37, 20032):
    for N in (192, 576, 768):
        check(M, N)
        torch.cuda.synchronize()
print("ok wreg")
