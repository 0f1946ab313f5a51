"""Helper for test_kernels_gpu.py::test_gemm_tile_modes: run in a subprocess with SA_GEMM_TILE / SA_GEMM_WGRAD_RING set
(the library reads them once), checks every operand layout of a large ragged problem against an fp64 CPU product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops


def bf(x):
    return x.to(torch.bfloat16)


def main():
    dev = "cuda"
    g = torch.Generator().manual_seed(5)
    M, N, K = 4000, 2104, 192 + 64          # 16 x 9 = 144 tiles of 256 x 256 -> the large-problem kernels; ragged M and N
    A = bf(torch.randn(M, K, generator=g)); B = bf(torch.randn(N, K, generator=g) * 0.1)
    bias = torch.randn(N, generator=g); res = torch.randn(M, N, generator=g)
    acc = A.double() @ B.double().T
    worst = 0.0
    for a_km in (True, False):
        for b_km in (True, False):
            Ad = (A if a_km else A.T.contiguous()).to(dev)
            Bd = (B if b_km else B.T.contiguous()).to(dev)
            out32 = torch.full((M, N), float("nan"), device=dev)
            out16 = torch.empty(M, N + 4, device=dev, dtype=torch.bfloat16)[:, :N]      # strided output rows
            ops.gemm(Ad, Bd, a_kmajor=a_km, b_kmajor=b_km, bias=bias.to(dev), residual=res.to(dev), out_f32=out32, out_bf16=out16)
            ref = acc + bias.double() + res.double()
            e32 = float((out32.cpu().double() - ref).norm() / ref.norm())
            e16 = float((out16.cpu().double() - ref).norm() / ref.norm())
            assert e32 < 1e-5 and e16 < 4e-3, (a_km, b_km, e32, e16)
            worst = max(worst, e32)
    # same shapes with N a multiple of 64 but not of 256 (the last column tile is partly outside N): compact epilogues must skip it
    N64 = 2112
    B64 = bf(torch.randn(N64, K, generator=g) * 0.1); bias64 = torch.randn(N64, generator=g); res64 = torch.randn(M, N64, generator=g)
    acc64 = A.double() @ B64.double().T
    for b_km in (True, False):
        Bd = (B64 if b_km else B64.T.contiguous()).to(dev)
        guard = torch.full((M + 2, N64), 7.0, device=dev, dtype=torch.bfloat16)                   # rows after the output must stay untouched
        ops.gemm(A.to(dev), Bd, b_kmajor=b_km, bias=bias64.to(dev), out_bf16=guard[:M])
        r1 = acc64 + bias64.double()
        assert float((guard[:M].cpu().double() - r1).norm() / r1.norm()) < 4e-3 and float((guard[M:].float() - 7.0).abs().max()) == 0.0
        o32 = torch.full((M + 2, N64), 7.0, device=dev)
        ops.gemm(A.to(dev), Bd, b_kmajor=b_km, bias=bias64.to(dev), residual=res64.to(dev), out_f32=o32[:M])
        r3 = acc64 + bias64.double() + res64.double()
        assert float((o32[:M].cpu().double() - r3).norm() / r3.norm()) < 1e-5 and float((o32[M:] - 7.0).abs().max()) == 0.0
    # the compact epilogues of the large kernels: bias -> bf16 only (qkv forward / plain dgrad), bias + residual -> fp32 only (proj / fc2)
    for a_km, b_km in ((True, True), (True, False)):
        Bd = (B if b_km else B.T.contiguous()).to(dev)
        o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(A.to(dev), Bd, b_kmajor=b_km, bias=bias.to(dev), out_bf16=o16)
        r1 = acc + bias.double()
        assert float((o16.cpu().double() - r1).norm() / r1.norm()) < 4e-3
        o32 = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(A.to(dev), Bd, b_kmajor=b_km, bias=bias.to(dev), residual=res.to(dev), out_f32=o32)
        r3 = acc + bias.double() + res.double()
        assert float((o32.cpu().double() - r3).norm() / r3.norm()) < 1e-5
        o16b = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(A.to(dev), Bd, b_kmajor=b_km, out_bf16=o16b, alpha=0.5)
        assert float((o16b.cpu().double() - 0.5 * acc).norm() / (0.5 * acc).norm()) < 4e-3
    # GELU epilogue with the pre-activation side output (forward fc1) and its derivative (dgrad through fc2)
    pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16); h = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(A.to(dev), B.to(dev), bias=bias.to(dev), act=1, aux_out=pre, out_bf16=h)
    x = acc + bias.double()
    assert float((pre.cpu().double() - x).norm() / x.norm()) < 4e-3
    gx = torch.nn.functional.gelu(x)
    assert float((h.cpu().double() - gx).norm() / gx.norm()) < 4e-3
    d = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(A.to(dev), B.to(dev), act=2, aux_in=pre, out_bf16=d)
    xx = pre.cpu().double().requires_grad_(True)
    torch.nn.functional.gelu(xx).sum().backward()
    refd = acc * xx.grad
    assert float((d.cpu().double() - refd).norm() / refd.norm()) < 4e-3
    # act 3 (GELU, aux = GELU') and act 4 (multiply by aux) on the large kernels, with the fused column sums on a 64-aligned width
    dgl = torch.empty(M, N64, device=dev, dtype=torch.bfloat16); h3 = torch.empty(M, N64, device=dev, dtype=torch.bfloat16)
    ops.gemm(A.to(dev), B64.to(dev), bias=bias64.to(dev), act=3, aux_out=dgl, out_bf16=h3)
    x64 = (acc64 + bias64.double()).requires_grad_(True)
    gx64 = torch.nn.functional.gelu(x64)
    gx64.sum().backward()
    assert float((h3.cpu().double() - gx64.detach()).norm() / gx64.detach().norm()) < 4e-3
    assert float((dgl.cpu().double() - x64.grad).norm() / x64.grad.norm()) < 4e-3
    d4 = torch.empty(M, N64, device=dev, dtype=torch.bfloat16); cs = torch.zeros(N64, device=dev)
    ops.gemm(A.to(dev), B64.T.contiguous().to(dev), b_kmajor=False, act=4, aux_in=dgl, out_bf16=d4, colsum_out=cs)
    ref4 = acc64 * dgl.cpu().double()
    assert float((d4.cpu().double() - ref4).norm() / ref4.norm()) < 4e-3
    assert float((cs.cpu().double() - ref4.sum(0)).norm() / ref4.sum(0).norm()) < 1e-4
    # split-K wgrad on the 256 tile (TN), ragged reduction
    T, Nw, Kw = 9000 + 17, 704, 1032
    dY = bf(torch.randn(T, Nw, generator=g)); X = bf(torch.randn(T, Kw, generator=g))
    out = torch.zeros(Nw, Kw, device=dev)
    ops.gemm(dY.to(dev), X.to(dev), a_kmajor=False, b_kmajor=False, out_f32=out, split_k=ops.pick_split_k(Nw, Kw, T, tile=256), tile256=True)
    refw = dY.double().T @ X.double()
    ew = float((out.cpu().double() - refw).norm() / refw.norm())
    assert ew < 1e-5, ew
    # dynamic tile hand-out (sa_set_dynamic_tiles): more tiles than CUs and >= 6 K-tiles per tile, so that the persistent AND the phased
    # kernel draw tickets; every epilogue family; bit-identical to static striding (the order tiles are drawn in cannot matter) and
    # right against fp64.  Ragged M: the last row panel is partly outside.
    Md, Nd, Kd = 8300, 2304, 768                        # 33 x 9 = 297 tiles of 256 x 256, 12 K-tiles each
    Ad = bf(torch.randn(Md, Kd, generator=g)); Bd_ = bf(torch.randn(Nd, Kd, generator=g) * 0.05)
    biasd = torch.randn(Nd, generator=g).to(dev); resd = torch.randn(Md, Nd, generator=g).to(dev)
    auxd = bf(torch.randn(Md, Nd, generator=g)).to(dev)
    accd = Ad.double() @ Bd_.double().T
    Ag, Bg = Ad.to(dev), Bd_.to(dev)
    runs = {}
    for dyn in (False, True):
        ops.set_dynamic_tiles(dyn)
        o1 = torch.empty(Md, Nd, device=dev, dtype=torch.bfloat16)
        ops.gemm(Ag, Bg, bias=biasd, out_bf16=o1)                                                   # kind 1
        o3 = torch.empty(Md, Nd, device=dev)
        ops.gemm(Ag, Bg, bias=biasd, residual=resd, out_f32=o3)                                     # kind 3 (phased)
        o6, a6 = torch.empty(Md, Nd, device=dev, dtype=torch.bfloat16), torch.empty(Md, Nd, device=dev, dtype=torch.bfloat16)
        ops.gemm(Ag, Bg, bias=biasd, act=3, aux_out=a6, out_bf16=o6)                                # kind 6
        o5 = torch.empty(Md, Nd, device=dev, dtype=torch.bfloat16); cs5 = torch.zeros(Nd, device=dev)
        ops.gemm(Ag, Bg, act=4, aux_in=auxd, out_bf16=o5, colsum_out=cs5)                           # kind 5 + fused column sums
        Bl = bf(torch.randn(Nd // 3, 4 * Kd, generator=torch.Generator().manual_seed(9)) * 0.03).to(dev)
        Al = bf(torch.randn(Md, 4 * Kd, generator=torch.Generator().manual_seed(10))).to(dev)
        ol = torch.empty(Md, Nd // 3, device=dev, dtype=torch.bfloat16)
        ops.gemm(Al, Bl, out_bf16=ol)                                                               # kind 1, K = 3072 (phased), 99 tiles: static
        torch.cuda.synchronize()
        runs[dyn] = (o1, o3, o6, a6, o5, cs5, ol)
    ops.set_dynamic_tiles(True)
    for x, y in zip(runs[False][:5], runs[True][:5]):
        assert torch.equal(x, y), "dynamic tile hand-out changed a result"
    assert torch.equal(runs[False][6], runs[True][6])
    assert float((runs[True][5] - runs[False][5]).abs().max()) <= 1e-3 * float(runs[False][5].abs().max())      # (column partials meet in fixed slots: equal up to nothing)
    r1 = accd + biasd.cpu().double()
    assert float((runs[True][0].cpu().double() - r1).norm() / r1.norm()) < 4e-3
    r3 = r1 + resd.cpu().double()
    assert float((runs[True][1].cpu().double() - r3).norm() / r3.norm()) < 1e-5
    g6 = torch.nn.functional.gelu(r1)
    assert float((runs[True][2].cpu().double() - g6).norm() / g6.norm()) < 4e-3
    r5 = accd * auxd.cpu().double()
    assert float((runs[True][4].cpu().double() - r5).norm() / r5.norm()) < 4e-3
    assert float((runs[True][5].cpu().double() - r5.sum(0)).norm() / r5.sum(0).norm()) < 1e-4
    # split-K weight gradient under dynamic hand-out with a RAGGED last K slice: 100 K-tiles in 13 slices of 8 leave a last
    # slice of 4, shorter than the phased kernel's ticket look-ahead, so that kernel must fall back to static striding for this launch;
    # 4 x 5 tiles x 13 slices = 260 units > 256 workgroups.  Atomic accumulation (no workspace) is the form SA_GEMM_WGRAD_PHASE=1 /
    # SA_GEMM_WGRAD_RING=1 take; dynamic and static must agree to fp32 summation-order noise and be right against fp64.
    Ts, Ns, Ks = 6400, 1024, 1280
    dYs = bf(torch.randn(Ts, Ns, generator=g)).to(dev); Xs = bf(torch.randn(Ts, Ks, generator=g)).to(dev)
    refs = dYs.cpu().double().T @ Xs.cpu().double()
    det = ops.DETERMINISTIC_WGRAD
    for deterministic in (True, False):
        ops.DETERMINISTIC_WGRAD = deterministic
        outs = []
        for dyn in (False, True):
            ops.set_dynamic_tiles(dyn)
            o = torch.zeros(Ns, Ks, device=dev)
            ops.gemm(dYs, Xs, a_kmajor=False, b_kmajor=False, out_f32=o, split_k=13, tile256=True)
            torch.cuda.synchronize()
            assert float((o.cpu().double() - refs).norm() / refs.norm()) < 1e-5, (deterministic, dyn)
            outs.append(o)
        if deterministic:
            assert torch.equal(outs[0], outs[1]), "dynamic hand-out changed a slice-ordered weight gradient"
        else:
            assert float((outs[0] - outs[1]).abs().max()) <= 1e-4 * float(outs[0].abs().max())
    ops.DETERMINISTIC_WGRAD = det
    torch.cuda.synchronize()
    print(f"ok tile={os.environ.get('SA_GEMM_TILE', 'default')} ring_wgrad={os.environ.get('SA_GEMM_WGRAD_RING', '0')} worst={worst:.2e} wgrad={ew:.2e}")


if __name__ == "__main__":
    main()
