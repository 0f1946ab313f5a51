"""Kernel-level parity on a real MI355X: every C-ABI entry point against a CPU restatement of the same op
(fp64/fp32 PyTorch or the oracle), on seeded inputs.  Tolerances are written next to each check."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ssl_audio_amd import ops  # noqa: E402
from ssl_audio_amd import frontend as fe  # noqa: E402

BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU box"
    ops.lib()  # fails loudly if the HIP library is missing
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale).to(dtype)


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("a_km,b_km", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (384, 192, 256), (200, 72, 64), (128, 128, 64 * 7)])
def test_gemm_layouts(dev, a_km, b_km, M, N, K):
    """All four operand layouts, ragged M/N tiles; bf16 inputs are exact, fp32 accumulation: rel err <= 1e-5."""
    A = bf(rnd((M, K), 1)); B = bf(rnd((N, K), 2))
    ref = A.double() @ B.double().T
    Ad = (A if a_km else A.T.contiguous()).to(dev)
    Bd = (B if b_km else B.T.contiguous()).to(dev)
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(Ad, Bd, a_kmajor=a_km, b_kmajor=b_km, out_f32=out)
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 1e-5
    # asymmetric check (a transposed C write would pass a symmetric problem): single hot element
    A2 = torch.zeros(M, K); A2[3, 5] = 1.0
    B2 = torch.zeros(N, K); B2[7, 5] = 2.0
    ops.gemm((bf(A2) if a_km else bf(A2).T.contiguous()).to(dev), (bf(B2) if b_km else bf(B2).T.contiguous()).to(dev),
             a_kmajor=a_km, b_kmajor=b_km, out_f32=out)
    torch.cuda.synchronize()
    o = out.cpu()
    assert o[3, 7] == 2.0 and o.abs().sum() == 2.0


def test_gemm_epilogues(dev):
    M, N, K = 300, 192, 128
    A = bf(rnd((M, K), 3)); W = bf(rnd((N, K), 4, 0.1)); bias = rnd((N,), 5); res = rnd((M, N), 6)
    acc = A.double() @ W.double().T
    # bias + GELU, pre-activation side output, bf16 + fp32 outputs
    out32 = torch.empty(M, N, device=dev); out16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), act=1, aux_out=pre, out_f32=out32, out_bf16=out16)
    h = acc + bias.double()
    ref = torch.nn.functional.gelu(h)
    assert rel_err(out32, ref) < 2e-5
    assert rel_err(out16, ref) < 4e-3          # bf16 rounding of the output
    assert rel_err(pre, h) < 4e-3
    # GELU' epilogue (backward through fc1 activation): dH = dA * gelu'(h)
    hb = bf(h.float())
    ops.gemm(A.to(dev), W.to(dev), act=2, aux_in=hb.to(dev), out_f32=out32)
    hh = hb.double().requires_grad_(True)
    torch.nn.functional.gelu(hh).sum().backward()
    assert rel_err(out32, acc * hh.grad) < 2e-5
    # act 3 / 4: the forward stores GELU'(h) instead of h, the backward multiplies by it (same results as act 1 / 2)
    dg = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), act=3, aux_out=dg, out_f32=out32)
    hh2 = h.clone().requires_grad_(True)
    torch.nn.functional.gelu(hh2).sum().backward()
    assert rel_err(out32, ref) < 2e-5 and rel_err(dg, hh2.grad) < 4e-3
    ops.gemm(A.to(dev), W.to(dev), act=4, aux_in=dg, out_f32=out32)
    assert rel_err(out32, acc * dg.cpu().double()) < 2e-5
    # bias + residual, alpha, accumulate
    ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), residual=res.to(dev), alpha=0.5, out_f32=out32)
    assert rel_err(out32, 0.5 * acc + bias.double() + res.double()) < 1e-5
    ops.gemm(A.to(dev), W.to(dev), out_f32=out32, accumulate=True)
    assert rel_err(out32, 1.5 * acc + bias.double() + res.double()) < 1e-5
    # patch-embed style: residual indexed modulo a period and output rows scattered past a CLS row
    per = 20
    pos = rnd((per, N), 7)
    tok = torch.zeros(M // per * (per + 1), N, device=dev)
    ops.gemm(A[:M // per * per].contiguous().to(dev), W.to(dev), bias=bias.to(dev), residual=pos.to(dev), res_mod=per, row_group=per, out_f32=tok)
    t = tok.cpu().view(M // per, per + 1, N)
    refp = (acc[:M // per * per] + bias.double()).view(M // per, per, N) + pos.double()
    assert rel_err(t[:, 1:], refp) < 1e-5 and float(t[:, 0].abs().max()) == 0.0


def test_gemm_split_k_wgrad(dev):
    """TN (wgrad) with a long ragged reduction and split-K atomics into a zeroed fp32 buffer."""
    Mtok, N, K = 1000, 192, 256
    dY = bf(rnd((Mtok, N), 8)); X = bf(rnd((Mtok, K), 9))
    ref = dY.double().T @ X.double()
    out = torch.zeros(N, K, device=dev)
    split = 4
    ops.gemm(dY.to(dev), X.to(dev), a_kmajor=False, b_kmajor=False, out_f32=out, split_k=split)
    assert rel_err(out, ref) < 1e-5
    assert ops.pick_split_k(768, 768, 63744) >= 8
    # the 256 x 256 split-K tile (one workgroup per CU) on ragged output and reduction sizes
    Mtok, N, K = 3000, 320, 520
    dY = bf(rnd((Mtok, N), 81)); X = bf(rnd((Mtok, K), 82))
    out = torch.zeros(N, K, device=dev)
    ops.gemm(dY.to(dev), X.to(dev), a_kmajor=False, b_kmajor=False, out_f32=out, split_k=5, tile256=True, alpha=0.5)
    assert rel_err(out, 0.5 * (dY.double().T @ X.double())) < 1e-5


@pytest.mark.parametrize("tile256", [False, True, 2])
def test_gemm_split_k_deterministic(dev, tile256, monkeypatch):
    """The workspace form of split-K (VERDICT r1 hygiene item: a deterministic wgrad reduction): slices are stored and added in slice
    order, so two runs are bit-identical, the result accumulates into out_f32 like the atomic form and equals it to fp32 rounding.
    Sizes: ragged M / N tiles, a K whose last slice is short and one whose trailing slice is empty (K = 9 steps over 5 slices)."""
    monkeypatch.setattr(ops, "DETERMINISTIC_WGRAD", True)
    # ... and many thin slices (>= 16: the four-waves-per-column-group form of the reduce, slice counts 17 / 42 leave ragged tails per wave)
    for Mtok, N, K, split in [(3000, 320, 520, 5), (9 * 64, 256, 384, 5), (63744 // 8, 768, 768, 7), (17 * 64 * 3, 192, 200, 17),
                              (42 * 128, 192, 768, 42)]:
        dY = bf(rnd((Mtok, N), 83)).to(dev); X = bf(rnd((Mtok, K), 84)).to(dev)
        base = torch.randn(N, K, device=dev)
        outs = []
        for _ in range(2):
            out = base.clone()
            ops.gemm(dY, X, a_kmajor=False, b_kmajor=False, out_f32=out, split_k=split, tile256=tile256, alpha=0.5)
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
        ref = base.double().cpu() + 0.5 * (dY.double().cpu().T @ X.double().cpu())
        assert rel_err(outs[0], ref) < 1e-5
        monkeypatch.setattr(ops, "DETERMINISTIC_WGRAD", False)
        out = base.clone()
        ops.gemm(dY, X, a_kmajor=False, b_kmajor=False, out_f32=out, split_k=split, tile256=tile256, alpha=0.5)
        monkeypatch.setattr(ops, "DETERMINISTIC_WGRAD", True)
        assert rel_err(out, outs[0].double().cpu()) < 1e-6


@pytest.mark.parametrize("N,K", [(576, 192), (192, 192), (768, 192), (192, 768), (200, 392), (64, 136), (1152, 384), (384, 1536)])
def test_gemm_streaming_narrow_wgrad(dev, N, K):
    """The 192 x 192 streaming split-K kernel (tile256 = 2; csrc/gemm_stream.hip) on ViT-T's four weight-gradient shapes (the products
    of models/mae.py:106-129,149-163's Linear layers at d = 192), two of the MAE decoder's (d = 384: 12 and 16 tiles) and on ragged outputs, against fp64 on the same bf16 operands: long
    slices (the three-stage ring wraps many times), one-K-step slices (only the prologue's requests are real), a short last slice, a
    reduction that is no multiple of 64 rows, and the split engine._wgrad picks at the full 127 488 rows."""
    for rows, split in [(127488 // 8, ops.pick_split_k(N, K, 127488 // 8, tile=192)), (64 * 7, 7), (64 * 9 + 17, 4), (64 * 40, 3)]:
        dY = bf(rnd((rows, N), 85)).to(dev); X = bf(rnd((rows, K), 86)).to(dev)
        base = torch.randn(N, K, device=dev)
        out = base.clone()
        ops.gemm(dY, X, a_kmajor=False, b_kmajor=False, out_f32=out, split_k=split, tile256=2, alpha=0.25)
        ref = base.double().cpu() + 0.25 * (dY.double().cpu().T @ X.double().cpu())
        assert rel_err(out, ref) < 1e-5, (rows, split)
        # ... and bit for bit what the 128 x 128 kernel's slices add up to when both cut the reduction at the same K-steps?  No: the
        # MFMA accumulation order inside a slice differs (one 64-row K-step at a time in both, but 32-deep halves in the same order),
        # so only fp32-rounding agreement is asserted
        o2 = base.clone()
        ops.gemm(dY, X, a_kmajor=False, b_kmajor=False, out_f32=o2, split_k=split, alpha=0.25)
        assert rel_err(out, o2.double().cpu()) < 1e-6


@pytest.mark.parametrize("det", [True, False])
def test_gemm_wgrad_bias_gradient_rides_along(dev, det, monkeypatch):
    """asum_out of the streaming split-K kernels: the q / v bias gradient of a packed qkv Linear (models/mae.py:125-128: column sums of dqkv,
    k's third skipped because its bias is fixed at zero) taken from the dY tiles of the weight-gradient launch -- 192 x 192 tiles (ViT-T),
    256 x 256 tiles (ViT-B), as a single product and inside a block's group; against fp64 column sums, the skipped rows untouched, the weight
    gradient itself unchanged by the extra fragment, the workspace form bit-reproducible; other kernels refuse the argument."""
    monkeypatch.setattr(ops, "DETERMINISTIC_WGRAD", det)
    for d, tile256, rows, split in [(192, 2, 64 * 83 + 24, 21), (768, 1, 64 * 40, 5)]:
        dqkv, h1 = bf(rnd((rows, 3 * d), 51)).to(dev), bf(rnd((rows, d), 52)).to(dev)
        ref = dqkv.double().cpu().sum(0)
        runs = []
        for _ in range(2):
            w, b = torch.zeros(3 * d, d, device=dev), torch.full((3 * d,), 0.25, device=dev)
            ops.gemm(dqkv, h1, a_kmajor=False, b_kmajor=False, out_f32=w, split_k=split, tile256=tile256, asum_out=b, asum_skip_lo=d, asum_skip_hi=2 * d)
            runs.append((w, b))
        w, b = runs[0]
        assert float((b[d:2 * d] - 0.25).abs().max()) == 0.0
        got = (b - 0.25).double().cpu()
        for lo, hi in [(0, d), (2 * d, 3 * d)]:
            assert float((got[lo:hi] - ref[lo:hi]).abs().max()) < 1e-4 * float(ref.abs().max()) + 1e-3
        w0 = torch.zeros_like(w)
        ops.gemm(dqkv, h1, a_kmajor=False, b_kmajor=False, out_f32=w0, split_k=split, tile256=tile256)
        assert torch.equal(w, w0) if det else rel_err(w, w0.double().cpu()) < 1e-6
        if det:
            assert torch.equal(runs[0][1], runs[1][1])
        # inside a group (the block's four products; the row sums belong to product 3)
        x2 = bf(rnd((rows, d), 53)).to(dev)
        outs = [torch.zeros(d, d, device=dev), torch.zeros(3 * d, d, device=dev)]
        bg = torch.zeros(3 * d, device=dev)
        ops.gemm_wgrad_group([x2, dqkv], [h1, h1], outs, split, tile=192 if tile256 == 2 else 256, asum_out=bg, asum_index=1, asum_skip_lo=d, asum_skip_hi=2 * d)
        assert rel_err(outs[1], w0.double().cpu()) < 1e-6 and float(bg[d:2 * d].abs().max()) == 0.0
        assert float((bg.double().cpu() - got).abs().max()) < 1e-4 * float(ref.abs().max()) + 1e-3
    with pytest.raises(RuntimeError, match="streaming split-K kernels only"):
        ops.gemm(dqkv, h1, a_kmajor=False, b_kmajor=False, out_f32=torch.zeros(3 * d, d, device=dev), split_k=4, asum_out=torch.zeros(3 * d, device=dev))


@pytest.mark.parametrize("det", [True, False])
def test_gemm_wgrad_group(dev, det, monkeypatch):
    """sa_gemm_wgrad_group: the four weight gradients of a ViT-T block (models/mae.py:106-129,149-163 at backward: qkv 576 x 192, proj
    192 x 192, fc1 768 x 192, fc2 192 x 768) over the same rows in ONE pair of launches, accumulated into non-zero buffers: against fp64
    on the same bf16 operands, against the four single launches, twice bit for bit (workspace form); also a group with strided operands
    (dq / dv column ranges of a packed dqkv), ragged outputs, a reduction that is no multiple of 64 and one-K-step slices; and the
    argument checks (mixed reductions, a k-major operand, nine products)."""
    monkeypatch.setattr(ops, "DETERMINISTIC_WGRAD", det)
    d = 192
    for rows, split in [(127488 // 8, 21), (64 * 21, 21), (64 * 9 + 40, 5)]:
        dqkv, h1 = bf(rnd((rows, 3 * d), 91)).to(dev), bf(rnd((rows, d), 92)).to(dev)
        dx2, ao = bf(rnd((rows, d), 93)).to(dev), bf(rnd((rows, d), 94)).to(dev)
        dpre, h2 = bf(rnd((rows, 4 * d), 95)).to(dev), bf(rnd((rows, d), 96)).to(dev)
        dx3, a = bf(rnd((rows, d), 97)).to(dev), bf(rnd((rows, 4 * d), 98)).to(dev)
        jobs = [(dx3, a), (dpre, h2), (dx2, ao), (dqkv, h1), (dqkv[:, 2 * d:], ao[:, :136])]      # (the last: strided dY, ragged 192 x 136 output)
        base = [torch.randn(dy.shape[1], x.shape[1], device=dev) for dy, x in jobs]
        runs = []
        for _ in range(2):
            outs = [b.clone() for b in base]
            ops.gemm_wgrad_group([j[0] for j in jobs], [j[1] for j in jobs], outs, split)
            runs.append(outs)
        for (dy, x), b, o, o2 in zip(jobs, base, runs[0], runs[1]):
            ref = b.double().cpu() + dy.double().cpu().T @ x.double().cpu()
            assert rel_err(o, ref) < 1e-5, (rows, split, tuple(o.shape))
            if det:
                assert torch.equal(o, o2)
            single = b.clone()
            ops.gemm(dy, x, a_kmajor=False, b_kmajor=False, out_f32=single, split_k=split, tile256=2)
            assert rel_err(o, single.double().cpu()) < 1e-6
    # the same on the 256 x 256 ring (wide outputs: a ViT-B block's shapes at a short reduction, plus a ragged product)
    d, rows, split = 768, 64 * 23 + 8, 5
    ops_in = [(bf(rnd((rows, n), 70 + i)).to(dev), bf(rnd((rows, k), 80 + i)).to(dev)) for i, (n, k) in enumerate([(d, 4 * d), (4 * d, d), (d, d), (3 * d, d), (520, 304)])]
    base = [torch.randn(dy.shape[1], x.shape[1], device=dev) for dy, x in ops_in]
    outs = [b_.clone() for b_ in base]
    ops.gemm_wgrad_group([j[0] for j in ops_in], [j[1] for j in ops_in], outs, split, tile=256)
    for (dy, x), b_, o in zip(ops_in, base, outs):
        assert rel_err(o, b_.double().cpu() + dy.double().cpu().T @ x.double().cpu()) < 1e-5, tuple(o.shape)
        single = b_.clone()
        ops.gemm(dy, x, a_kmajor=False, b_kmajor=False, out_f32=single, split_k=split, tile256=True)
        assert rel_err(o, single.double().cpu()) < 1e-6
    d = 192
    o = [torch.zeros(d, d, device=dev), torch.zeros(d, d, device=dev)]
    with pytest.raises((RuntimeError, ValueError), match="same in the whole group|do not form"):
        ops.gemm_wgrad_group([dx2, dx2[:320]], [ao, ao[:320]], o, 4)
    with pytest.raises(RuntimeError, match="1 .. 8 products"):
        ops.gemm_wgrad_group([dx2] * 9, [ao] * 9, [o[0]] * 9, 4)


@pytest.mark.parametrize("env", [{}, {"SA_GEMM_TILE": "6"}, {"SA_GEMM_TILE": "8", "SA_GEMM_WGRAD_RING": "1"}, {"SA_GEMM_TILE": "1"},
                                 {"SA_GEMM_TILE": "A", "SA_GEMM_WGRAD_PHASE": "1"}, {"SA_GEMM_TILE": "P"},
                                 {"SA_GEMM_WGRAD_STREAM256": "0"}])
def test_gemm_tile_modes(dev, env):
    """Large ragged problem through every tile variant of sa_gemm_bf16 (default dispatch first).  The variant is chosen by an
    environment variable the library reads once, hence one subprocess per variant (sequential: one GPU process at a time)."""
    import os, subprocess, sys
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "gemm_mode_check.py")], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok tile=" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_gemm_gelu_only_compact_epilogue(dev):
    """fc1 forward WITHOUT a backward to come (act = 1, no aux: the EMA target network of main_bt_byol.py:97-101, encode_vit, the HEAR
    wrappers): on the persistent 256 x 256 kernel it takes the compact GELU-only epilogue (kind 8; until round 5 the general one).
    Exact-erf GELU against torch (bf16 output), ragged last row tile, and equal to the GELU half of the act = 3 launch bit for bit."""
    M, N, K = 33000, 768, 256
    A = bf(rnd((M, K), 61)); W = bf(rnd((N, K), 62, 0.1)); bias = rnd((N,), 63)
    ref = torch.nn.functional.gelu(A.double() @ W.double().t() + bias.double())
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), act=1, out_bf16=out)
    assert rel_err(out, ref) < 4e-3
    assert float((out.double().cpu() - ref).abs().max()) < 2.0 ** -7 * float(ref.abs().max())
    out3, dg = torch.empty_like(out), torch.empty_like(out)
    ops.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), act=3, aux_out=dg, out_bf16=out3)
    assert torch.equal(out, out3)


def test_gemm_weights_in_registers(dev):
    """The opt-in weights-in-registers streaming kernel (gemm_wreg.hip, SA_GEMM_WREG: forward layout, K = 192, N in {192, 576, 768}) with each
    compact epilogue it serves, ragged last stage included -- tests/gemm_wreg_check.py in a subprocess (the library reads the variable
    once), with SA_GEMM_WREG=2 so that a launch the kernel does not cover fails instead of silently taking the tiled kernels."""
    import os, subprocess, sys
    e = dict(os.environ); e["SA_GEMM_WREG"] = "2"
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "gemm_wreg_check.py")], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok wreg" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_gemm_elementwise_at_step_shapes(dev):
    """Every output ELEMENT of the tiled kernels' compact epilogues at step-sized shapes against an fp64 reference (norm-relative bounds can
    hide a handful of corrupted elements).  Motivation: a store-data hazard found while building gemm_wreg.hip -- a 16-byte buffer store
    with an SGPR in its scalar-offset field (which hipcc does not pad) followed, one instruction later, by a VALU write of a data
    register stored the NEW value for some lanes; scripts/diag/scan_store_hazard.py found the same instruction pattern in the tiled kernels' ISA (one source site, since
    rewritten to per-lane offsets); this test showed it never bit there (0 elements out of bound before and after) and stays as the guard."""
    g = torch.Generator(device=dev).manual_seed(0)
    for (M, N, K) in [(63744, 768, 768), (63744, 3072, 768), (127488, 768, 192), (33000 + 77, 768, 3072)]:
        A = torch.randn(M, K, device=dev, generator=g).to(BF16)
        W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(BF16)
        bias = torch.randn(N, device=dev, generator=g)
        res = torch.randn(M, N, device=dev, generator=g)
        aux = torch.randn(M, N, device=dev, generator=g).to(BF16)
        acc = A.double() @ W.double().t()
        scale = float(acc.abs().max())
        for tag, kw, ref, is32 in [("bias->bf16", dict(bias=bias), acc + bias.double(), False),
                                   ("bias+res->f32", dict(bias=bias, residual=res), acc + bias.double() + res.double(), True),
                                   ("x aux + colsum", dict(act=4, aux_in=aux), acc * aux.double(), False),
                                   ("gelu pair", dict(bias=bias, act=3), torch.nn.functional.gelu(acc + bias.double()), False)]:
            kw = dict(kw)
            if is32:
                out = torch.empty(M, N, device=dev); ops.gemm(A, W, out_f32=out, **kw)
            else:
                out = torch.empty(M, N, device=dev, dtype=BF16)
                if tag == "gelu pair": kw["aux_out"] = torch.empty_like(out)
                if tag.startswith("x aux"): kw["colsum_out"] = torch.zeros(N, device=dev)
                ops.gemm(A, W, out_bf16=out, **kw)
            # one rounding of the element (bf16: 2^-8, fp32: 2^-20 relative) + accumulation noise (+ the erf approximation's 2e-3 absolute for GELU)
            bound = (2.0 ** -20 if is32 else 2.0 ** -8) * ref.abs() + 2e-5 * scale * (K / 768) ** 0.5 + (2e-3 if tag == "gelu pair" else 0.0)
            nbad = int(((out.double() - ref).abs() > bound).sum())
            assert nbad == 0, (M, N, K, tag, nbad)
        del acc, res, aux
    # the ring kernel (k-strided weight: the data-gradient layout without a transposed copy) shares the bf16 row stores: plain and x GELU'(aux)
    M, N, K = 63744, 768, 3072
    A = torch.randn(M, K, device=dev, generator=g).to(BF16)
    W = (torch.randn(K, N, device=dev, generator=g) * 0.05).to(BF16)
    pre = torch.randn(M, N, device=dev, generator=g).to(BF16)
    acc = A.double() @ W.double()
    scale = float(acc.abs().max())
    hh = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(hh).sum().backward()
    for tag, kw, ref in [("NN ->bf16", dict(), acc), ("NN x GELU'(aux)", dict(act=2, aux_in=pre), acc * hh.grad)]:
        out = torch.empty(M, N, device=dev, dtype=BF16)
        ops.gemm(A, W, b_kmajor=False, out_bf16=out, **kw)
        bound = 2.0 ** -8 * ref.abs() + 2e-5 * scale * (K / 768) ** 0.5
        nbad = int(((out.double() - ref).abs() > bound).sum())
        assert nbad == 0, (tag, nbad)


@pytest.mark.parametrize("M,N,K", [(300, 192, 128), (4000, 2112, 256)])
def test_gemm_fused_column_sums(dev, M, N, K):
    """colsum_out += column sums of the fp32 epilogue result (fc1's bias gradient taken in the fc2-dgrad epilogue), small
    (128^2 kernel) and large (ring kernel, ragged last row tile) problems, NN layout with the GELU' epilogue."""
    A = bf(rnd((M, K), 31)); W = bf(rnd((K, N), 32, 0.1)); pre = bf(rnd((M, N), 33))
    acc = A.double() @ W.double()
    hh = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(hh).sum().backward()
    ref = acc * hh.grad
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    cs = torch.full((N,), 2.0, device=dev)
    ops.gemm(A.to(dev), W.to(dev), b_kmajor=False, act=2, aux_in=pre.to(dev), out_bf16=out, colsum_out=cs)
    assert rel_err(out, ref) < 4e-3
    assert rel_err(cs, 2.0 + ref.sum(0)) < 1e-4
    with pytest.raises(RuntimeError, match="colsum_out"):
        ops.gemm(A.to(dev), W[:, :N - 8].contiguous().to(dev), b_kmajor=False, out_bf16=out[:, :N - 8], colsum_out=cs[:N - 8])


def test_cast_and_colsum(dev):
    x = rnd((1000, 300), 10)
    y = ops.cast_bf16(x.to(dev))
    assert torch.equal(y.cpu(), bf(x))
    out = torch.empty(300, device=dev)
    ops.colsum_bf16(y, out)
    assert rel_err(out, bf(x).double().sum(0)) < 1e-5
    ops.colsum_bf16(y, out, accumulate=True)
    assert rel_err(out, 2 * bf(x).double().sum(0)) < 1e-5
    # two column ranges of one packed matrix in the same launches (q / v bias gradients out of dqkv; models/mae.py:125-128): both the
    # vectorised (d % 8 == 0) and the scalar kernel, ordered (workspace) and atomic forms, gap columns of the output untouched
    for d, M in ((192, 5000), (100, 777)):
        dq = bf(rnd((M, 3 * d), 14)).to(dev)
        ref = dq.double().cpu().sum(0)
        for det in (True, False):
            ops.DETERMINISTIC_WGRAD, keep = det, ops.DETERMINISTIC_WGRAD
            g3 = torch.full((3 * d,), 7.0, device=dev)
            ops.colsum_qv(dq, d, g3[:d], g3[2 * d:])
            ops.DETERMINISTIC_WGRAD = keep
            assert rel_err(g3[:d], 7.0 + ref[:d]) < 1e-5 and rel_err(g3[2 * d:], 7.0 + ref[2 * d:]) < 1e-5
            assert torch.all(g3[d:2 * d] == 7.0)
        gq, gv = torch.zeros(d, device=dev), torch.zeros(d, device=dev)          # separate buffers: the two-launch route
        ops.colsum_qv(dq, d, gq, gv)
        assert rel_err(gq, ref[:d]) < 1e-5 and rel_err(gv, ref[2 * d:]) < 1e-5


# ------------------------------------------------------------------------------------------------ LayerNorm
# (D <= 256: four rows per wave, D <= 512: two -- with row counts that leave the last wave partly empty)
@pytest.mark.parametrize("M,D", [(257, 768), (64, 192), (33, 1024), (5, 64), (4099, 192), (1, 192), (403, 256), (1031, 384), (7, 512), (130, 100)])
def test_layernorm(dev, M, D):
    x = rnd((M, D), 11, 2.0) + 0.3; g = rnd((D,), 12) * 0.2 + 1; b = rnd((D,), 13) * 0.2
    xd = x.double().requires_grad_(True); gd = g.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (D,), gd, bd, 1e-6)
    y32 = torch.empty(M, D, device=dev); y16 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev), 1e-6, y_bf16=y16, y_f32=y32, mean=mean, rstd=rstd)
    assert rel_err(y32, ref) < 2e-6 and rel_err(y16, ref) < 4e-3
    dy = rnd((M, D), 14); dres = rnd((M, D), 15)
    ref.backward(dy.double())
    dx = torch.empty(M, D, device=dev); dx16 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    dxs = torch.zeros(D, device=dev)
    ops.layernorm_bwd(dy.to(dev), x.to(dev), g.to(dev), mean, rstd, dres=dres.to(dev), dx_f32=dx, dx_bf16=dx16, dgamma=dg, dbeta=db, dxsum=dxs)
    assert rel_err(dx, xd.grad + dres.double()) < 5e-6
    assert rel_err(dxs, (xd.grad + dres.double()).sum(0)) < 1e-5        # fused bias-gradient column sum of the output
    assert rel_err(dx16, xd.grad + dres.double()) < 4e-3
    assert rel_err(dg, gd.grad) < 1e-5 and rel_err(db, bd.grad) < 1e-5
    # bf16 upstream gradient path (what the dgrad GEMMs hand over)
    ops.layernorm_bwd(bf(dy).to(dev), x.to(dev), g.to(dev), mean, rstd, dx_f32=dx)
    xd.grad = None
    torch.nn.functional.layer_norm(xd, (D,), gd, bd, 1e-6).backward(bf(dy).double())
    assert rel_err(dx, xd.grad) < 5e-6


def test_layernorm_strided_cls_rows(dev):
    """Final norm on the CLS rows only: x viewed with ld = N*d."""
    S, N, D = 6, 25, 192
    x = rnd((S, N, D), 16); g = torch.ones(D); b = torch.zeros(D)
    y = torch.empty(S, D, device=dev)
    xd = x.to(dev)
    ops.layernorm_fwd(xd.view(S, N * D)[:, :D], g.to(dev), b.to(dev), 1e-6, y_f32=y)
    assert rel_err(y, torch.nn.functional.layer_norm(x[:, 0].double(), (D,), eps=1e-6)) < 2e-6


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(qkv, H, N):
    rows, w = qkv.shape
    C_ = w // 3
    S = rows // N
    q, k, v = qkv.double().view(S, N, 3, H, C_ // H).permute(2, 0, 3, 1, 4)
    a = ((q @ k.transpose(-2, -1)) * (C_ // H) ** -0.5).softmax(-1)
    return (a @ v).transpose(1, 2).reshape(rows, C_)


@pytest.mark.parametrize("S,H,N", [(3, 3, 249), (2, 2, 25), (2, 1, 63), (1, 12, 256), (2, 2, 40), (1, 1, 1),
                                   (2, 3, 501), (1, 2, 257), (1, 1, 512)])       # > 256 tokens: the NMAX = 512 instantiation (16 x 8 patches at 10 s = 501)
def test_attention_fwd_bwd(dev, S, H, N):
    """bf16 in/out, P in bf16: forward rel err <= 1e-2 vs fp64 on the same bf16 inputs, gradients <= 2e-2."""
    C_ = 64 * H
    qkv = bf(rnd((S * N, 3 * C_), 17, 1.0))
    # an outlier key per sequence exercises the max-subtraction
    qkv[5 % (S * N), C_:C_ + 64] *= 6.0
    qd = qkv.double().requires_grad_(True)
    ref = attn_ref(qd, H, N)
    out = torch.full((S * N, C_), float("nan"), device=dev, dtype=torch.bfloat16)
    lse = torch.empty(S * H, N, device=dev)
    ops.attention_fwd(qkv.to(dev), H, N, 0.125, out, lse)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert rel_err(out, ref) < 1e-2
    q, k, _ = qkv.double().view(S, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    lse_ref = torch.logsumexp((q @ k.transpose(-2, -1)) * 0.125, -1).reshape(S * H, N)
    assert float((lse.cpu().double() - lse_ref).abs().max()) < 2e-2
    dout = bf(rnd((S * N, C_), 18))
    ref.backward(dout.double())
    dqkv = torch.full((S * N, 3 * C_), float("nan"), device=dev, dtype=torch.bfloat16)
    ops.attention_bwd(qkv.to(dev), H, N, 0.125, out, dout.to(dev), lse, dqkv)
    torch.cuda.synchronize()
    assert torch.isfinite(dqkv.float()).all()
    for name, sl in [("dq", slice(0, C_)), ("dk", slice(C_, 2 * C_)), ("dv", slice(2 * C_, 3 * C_))]:
        assert rel_err(dqkv[:, sl], qd.grad[:, sl]) < 2e-2, name


# ------------------------------------------------------------------------------------------------ BN pieces / loss pieces
@pytest.mark.parametrize("B,C_", [(128, 8192), (7, 24), (32, 256)])
def test_bn_pieces(dev, B, C_):
    x = rnd((B, C_), 19, 1.5) + 0.2; g = rnd((C_,), 20) * 0.2 + 1; b = rnd((C_,), 21) * 0.2
    xd = x.double().requires_grad_(True); gd = g.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    mu = xd.mean(0); var = xd.var(0, unbiased=False)
    ref = torch.relu((xd - mu) * torch.rsqrt(var + 1e-5) * gd + bd)
    mean = torch.empty(C_, device=dev); m2 = torch.empty(C_, device=dev)
    ops.bn_colstats(x.to(dev), mean, m2)
    assert rel_err(mean, mu) < 1e-5 and rel_err(m2 / B, var) < 1e-5
    rstd = torch.rsqrt(m2 / B + 1e-5)
    y = torch.empty(B, C_, device=dev); y16 = torch.empty(B, C_, device=dev, dtype=torch.bfloat16)
    ops.bn_apply(x.to(dev), mean, rstd, g.to(dev), b.to(dev), True, y_f32=y, y_bf16=y16)
    assert rel_err(y, ref) < 5e-6 and rel_err(y16, ref) < 4e-3
    dy = rnd((B, C_), 22)
    ref.backward(dy.double())
    s1 = torch.empty(C_, device=dev); s2 = torch.empty(C_, device=dev)
    ops.bn_bwd_stats(dy.to(dev), x.to(dev), mean, rstd, g.to(dev), b.to(dev), True, s1, s2)
    assert rel_err(s1, bd.grad) < 1e-5 and rel_err(s2, gd.grad) < 1e-5
    dx = torch.empty(B, C_, device=dev)
    ops.bn_bwd_apply(dy.to(dev), x.to(dev), mean, rstd, g.to(dev), b.to(dev), True, s1, s2, 1.0 / B, dx_f32=dx)
    assert rel_err(dx, xd.grad) < 2e-5


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (24, 24, 7), (128, 256, 256), (33, 70, 65)])
def test_matmul_f32(dev, M, N, K):
    """Exact-fp32 MFMA: error of an fp32 dot product (<= 1e-6 relative), all transposes."""
    A = rnd((M, K), 23); B = rnd((K, N), 24)
    ref = A.double() @ B.double()
    out = torch.empty(M, N, device=dev)
    ops.matmul_f32(A.to(dev), B.to(dev), out)
    assert rel_err(out, ref) < 1e-6
    ops.matmul_f32(A.T.contiguous().to(dev), B.to(dev), out, trans_a=True, alpha=0.5)
    assert rel_err(out, 0.5 * ref) < 1e-6
    ops.matmul_f32(A.to(dev), B.T.contiguous().to(dev), out, trans_b=True)
    assert rel_err(out, ref) < 1e-6


@pytest.mark.parametrize("hsic", [False, True])
def test_bt_loss_grad(dev, hsic):
    D = 256
    c = rnd((D, D), 25, 0.1) + torch.eye(D) * 0.9
    cd = c.double().requires_grad_(True)
    on = (torch.diagonal(cd) - 1).pow(2).sum()
    offm = cd - torch.diag(torch.diagonal(cd))
    off = ((offm + (1 - torch.eye(D, dtype=torch.float64))) if hsic else offm).pow(2).sum()
    ref = 1.0 * on + 0.005 * off
    ref.backward()
    loss = torch.empty(1, device=dev); G = torch.empty(D, D, device=dev)
    ops.bt_loss_grad(c.to(dev), 1.0, 0.005, hsic, loss, G)
    assert abs(float(loss) - float(ref)) / float(ref) < 1e-5
    assert rel_err(G, cd.grad) < 1e-6


@pytest.mark.parametrize("n", [100003, 100000])          # scalar path / 16-byte vector path
def test_adamw_and_ema(dev, n):
    p = rnd((n,), 26); g = rnd((n,), 27, 0.1)
    pr = torch.nn.Parameter(p.clone().double())
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=0.06)
    pd, m, v = p.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    p16 = torch.empty(n, device=dev, dtype=torch.bfloat16)
    for step in (1, 2, 3):
        pr.grad = g.double() * step
        opt.step()
        ops.adamw_step(pd, (g * step).to(dev), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.06, step, p_bf16=p16)
    assert float((pd.cpu().double() - pr.data).abs().max()) < 2e-6
    assert torch.equal(p16.cpu(), bf(pd.cpu()))
    t = rnd((n,), 28).to(dev)
    t0 = t.cpu().clone()
    ops.ema_update(t, pd, 0.99)
    assert float((t.cpu() - (0.99 * t0 + 0.01 * pd.cpu())).abs().max()) < 1e-6


# ------------------------------------------------------------------------------------------------ frontend
def synth_wave(n, L, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(L, dtype=torch.float64) / 16000.0
    w = 0.1 * torch.randn(n, L, generator=g, dtype=torch.float64)
    for _ in range(3):
        f = 100 + 6900 * torch.rand(n, 1, generator=g, dtype=torch.float64)
        a = 0.05 + 0.45 * torch.rand(n, 1, generator=g, dtype=torch.float64)
        w = w + a * torch.sin(2 * math.pi * f * t)
    return w.float()


@pytest.mark.parametrize("L,T,start", [(160000, 1001, 0), (16000, 101, 0), (15200, 96, 0), (16000, 96, 3), (16000, 120, 0),
                                       (600, 16, 0), (2000, 5, 8), (1030, 7, 0), (4000, 33, 0)])
def test_logmel(dev, L, T, start):
    """fp32 FFT vs the fp64 oracle: |diff| <= 2e-3 in the log-mel domain (values span ~[-8, 6]).  The short cases: a clip barely longer than
    half a window (every frame reflects at both ends), a crop that ends on the last frame, clips shorter than the crop (right zero padding before the
    normalisation, datasets.py:346-354), group counts of 1 - 3 with ragged last groups."""
    from oracle import frontend as ofe
    wave = synth_wave(3, L, 29)
    mel = fe.MelSpectrogram()
    out = mel(wave.to(dev), crop_frames=T, start=start, norm_stats=(-0.8294, 4.6230))
    ref = ofe.crop_pad_normalize(ofe.logmel(wave.numpy()), T, start, -0.8294, 4.6230)
    d = np.abs(out.cpu().numpy()[:, 0] - ref)
    assert out.shape == (3, 1, 64, T)
    assert d.max() < 2e-3, d.max()


def test_logmel_per_clip_start_length_offset(dev):
    """ABI v6: ONE launch does what Dataset.__getitem__ does per sample (datasets.py:342-351): every clip its own crop start
    (`np.random.randint(l - crop_frames)`), its own length (frames end -- and the reflect padding mirrors -- at ITS last sample; shorter
    than the crop: right zero pad before the normalisation) and, for the waveform-level crop of datasets.py:108-112, its own first
    sample inside the row (odd offsets take the element-wise load path).  Against the oracle run clip by clip on the cropped waveform."""
    from oracle import frontend as ofe
    T, hop = 96, 160
    lens = [160000, 15200, 48000, 9000, 600, 160000, 31999, 16000, 511, 20000]     # 511 samples: too short to reflect-pad -> all pad
    offs = [0, 2, 1001, 0, 64, 0, 7, 3, 0, 139999]
    n = len(lens)
    wave = synth_wave(n, 160000, 43)
    rng = np.random.RandomState(3)
    starts = []
    for L_, o in zip(lens, offs):
        assert o + L_ <= 160000
        l = 1 + L_ // hop
        starts.append(int(rng.randint(l - T)) if l > T else 0)
    starts[5] = (1 + 160000 // hop) - T                      # a crop that ends on the clip's last frame
    starts[2] = (1 + 48000 // hop) - 40                      # a "start" past l - crop: 40 live frames, then pad (the kernel must not read on)
    mel = fe.MelSpectrogram()
    out = mel(wave.to(dev), crop_frames=T, start=starts, lengths=lens, offsets=offs, norm_stats=(-0.8294, 4.6230)).cpu().numpy()[:, 0]
    pad = (0.0 + 0.8294) / 4.6230
    for b in range(n):
        if lens[b] <= 512:
            assert np.all(out[b] == np.float32(pad)), b
            continue
        lms = ofe.logmel(wave[b, offs[b]:offs[b] + lens[b]].numpy())
        ref = (np.pad(lms[:, starts[b]:starts[b] + T], [(0, 0), (0, max(0, T - (lms.shape[-1] - starts[b])))]) + 0.8294) / 4.6230
        d = np.abs(out[b] - ref)
        assert d.max() < 2e-3, (b, d.max())
    # the arrays are optional one by one, and device tensors are accepted as they are
    o2 = mel(wave.to(dev), crop_frames=T, start=torch.tensor(starts, dtype=torch.int32, device=dev), norm_stats=(-0.8294, 4.6230))
    o3 = mel(wave[:1].to(dev), crop_frames=T, start=starts[0], norm_stats=(-0.8294, 4.6230))
    assert torch.equal(o2[0], o3[0])
    with pytest.raises(ValueError):
        mel(wave.to(dev), crop_frames=T, start=starts[:3])


def test_logmel_per_clip_out_of_range_values_are_clamped(dev):
    """The per-clip vectors come from the caller's sampler: values that would reach outside a clip's row must not turn into out-of-range
    reads (the kernel clamps; include/ssl_audio_hip.h).  Zero or too-short lengths, an offset at / past the row's end and a start past
    the clip's last frame give the normalised zero pad; a negative start or offset counts as 0; a length longer than the row is cut to it."""
    from oracle import frontend as ofe
    T, n = 40, 16000
    wave = synth_wave(7, n, 53)
    pad = np.float32((0.0 + 0.8294) / 4.6230)
    starts = [0, 0, 0, 10 ** 6, -5, 0, 3]
    lens = [0, 512, n, n, n, 10 ** 7, n]
    offs = [0, 0, n, 0, -9, 0, n - 700]
    out = fe.MelSpectrogram()(wave.to(dev), crop_frames=T, start=starts, lengths=lens, offsets=offs, norm_stats=(-0.8294, 4.6230)).cpu().numpy()[:, 0]
    for b in (0, 1, 2, 3):
        assert np.all(out[b] == pad), b
    full = ofe.crop_pad_normalize(ofe.logmel(wave.numpy()), T, 0, -0.8294, 4.6230)
    assert np.abs(out[4] - full[4]).max() < 2e-3 and np.abs(out[5] - full[5]).max() < 2e-3
    tail = ofe.logmel(wave[6, n - 700:].numpy())                                   # 700 samples: 5 frames, 3 skipped -> 2 live frames, then pad
    assert np.abs(out[6][:, :2] - (tail[:, 3:5] + 0.8294) / 4.6230).max() < 2e-3 and np.all(out[6][:, 2:] == pad)
    assert np.isfinite(out).all()


def test_logmel_padding_groups_do_no_transform_work(dev):
    """Clips shorter than the crop: the 16-frame groups that lie wholly in the padding are written without FFT / MFMA work, so a batch
    of 1 s clips padded to 1001 frames costs about a tenth of a batch of 10 s clips (and equals the oracle's right zero pad)."""
    from oracle import frontend as ofe
    B, T = 128, 1001
    wave = synth_wave(B, 160000, 47).to(dev)
    mel = fe.MelSpectrogram()
    lens = torch.full((B,), 16000, dtype=torch.int32, device=dev)       # (a device vector: no host staging inside the timed calls)
    full = mel(wave, crop_frames=T, norm_stats=(-0.8294, 4.6230))
    short = mel(wave, crop_frames=T, lengths=lens, norm_stats=(-0.8294, 4.6230))

    def timed(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
    t_full = timed(lambda: mel(wave, crop_frames=T, norm_stats=(-0.8294, 4.6230), out=full))
    t_short = timed(lambda: mel(wave, crop_frames=T, lengths=lens, norm_stats=(-0.8294, 4.6230), out=short))
    ref = ofe.crop_pad_normalize(ofe.logmel(wave[:2, :16000].cpu().numpy()), T, 0, -0.8294, 4.6230)
    assert np.abs(short[:2, 0].cpu().numpy() - ref).max() < 2e-3
    assert torch.all(short[:, :, :, 101:] == short[0, 0, 0, 1000])
    print(f"logmel 128 clips: 10 s {t_full * 1e3:.1f} us, 1 s padded to 1001 frames {t_short * 1e3:.1f} us")
    assert t_short < 0.5 * t_full, (t_short, t_full)


def test_logmel_many_groups(dev):
    """More 16-frame groups than the persistent kernel has workgroups (3 per CU): every workgroup walks several (clip, group) pairs,
    the last group of a clip is ragged (101 frames = 6 x 16 + 5) and the first frame of the next pair is prefetched across the boundary."""
    from oracle import frontend as ofe
    n = 3 * torch.cuda.get_device_properties(0).multi_processor_count // 7 + 9
    wave = synth_wave(n, 16000, 31)
    out = fe.MelSpectrogram()(wave.to(dev), crop_frames=101, start=0, norm_stats=(-0.8294, 4.6230))
    ref = ofe.crop_pad_normalize(ofe.logmel(wave.numpy()), 101, 0, -0.8294, 4.6230)
    d = np.abs(out.cpu().numpy()[:, 0] - ref)
    assert d.max() < 2e-3, d.max()


def test_logmel_full_size_properties(dev):
    """BASELINE's frontend shape (128 clips of 10 s: 8 064 groups of 16 frames over 768 persistent workgroups), checked through properties
    that need no oracle: (i) a clip delayed by one hop gives the same frames one column later, BIT FOR BIT away from the clip's ends (a frame's
    arithmetic does not depend on where in a clip, a group or a workgroup's walk it sits); (ii) a gain of a shifts the un-normalised
    log-mel by 2 ln a wherever the power is far above eps; (iii) two launches agree bit for bit."""
    B, L, T, hop = 128, 160000, 1001, 160
    g = torch.Generator(device=dev).manual_seed(41)
    base = 0.1 * torch.randn(B, L + hop, device=dev, generator=g)
    mel = fe.MelSpectrogram()
    a = mel(base[:, hop:].contiguous(), crop_frames=T, start=0)[:, 0]          # frames of x[hop:]
    b = mel(base[:, :L].contiguous(), crop_frames=T, start=0)[:, 0]            # frames of x[:L]: frame t + 1 sees what frame t of `a` sees
    assert torch.equal(a[:, :, 4:T - 5], b[:, :, 5:T - 4])                     # (4 frames at each end touch the reflect padding)
    assert torch.equal(a, mel(base[:, hop:].contiguous(), crop_frames=T, start=0)[:, 0])
    c = mel((3.0 * base[:, hop:]).contiguous(), crop_frames=T, start=0)[:, 0]
    assert float((c - a - 2.0 * math.log(3.0)).abs().max()) < 1e-4
    assert torch.isfinite(a).all()


@pytest.mark.parametrize("bank", ["wide", "sparse"])
def test_logmel_other_filter_banks(dev, bank):
    """sa_logmel_fwd takes the filter bank as a table (per band: first bin, length, weights): a bank whose 16-band groups cover more bins
    than the kernel holds weights for in registers ("wide": 60-72 bins per band, the one-product-at-a-time path) and one that leaves whole
    groups empty ("sparse": 8 bands of 4 bins, 56 bands of none -> log(eps)), against W . |STFT|^2 from the oracle's power spectrogram."""
    from oracle import frontend as ofe
    rng = np.random.RandomState(5)
    if bank == "wide":
        lo = (4 + 7 * np.arange(64)).astype(np.int32)
        ln = np.minimum(60 + 3 * (np.arange(64) % 5), 513 - lo).astype(np.int32)
    else:
        lo = np.where(np.arange(64) < 8, 10 + 5 * np.arange(64), 0).astype(np.int32)
        ln = np.where(np.arange(64) < 8, 4, 0).astype(np.int32)
    maxlen = int(ln.max())
    w = np.zeros((maxlen, 64))
    dense = np.zeros((64, 513))
    for m in range(64):
        w[:ln[m], m] = rng.rand(ln[m]) / max(int(ln[m]), 1)
        dense[m, lo[m]:lo[m] + ln[m]] = w[:ln[m], m].astype(np.float32)
    wave = synth_wave(2, 16000, 37)
    tb = dict(fe.build_tables(dev, n_fft=1024, n_mels=64, f_min=60.0, f_max=7800.0, sample_rate=16000, win_length=1024))
    tb["mel_weights"] = torch.tensor(w, dtype=torch.float32, device=dev)
    tb["mel_lo"], tb["mel_len"] = torch.tensor(lo, device=dev), torch.tensor(ln, device=dev)
    out = torch.empty(2, 64 * 101, device=dev)
    ops.logmel_fwd(wave.to(dev), tb, out, 101, 0, 0.0, 1.0, 160)
    ref = np.log(np.einsum("mk,nkt->nmt", dense, ofe.power_spectrogram(wave.numpy())) + np.finfo(np.float32).eps)
    d = np.abs(out.view(2, 64, 101).cpu().numpy() - ref)
    assert d.max() < 2e-3, d.max()


# ------------------------------------------------------------------------------------------------ augmentation
def run_views(dev, lms, recs, out_size, canvas, vcs_ratio, do_fade=True):
    V = len(recs)
    F_in, T_in = lms.shape[-2:]
    params = torch.zeros(V, 8)
    src = torch.zeros(V, dtype=torch.int32); mix = torch.full((V,), -1, dtype=torch.int32)
    for v, r in enumerate(recs):
        i, j, h, w = r["rrc"]
        ht = r.get("head_tail", (0.0, 0.0))
        params[v] = torch.tensor([r.get("alpha", 0.0), i, j, h, w, ht[0], ht[1], 0.0])
        src[v] = r["src"]; mix[v] = r.get("mix", -1)
    out = torch.empty(V, 1, out_size[0], out_size[1], device=dev)
    ops.augment_views(lms.to(dev), F_in * T_in, src.to(dev), mix.to(dev), params.to(dev), out, F_in, T_in, canvas, vcs_ratio, do_fade)
    return out.cpu().numpy()


def test_augment_rrc_golden(dev, golden):
    """RandomResizeCrop alone against vectors captured from the reference: |diff| <= 1e-4."""
    g = golden("augment")
    for tag in ["t96", "t96b", "t1001", "t1001b", "local"]:
        x = torch.from_numpy(g[f"rrc_{tag}_x"])
        cfg = g[f"rrc_{tag}_cfg"]
        out_size = (int(cfg[0]), int(cfg[1]))
        canvas = (int(x.shape[-2] * cfg[2]), int(x.shape[-1] * cfg[3]))
        rec = {"rrc": tuple(int(v) for v in g[f"rrc_{tag}_params"]), "src": 0}
        y = run_views(dev, x, [rec], out_size, canvas, canvas[1] / max(out_size[1] - 1, 1), do_fade=False)
        assert np.abs(y[0] - g[f"rrc_{tag}_y"]).max() < 1e-4, tag


@pytest.mark.parametrize("tag", ["seq96", "seq208"])
def test_augment_sequence_golden(dev, golden, tag):
    """Whole AudioPairTransform sequence (mixup bank + RRC + fader) against the reference's outputs, driven by the
    oracle's recorded draws: the batched kernel reproduces the sequential bank semantics exactly."""
    from oracle import augment as oaug
    g = golden("augment")
    clips = g[f"apt_{tag}_clips"]
    T_ = clips.shape[-1]
    tfm = oaug.PairTransformOracle(crop_frames=T_, seed=int(g[f"apt_{tag}_seed"]))
    for c in clips:
        tfm(c)
    recs = []
    for e, r in enumerate(tfm.records):          # event e = 2*clip + view; bank entry k <-> clip k // 2
        recs.append({"alpha": r["alpha"], "rrc": r["rrc"], "head_tail": r["head_tail"], "src": e // 2,
                     "mix": r["bank_index"] // 2 if r["bank_index"] >= 0 else -1})
    lms = torch.from_numpy(clips[:, 0])
    canvas = (64, int(T_ * 1.5))
    y = run_views(dev, lms, recs, (64, T_), canvas, canvas[1] / (T_ - 1))
    ref = g[f"apt_{tag}_views"].reshape(-1, 1, 64, T_)
    assert np.abs(y - ref).max() < 2e-4


def test_normalize_batch_and_patchify(dev):
    x = rnd((5, 1, 64, 96), 30, 3.0) + 1
    y = torch.empty_like(x, device=dev); ws = torch.zeros(2, dtype=torch.float64, device=dev)
    ops.normalize_batch(x.to(dev), y, 1.0, ws, 1.1920929e-07)
    ref = (x.double() - x.double().mean()) / x.double().std()
    assert float((y.cpu().double() - ref).abs().max()) < 5e-6
    img = rnd((3, 1, 64, 1001), 31)
    out = torch.empty(3 * 4 * 62, 256, device=dev, dtype=torch.bfloat16)
    ops.patchify_bf16(img.to(dev), out, 16, 16)
    ref = img[..., :992].reshape(3, 1, 4, 16, 62, 16).permute(0, 2, 4, 3, 5, 1).reshape(3 * 248, 256)
    assert torch.equal(out.cpu(), bf(ref))


def test_token_ops(dev):
    S, N, d = 4, 9, 64
    x = torch.zeros(S, N, d, device=dev); cls = rnd((d,), 32).to(dev); pos0 = rnd((d,), 33).to(dev)
    ops.fill_cls(x, S, N * d, d, cls, pos0)
    assert torch.allclose(x[:, 0].cpu(), (cls + pos0).cpu().expand(S, d)) and float(x[:, 1:].abs().max()) == 0
    dx = rnd((S, N, d), 34).to(dev); dcls = torch.zeros(d, device=dev)
    ops.cls_grad(dx, S, N * d, d, dcls)
    assert rel_err(dcls, dx[:, 0].sum(0)) < 1e-6
    idx = torch.stack([torch.randperm(N - 1)[:5] for _ in range(S)]).to(torch.int32).to(dev)
    src = rnd((S, N, d), 35).to(dev); dst = torch.zeros(S, 6, d, device=dev)
    ops.gather_rows(src, N * d, 1, idx, dst, 6 * d, 1, S, d)
    ref = torch.gather(src[:, 1:], 1, idx.long().unsqueeze(-1).expand(-1, -1, d))
    assert torch.equal(dst[:, 1:], ref)
    back = torch.zeros(S, N, d, device=dev)
    ops.scatter_add_rows(dst, 6 * d, 1, idx, back, N * d, 1, S, d)
    chk = torch.zeros(S, N - 1, d, device=dev).scatter_add_(1, idx.long().unsqueeze(-1).expand(-1, -1, d), ref)
    assert torch.equal(back[:, 1:], chk)


def test_attention_n_query(dev):
    """n_query = 1 (the CLS-only last block): forward equals row 0 of the full result; backward equals the full kernel fed
    with an upstream gradient that is zero outside row 0 (dQ rows > 0 come out exactly zero)."""
    S, H, N = 3, 2, 249
    C_ = 64 * H
    qkv = bf(rnd((S * N, 3 * C_), 40)).to(dev)
    full = torch.empty(S * N, C_, device=dev, dtype=torch.bfloat16); lse_full = torch.empty(S * H, N, device=dev)
    ops.attention_fwd(qkv, H, N, 0.125, full, lse_full)
    part = torch.zeros(S * N, C_, device=dev, dtype=torch.bfloat16); lse = torch.zeros(S * H, N, device=dev)
    ops.attention_fwd(qkv, H, N, 0.125, part, lse, n_query=1)
    rows0 = torch.arange(S, device=dev) * N
    assert torch.equal(part[rows0], full[rows0]) and float(part.float().abs().sum()) == float(part[rows0].float().abs().sum())
    assert torch.equal(lse[:, 0], lse_full[:, 0])
    dout = torch.zeros(S * N, C_, device=dev, dtype=torch.bfloat16)
    dout[rows0] = bf(rnd((S, C_), 41)).to(dev)
    ref = torch.empty(S * N, 3 * C_, device=dev, dtype=torch.bfloat16)
    ops.attention_bwd(qkv, H, N, 0.125, full, dout, lse_full, ref)
    got = torch.full((S * N, 3 * C_), float("nan"), device=dev, dtype=torch.bfloat16)
    ops.attention_bwd(qkv, H, N, 0.125, part, dout, lse, got, n_query=1)
    assert torch.isfinite(got.float()).all()
    assert rel_err(got, ref) < 1e-6
    mask = torch.ones(S * N, dtype=torch.bool, device=dev); mask[rows0] = False
    assert float(got[mask][:, :C_].float().abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ MAE decoder glue, mean pooling
def test_mean_tokens(dev):
    S, N, d = 5, 9, 64
    y = rnd((S, N, d), 70)
    out = torch.empty(S, d, device=dev)
    ops.mean_tokens_fwd(y.to(dev), out)
    assert rel_err(out, y[:, 1:].double().mean(1)) < 1e-6
    g = rnd((S, d), 71)
    dy = torch.full((S, N, d), float("nan"), device=dev)
    ops.mean_tokens_bwd(g.to(dev), dy)
    ref = torch.zeros(S, N, d, dtype=torch.float64); ref[:, 1:] = (g.double() / (N - 1)).unsqueeze(1)
    assert rel_err(dy, ref) < 1e-6 and float(dy[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("B,L,keep,d", [(3, 24, 6, 64), (2, 248, 62, 384), (2, 10, 10, 32), (2, 10, 0, 32)])
def test_mae_unshuffle(dev, B, L, keep, d):
    """forward_decoder's token assembly and its adjoint vs the reference formulation (cat / gather, models/mae.py:413-420)."""
    g = torch.Generator().manual_seed(72)
    x = torch.randn(B, 1 + keep, d, generator=g, requires_grad=True)
    mt = torch.randn(1, 1, d, generator=g, requires_grad=True)
    pos = torch.randn(1, 1 + L, d, generator=g)
    ids_restore = torch.argsort(torch.argsort(torch.rand(B, L, generator=g), dim=1), dim=1)
    mask_tokens = mt.repeat(B, L + 1 - x.shape[1], 1)
    x_ = torch.cat([x[:, 1:, :], mask_tokens], dim=1)
    x_ = torch.gather(x_, dim=1, index=ids_restore.unsqueeze(-1).repeat(1, 1, d))
    ref = torch.cat([x[:, :1, :], x_], dim=1) + pos
    w = torch.randn(B, 1 + L, d, generator=g)
    (ref * w).sum().backward()
    out = torch.empty(B, 1 + L, d, device=dev)
    ids = ids_restore.to(torch.int32).to(dev)
    ops.mae_unshuffle_fwd(x.detach().to(dev), mt.detach().reshape(-1).to(dev), pos.reshape(-1, d).to(dev), ids, out)
    assert torch.equal(out.cpu(), ref.detach())                       # pure data movement + one add: bit exact
    dx = torch.full((B, 1 + keep, d), float("nan"), device=dev)
    dm = torch.zeros(d, device=dev)
    ops.mae_unshuffle_bwd(w.to(dev), keep, ids, dx, dm)
    assert torch.equal(dx.cpu(), x.grad)
    assert rel_err(dm, mt.grad.reshape(-1).double()) < 1e-5 if keep < L else float(dm.abs().max()) == 0.0


@pytest.mark.parametrize("B,F,T,row0", [(3, 64, 96, 1), (2, 64, 992, 1), (2, 32, 48, 0)])
def test_mae_recon_loss(dev, B, F, T, row0):
    """forward_loss + patchify (models/mae.py:437-453, 282-293) and its gradient vs autograd on the reference formulation."""
    ph = pw = 16
    h, w = F // ph, T // pw
    L, P = h * w, ph * pw
    g = torch.Generator().manual_seed(73)
    imgs = torch.randn(B, 1, F, T, generator=g)
    pred_full = torch.randn(B, row0 + L, P, generator=g, requires_grad=True)
    mask = (torch.rand(B, L, generator=g) < 0.75).float()
    xx = imgs.reshape(B, 1, h, ph, w, pw)
    target = torch.einsum('nchpwq->nhwpqc', xx).reshape(B, L, P)
    pred = pred_full[:, row0:]
    loss = (((pred - target) ** 2).mean(dim=-1) * mask).sum() / mask.sum()
    (3.0 * loss).backward()
    acc2 = torch.empty(2, device=dev); out = torch.empty(1, device=dev)
    pd = pred_full.detach().to(dev)
    ops.mae_recon_loss_fwd(pd, row0, imgs.to(dev), mask.to(dev), ph, pw, acc2, out)
    assert abs(float(out) - float(loss)) < 1e-5 * abs(float(loss))
    assert float(acc2[1]) == float(mask.sum())
    dpred = torch.full_like(pd, float("nan"))
    ops.mae_recon_loss_bwd(pd, row0, imgs.to(dev), mask.to(dev), ph, pw, acc2, torch.tensor([3.0], device=dev), dpred)
    assert rel_err(dpred, pred_full.grad.double()) < 1e-5


def test_transpose_bf16_and_nt_dgrad(dev):
    """sa_transpose_bf16 (bit-exact, ragged 64-tiles) and the data gradient through the transposed copy: dX = dY W read as NT against W^T is
    bit-identical to the k-strided NN read of W (same products, same fp32 accumulation order per tile), incl. the fused GELU' / column sums."""
    g = torch.Generator().manual_seed(3)
    for R, C in [(768, 2304), (3072, 768), (200, 72), (8, 8), (136, 1000)]:
        a = bf(torch.randn(R, C, generator=g)).to(torch.bfloat16).to(dev)
        assert torch.equal(ops.transpose_bf16(a), a.t().contiguous()), (R, C)
    M, N, K = 3000, 512, 320
    dY = bf(torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev)
    W = bf(torch.randn(K, N, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    aux = bf(torch.randn(M, N, generator=g)).to(torch.bfloat16).to(dev)
    o1, o2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    c1, c2 = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    ops.gemm(dY, W, b_kmajor=False, act=4, aux_in=aux, out_bf16=o1, colsum_out=c1)
    ops.gemm(dY, ops.transpose_bf16(W), b_kmajor=True, act=4, aux_in=aux, out_bf16=o2, colsum_out=c2)
    ref = (dY.double().cpu() @ W.double().cpu()) * aux.double().cpu()
    assert rel_err(o2, ref) < 6e-3 and rel_err(c2, ref.sum(0)) < 2e-3
    assert rel_err(o1, o2.double().cpu()) < 1e-6 and rel_err(c1, c2.double().cpu()) < 1e-5


def test_c_abi_launches_are_graph_capturable(dev):
    """include/ssl_audio_hip.h promises that no entry point allocates, frees or synchronises: a GEMM (fused epilogue + column sums through
    the cached workspace), a LayerNorm, the fused attention and a split-K weight gradient are captured into ONE HIP graph and replayed on
    new inputs; the replay must equal the eager launches bit for bit (the split-K sum in slice order)."""
    g = torch.Generator().manual_seed(9)
    M, d, H, N = 4 * 64, 128, 2, 64
    x = torch.randn(M, d, generator=g).to(dev)
    w = bf(torch.randn(3 * d, d, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    h16 = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    qkv = torch.empty(M, 3 * d, dtype=torch.bfloat16, device=dev)
    ao = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(M // N * H, N, device=dev)
    dw = torch.zeros(3 * d, d, device=dev)
    bias = torch.randn(3 * d, generator=g).to(dev)

    def launches():
        ops.layernorm_fwd(x, gamma, beta, 1e-6, y_bf16=h16, mean=mean, rstd=rstd)
        ops.gemm(h16, w, bias=bias, out_bf16=qkv)
        ops.attention_fwd(qkv, H, N, 0.125, ao, lse)
        dw.zero_()
        ops.gemm(qkv, h16, a_kmajor=False, b_kmajor=False, out_f32=dw, split_k=2)

    old = ops.DETERMINISTIC_WGRAD
    ops.DETERMINISTIC_WGRAD = True
    try:
        launches()                                   # warm-up: workspaces, function attributes
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                launches()
        x.copy_(torch.randn(M, d, generator=g))      # new input, same buffers
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in (h16, qkv, ao, lse, dw)]
        launches()
        torch.cuda.synchronize()
        for a, b in zip(got, (h16, qkv, ao, lse, dw)):
            assert torch.equal(a, b)
        assert float(dw.abs().sum()) > 0
    finally:
        ops.DETERMINISTIC_WGRAD = old


def test_custom_ops_reach_the_same_kernels(dev):
    """torch.ops.ssl_audio.* (torch.library) against the direct ctypes route: same symbol, same bits -- GEMM with a fused epilogue,
    LayerNorm forward, the fused AdamW step (in-place mutation visible through the dispatcher)."""
    import ssl_audio_amd.custom_ops  # noqa: F401
    g = torch.Generator(device=dev).manual_seed(3)
    M, N, K = 300, 192, 128
    A = torch.randn(M, K, device=dev, generator=g).to(BF16); B = torch.randn(N, K, device=dev, generator=g).to(BF16)
    bias = torch.randn(N, device=dev, generator=g); res = torch.randn(M, N, device=dev, generator=g)
    o1, o2 = torch.empty(M, N, device=dev), torch.full((M, N), float("nan"), device=dev)
    ops.gemm(A, B, bias=bias, residual=res, out_f32=o1)
    torch.ops.ssl_audio.gemm(A, B, bias=bias, residual=res, out_f32=o2)
    assert torch.equal(o1, o2)
    x = torch.randn(37, 768, device=dev, generator=g); w = torch.randn(768, device=dev, generator=g); b = torch.randn(768, device=dev, generator=g)
    y1, y2 = torch.empty(37, 768, dtype=BF16, device=dev), torch.empty(37, 768, dtype=BF16, device=dev)
    ops.layernorm_fwd(x, w, b, 1e-6, y_bf16=y1)
    torch.ops.ssl_audio.layernorm_fwd(x, w, b, 1e-6, y_bf16=y2)
    assert torch.equal(y1, y2)
    p1 = torch.randn(4096, device=dev, generator=g); gr = torch.randn(4096, device=dev, generator=g)
    p2, m1, v1, m2, v2 = p1.clone(), torch.zeros(4096, device=dev), torch.zeros(4096, device=dev), torch.zeros(4096, device=dev), torch.zeros(4096, device=dev)
    ops.adamw_step(p1, gr, m1, v1, 1e-3, 0.9, 0.999, 1e-8, 0.05, 1)
    ver = p2._version
    torch.ops.ssl_audio.adamw_step(p2, gr, m2, v2, 1e-3, 0.9, 0.999, 1e-8, 0.05, 1)
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2) and p2._version > ver
    # the frontend operator takes its five tables as five tensors (a `Tensor tables` schema could not be called)
    from ssl_audio_amd import frontend
    tb = frontend.build_tables(dev)
    wave = 0.1 * torch.randn(3, 16000, device=dev, generator=g)
    l1, l2 = torch.empty(3, 64, 96, device=dev), torch.full((3, 64, 96), float("nan"), device=dev)
    ops.logmel_fwd(wave, tb, l1, 96, 2, -0.8, 4.6, 160)
    torch.ops.ssl_audio.logmel_fwd(wave, tb["window"], tb["twiddle"], tb["mel_weights"], tb["mel_lo"], tb["mel_len"], l2, 96, 2, -0.8, 4.6, 160)
    assert torch.equal(l1, l2)
    # column sums: the ordered (workspace) form is bit-reproducible and equals the direct route
    y = torch.randn(5000, 768, device=dev, generator=g).to(BF16)
    c1, c2, c3 = torch.empty(768, device=dev), torch.empty(768, device=dev), torch.empty(768, device=dev)
    ops.colsum_bf16(y, c1); ops.colsum_bf16(y, c2); torch.ops.ssl_audio.colsum_bf16(y, c3)
    assert torch.equal(c1, c2) and torch.equal(c1, c3)
    assert float((c1.double().cpu() - y.double().sum(0).cpu()).abs().max()) < 1e-2


def test_cast_bf16_to_f32_and_back(dev):
    """sa_cast_bf16_to_f32 (ABI v6, the way back from a bf16 gradient bucket): exact widening, ragged tail, round trip with sa_cast_f32_to_bf16."""
    for n in (8, 4096, 1000003):
        x = torch.randn(n, device=dev) * 3.0
        h = ops.cast_bf16(x)
        assert torch.equal(h, x.to(torch.bfloat16))
        y = torch.full((n,), 7.0, device=dev)
        ops.cast_f32_from_bf16(h, y)
        assert torch.equal(y, h.float())
    with pytest.raises(ValueError):
        ops.cast_f32_from_bf16(h, torch.empty(5, device=dev))


def test_partial_sum_workspaces_are_per_stream(dev):
    """the block partials of sa_bt_loss_grad / sa_mae_recon_loss_fwd / sa_mae_unshuffle_bwd are the CALLER's workspace (ABI v6),
    and `ops` keeps one per stream: loss launches issued back to back on two streams (the producer kernel of one may run between the
    producer and the adding kernel of the other) each get their own sums."""
    D = 256
    g = torch.Generator(device=dev).manual_seed(3)
    cs = [torch.randn(D, D, device=dev, generator=g) * (k + 1) for k in range(2)]
    ref = []
    for c in cs:
        off = c - torch.diag(torch.diag(c))
        ref.append(float(((torch.diag(c) - 1) ** 2).sum() + 0.005 * (off ** 2).sum()))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    losses = [torch.zeros(1, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for it in range(50):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                ops.bt_loss_grad(cs[k], 1.0, 0.005, False, losses[k])
        torch.cuda.synchronize()
        for k in range(2):
            assert abs(float(losses[k]) - ref[k]) <= 2e-5 * abs(ref[k]), (it, k, float(losses[k]), ref[k])
    keys = [k for k in ops._WORKSPACES if k[1] == "bt_loss"]
    assert len({k[2] for k in keys}) >= 2
