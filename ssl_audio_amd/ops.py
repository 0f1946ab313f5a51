"""Thin torch-tensor wrappers over the C ABI (include/ssl_audio_hip.h).

PyTorch is used for device memory, streams and (elsewhere) torch.distributed only; every arithmetic op on the
hot path goes through libssl_audio_hip.so.  All calls are enqueued on torch's current HIP stream.
"""
import ctypes as C

import os

import torch

from . import _lib
from ._lib import SaGemmArgs, check, lib

BF16 = torch.bfloat16
F32 = torch.float32
# Weight gradients are split-K sums.  By default every K slice stores its partial tile in a workspace and a second launch adds the slices
# in slice order: bit-reproducible from run to run, and measured 2 % FASTER than fp32 atomics on the ViT-B shapes (266 vs 273 us on the qkv
# weight gradient: 16 M four-byte atomics cost more than writing and re-reading 64 MB).  SA_DETERMINISTIC=0 selects the atomic form.
DETERMINISTIC_WGRAD = os.environ.get("SA_DETERMINISTIC", "1") != "0"
GEMM_PROFILE = None   # set to a list by bench.py to collect (start event, end event, flops, layout) per GEMM launch
STREAM_PROFILE = None  # set to a dict by bench.py: kernel name -> [(start event, end event, algorithmic bytes)] for the HBM-bound front kernels


def _timed(name, nbytes, launch, flops=0.0, mfma_flops=0.0):
    """bench.py's roofline leg for the HBM-bound kernels: HIP events on the launch stream around one launch
    (flops: fp32 vector work of the launch, for a kernel whose arithmetic outweighs its bytes; mfma_flops: bf16 matrix work, for the
    attention kernels, which the bench line prices against both roofs)."""
    if STREAM_PROFILE is None:
        launch()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    STREAM_PROFILE.setdefault(name, []).append((e0, e1, float(nbytes), float(flops), float(mfma_flops)))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _req(t, dtype, name):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor (ssl_audio_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _rows(t, name):
    """2-D row-major view info: (rows, cols, leading dim)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(A, B, *, a_kmajor=True, b_kmajor=True, alpha=1.0, bias=None, act=0, aux_in=None, aux_out=None, residual=None,
         res_mod=0, out_f32=None, out_bf16=None, row_group=0, split_k=1, accumulate=False, tile256=False, colsum_out=None,
         asum_out=None, asum_skip_lo=0, asum_skip_hi=0):
    """D = epilogue(alpha * A.B).  A: [M,K] (a_kmajor) or [K,M]; B: [N,K] (b_kmajor, nn.Linear weight) or [K,N].
    asum_out (streaming split-K weight gradients only): [M] fp32, += sum_k A[k][m] outside rows [asum_skip_lo, asum_skip_hi) -- the bias
    gradient of the Linear, out of the dY tiles the kernel holds anyway."""
    _req(A, BF16, "A"), _req(B, BF16, "B")
    ar, ac, lda = _rows(A, "A")
    br, bc, ldb = _rows(B, "B")
    M, K = (ar, ac) if a_kmajor else (ac, ar)
    N, Kb = (br, bc) if b_kmajor else (bc, br)
    if K != Kb:
        raise ValueError(f"gemm: inner dims differ ({K} vs {Kb})")
    a = SaGemmArgs()
    a.A, a.lda, a.a_kmajor = A.data_ptr(), lda, int(a_kmajor)
    a.B, a.ldb, a.b_kmajor = B.data_ptr(), ldb, int(b_kmajor)
    a.M, a.N, a.K, a.alpha = M, N, K, float(alpha)
    a.bias = _req(bias, F32, "bias").data_ptr() if bias is not None else None
    a.act = act
    if aux_in is not None:
        a.aux_in, a.ldaux = _req(aux_in, BF16, "aux_in").data_ptr(), _rows(aux_in, "aux_in")[2]
    if aux_out is not None:
        a.aux_out, a.ldaux = _req(aux_out, BF16, "aux_out").data_ptr(), _rows(aux_out, "aux_out")[2]
    if residual is not None:
        a.residual, a.ldr = _req(residual, F32, "residual").data_ptr(), _rows(residual, "residual")[2]
    a.res_mod = res_mod
    if out_f32 is not None:
        a.out_f32, a.ldo_f32 = _req(out_f32, F32, "out_f32").data_ptr(), _rows(out_f32, "out_f32")[2]
    if out_bf16 is not None:
        a.out_bf16, a.ldo_bf16 = _req(out_bf16, BF16, "out_bf16").data_ptr(), _rows(out_bf16, "out_bf16")[2]
    a.row_group, a.split_k, a.accumulate, a.tile256 = row_group, split_k, int(accumulate), int(tile256)
    if split_k > 1 and DETERMINISTIC_WGRAD:
        if N % 4 == 0:
            a.splitk_ws = _workspace(lib().sa_gemm_splitk_workspace_bytes(M, N, split_k), A.device, "gemm_splitk").data_ptr()
        elif N not in _NONDET_WARNED:                # the slice-ordered reduce works on float4 columns
            _NONDET_WARNED.add(N)
            import warnings
            warnings.warn(f"sa_gemm_bf16: split-K output with N = {N} (not a multiple of 4) falls back to fp32 atomics: this weight "
                          "gradient is summed in a run-dependent order (reproducible to fp32 rounding, not bit for bit)")
    if asum_out is not None:
        a.asum_out, a.asum_skip_lo, a.asum_skip_hi = _req(asum_out, F32, "asum_out").data_ptr(), int(asum_skip_lo), int(asum_skip_hi)
        if asum_out.numel() < M or not asum_out.is_contiguous():
            raise ValueError("gemm: asum_out must be a contiguous fp32 vector of at least M elements")
        if a.splitk_ws:
            a.asum_ws = _workspace(4 * split_k * M, A.device, "gemm_asum").data_ptr()
    if colsum_out is not None:          # colsum_out[n] += sum_m (fp32 epilogue result)[m][n], through a scratch of per-64-row partials
        ws = _workspace(lib().sa_gemm_colsum_workspace_bytes(M, N), A.device, "gemm_colsum")
        a.colsum_out, a.colsum_ws = _req(colsum_out, F32, "colsum_out").data_ptr(), ws.data_ptr()
    if GEMM_PROFILE is None:
        check(lib().sa_gemm_bf16(C.byref(a), _stream()), "sa_gemm_bf16")
        return
    # bench.py's roofline leg: HIP events on the launch stream around this launch (flops = 2*M*N*K, algorithmic)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().sa_gemm_bf16(C.byref(a), _stream()), "sa_gemm_bf16")
    e1.record()
    kind = ("NT" if b_kmajor else "NN") if a_kmajor else ("TT" if b_kmajor else "TN")
    epi3 = (residual is not None and not res_mod and not row_group and out_f32 is not None and out_bf16 is None and act == 0 and aux_in is None
            and aux_out is None and colsum_out is None and not accumulate and N % 64 == 0)
    epi1 = (out_bf16 is not None and out_f32 is None and residual is None and act == 0 and aux_in is None and aux_out is None and colsum_out is None
            and not accumulate and not row_group and N % 64 == 0)
    kname = gemm_kernel_family(M, N, K, a_kmajor, b_kmajor, split_k, tile256, epi3, epi1)
    nbytes = 2.0 * (M * K + N * K) + (4.0 * M * N if out_f32 is not None else 0.0) + (2.0 * M * N if out_bf16 is not None else 0.0) \
        + (2.0 * M * N if (aux_in is not None or aux_out is not None) else 0.0) + (4.0 * M * N if residual is not None and not res_mod else 0.0)
    # the epilogue specialisation the launch compiles to (mirror of p.epi_kind in gemm_bf16.hip): with it the label is ONE rocprofv3 symbol
    # (template instance), not a kernel family -- bench.py reports both groupings
    bf16_only = out_bf16 is not None and out_f32 is None and residual is None
    plain = split_k == 1 and not row_group and not res_mod and not accumulate and N % 64 == 0
    epi = 0
    if plain and bf16_only and act == 0 and aux_in is None and aux_out is None and colsum_out is None:
        epi = 1
    elif plain and bf16_only and act == 3 and aux_out is not None and colsum_out is None:
        epi = 6
    elif plain and bf16_only and act == 1 and aux_out is None and aux_in is None and colsum_out is None:
        epi = 8
    elif plain and epi3:
        epi = 3
    elif plain and bf16_only and act in (2, 4) and aux_in is not None and bias is None:
        epi = 4 if act == 2 else 5
    GEMM_PROFILE.append((e0, e1, 2.0 * M * N * K, kind + ("/splitk" if split_k > 1 else ""), nbytes, kname, f"{kname}<{kind}, epilogue {epi}>"))


_WORKSPACES = {}
_NONDET_WARNED = set()


def _workspace(nbytes, device, tag):
    """Grow-only scratch buffer per (device, tag, STREAM): the launches that share one are ordered on the stream they were issued on, so
    they can reuse it (no per-launch allocation: the step stays graph-capturable and free of allocator calls), while launches on another
    stream -- a second trainer, loss terms moved to side streams -- get their own and cannot overwrite each other's partial sums between a
    producer kernel and the kernel that adds them.  Threads sharing ONE stream must serialise their launches themselves,
    as for any stream-ordered work."""
    key = (device, tag, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(nbytes // 4, 1), dtype=F32, device=device)
        _WORKSPACES[key] = ws
    return ws


CU_BUDGET = None      # CUs the big GEMM grids may use (None = the whole device); see set_cu_budget


def set_cu_budget(cus):
    """Reserve CUs for concurrently running collective kernels: persistent GEMM grids and the split-K heuristic use `cus` CUs."""
    global CU_BUDGET
    check(lib().sa_set_cu_budget(int(cus or 0)), "sa_set_cu_budget")
    CU_BUDGET = int(cus) if cus else None


def set_dynamic_tiles(on):
    """Dynamic tile hand-out in the persistent GEMM kernels (include/ssl_audio_hip.h: sa_set_dynamic_tiles)."""
    check(lib().sa_set_dynamic_tiles(int(bool(on))), "sa_set_dynamic_tiles")


def gemm_wgrad_group(dY, X, out, split_k, tile=192, asum_out=None, asum_index=-1, asum_skip_lo=0, asum_skip_hi=0):
    """dY[i] [rows, N_i] bf16, X[i] [rows, K_i] bf16, out[i] [N_i, K_i] fp32, all over the same rows: out_i += dY_i^T X_i for the whole
    group in ONE pair of launches of the streaming split-K kernel (include/ssl_audio_hip.h: sa_gemm_wgrad_group; tile 192 for narrow
    outputs, 256 for wide ones) -- the four weight gradients of a transformer block at backward (models/mae.py:106-129,149-163)."""
    if tile not in (192, 256):
        raise ValueError("gemm_wgrad_group: tile must be 192 or 256")
    if asum_out is not None and not 0 <= asum_index < len(dY):
        raise ValueError("gemm_wgrad_group: asum_index must name the product whose dY column sums go to asum_out")
    jobs = list(zip(dY, X, out))
    n = len(jobs)
    arr = (SaGemmArgs * n)()
    rows = jobs[0][0].shape[0]
    sizes = [lib().sa_gemm_splitk_workspace_bytes(dY.shape[1], X.shape[1], split_k) for dY, X, _ in jobs]
    det = DETERMINISTIC_WGRAD and all(X.shape[1] % 4 == 0 for _, X, _ in jobs)
    ws = _workspace(sum((s + 255) // 256 * 256 for s in sizes), jobs[0][0].device, "gemm_splitk_group") if det else None
    off, flops, nbytes = 0, 0.0, 0.0
    for idx, (a, (dY, X, out), size) in enumerate(zip(arr, jobs, sizes)):
        _req(dY, BF16, "dY"), _req(X, BF16, "X"), _req(out, F32, "out")
        (ra, M, lda), (rb, N, ldb) = _rows(dY, "dY"), _rows(X, "X")
        if ra != rows or rb != rows or tuple(out.shape) != (M, N):
            raise ValueError(f"gemm_wgrad_group: dY {tuple(dY.shape)}, X {tuple(X.shape)}, out {tuple(out.shape)} do not form out = dY^T X over {rows} rows")
        a.A, a.lda, a.a_kmajor = dY.data_ptr(), lda, 0
        a.B, a.ldb, a.b_kmajor = X.data_ptr(), ldb, 0
        a.M, a.N, a.K, a.alpha = M, N, rows, 1.0
        a.out_f32, a.ldo_f32 = out.data_ptr(), _rows(out, "out")[2]
        a.split_k, a.tile256 = split_k, (2 if tile == 192 else 1)
        if det:
            a.splitk_ws = ws.data_ptr() + off
            off += (size + 255) // 256 * 256
        if asum_out is not None and idx == asum_index:
            # (product asum_index also leaves asum_out[m] += sum_rows dY[row][m] outside [asum_skip_lo, asum_skip_hi): its Linear's bias gradient)
            if _req(asum_out, F32, "asum_out").numel() < M or not asum_out.is_contiguous():
                raise ValueError("gemm_wgrad_group: asum_out must be a contiguous fp32 vector of at least M elements")
            a.asum_out, a.asum_skip_lo, a.asum_skip_hi = asum_out.data_ptr(), int(asum_skip_lo), int(asum_skip_hi)
            if det:
                a.asum_ws = _workspace(4 * split_k * M, dY.device, "gemm_asum").data_ptr()
        flops += 2.0 * M * N * rows
        nbytes += 2.0 * (M + N) * rows + 4.0 * M * N
    if GEMM_PROFILE is None:
        check(lib().sa_gemm_wgrad_group(arr, n, _stream()), "sa_gemm_wgrad_group")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().sa_gemm_wgrad_group(arr, n, _stream()), "sa_gemm_wgrad_group")
    e1.record()
    kname = "gemm_tn_stream_kernel" if tile == 192 else "gemm_tn_stream256_kernel"
    GEMM_PROFILE.append((e0, e1, flops, "TN/splitk", nbytes, kname, f"{kname}<group of {n}>"))


STREAM256 = os.environ.get("SA_GEMM_WGRAD_STREAM256", "1") != "0"      # (mirror of gemm_dispatch's default for TN split-K on the 256 tile)


def gemm_kernel_family(M, N, K, a_kmajor, b_kmajor, split_k, tile256, epi3=False, epi1=False):
    """Which device kernel sa_gemm_bf16 dispatches to by default (mirror of the selection in gemm_bf16.hip, no SA_GEMM_TILE
    override); used to label bench.py's per-launch timings with the names rocprofv3 reports.  epi3: the launch has the compact
    bias + residual -> fp32 epilogue (proj / fc2 forward), which the forward layout runs on the phased kernel; epi1: the compact
    (bias ->) bf16 epilogue, phased too once the reduction is long (K >= 1536: the fc1 / qkv data gradients)."""
    if split_k > 1:
        if int(tile256) == 2:
            return "gemm_tn_stream_kernel"
        if tile256:
            return "gemm_tn_stream256_kernel" if (not a_kmajor and not b_kmajor and STREAM256) else "gemm256_kernel<split-K>"
        return "gemm_kernel<split-K>"
    big = M >= 1024 and N >= 192 and ((M + 255) // 256) * ((N + 255) // 256) >= 128
    if not big:
        return "gemm_kernel"
    if a_kmajor and not b_kmajor:
        return "gemm256_ring_kernel"
    phased = a_kmajor and b_kmajor and K >= 128 and (epi3 or (epi1 and (K >= 1536 or (N <= 256 and K >= 512))))
    return "gemm256_phase_kernel" if phased else "gemm256_persist_kernel"


def pick_split_k(M, N, K, cu_count=None, tile=128, tiles=None):
    """Split the reduction of a wgrad-shaped GEMM (few output tiles, long K) until ~2 waves of workgroups exist.  `tiles`: the tile
    count of a whole group of products over the same K (gemm_wgrad_group) instead of the one [M, N] output's."""
    if cu_count is None:
        cu_count = CU_BUDGET or 256
    if tiles is None:
        tiles = ((M + tile - 1) // tile) * ((N + tile - 1) // tile)
    ksteps = (K + 63) // 64
    slots = (2 if tile == 128 else 1) * cu_count       # the 128^2 kernel runs two workgroups per CU, the 256^2 and 192^2 ones one
    if tiles >= slots or ksteps < 16:
        return 1
    # time ~ (rounds of resident workgroups) x (K-iterations per workgroup + fill/epilogue): a split that leaves the last
    # round nearly empty (e.g. 576 workgroups on 512 slots) costs a whole extra round, so search instead of doubling
    best, best_cost = 1, None
    for s in range(1, min(int(os.environ.get("SA_SPLIT_MAX", "256")) if tile == 192 else 128, ksteps // 4) + 1):
        rounds = -(-tiles * s // slots)
        cost = rounds * (-(-ksteps // s) + 8)
        if best_cost is None or cost < best_cost:
            best, best_cost = s, cost
    return best


def transpose_bf16(src, dst=None):
    """dst [C, R] = src [R, C]^T (bf16): the k-contiguous copy of a Linear weight for its data-gradient GEMM."""
    R, Cn = _req(src, BF16, "src").shape
    if not src.is_contiguous():
        raise ValueError("transpose_bf16: src must be contiguous")
    if dst is None:
        dst = torch.empty(Cn, R, dtype=BF16, device=src.device)
    check(lib().sa_transpose_bf16(_p(src), R, Cn, _p(_req(dst, BF16, "dst")), _stream()), "sa_transpose_bf16")
    return dst


def transpose_bf16_batch(desc, n_tiles):
    """desc: int64 [n, 4] on the device ({src ptr, dst ptr, R | C << 32, first tile}); one launch transposes all n matrices."""
    if desc.dtype != torch.int64 or desc.dim() != 2 or desc.shape[1] != 4 or not desc.is_cuda or not desc.is_contiguous():
        raise ValueError("transpose_bf16_batch: desc must be a contiguous int64 [n, 4] GPU tensor")
    check(lib().sa_transpose_bf16_batch(_p(desc), desc.shape[0], int(n_tiles), _stream()), "sa_transpose_bf16_batch")


def cast_bf16(src, dst=None):
    _req(src, F32, "src")
    src = src.contiguous()
    if dst is None:
        dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    check(lib().sa_cast_f32_to_bf16(_p(src), _p(dst), src.numel(), _stream()), "sa_cast_f32_to_bf16")
    return dst


def cast_f32_from_bf16(src, dst):
    """dst (fp32) = src (bf16), flat: the way back from a bf16 gradient bucket (include/ssl_audio_hip.h: sa_cast_bf16_to_f32)."""
    _req(src, BF16, "src"); _req(dst, F32, "dst")
    if src.numel() != dst.numel() or not (src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("cast_f32_from_bf16: src and dst must be contiguous and of equal size")
    check(lib().sa_cast_bf16_to_f32(_p(src), _p(dst), src.numel(), _stream()), "sa_cast_bf16_to_f32")
    return dst


def colsum_bf16(x, out, accumulate=False, n_ranges=1, range_stride=0):
    """out[n] (+)= sum_m x[m][n].  n_ranges > 1: x is the first column range (a [M, N] view) of a wider matrix; the ranges z * range_stride
    columns further on are summed into out + z * range_stride by the same launches."""
    M, N, ld = _rows(_req(x, BF16, "x"), "x")
    # SA_DETERMINISTIC (default on): slab partials through a workspace, added in slab order -- no float atomics, bit-reproducible
    ws = _p(_workspace(n_ranges * lib().sa_colsum_workspace_bytes(M, N), x.device, "colsum")) if DETERMINISTIC_WGRAD else None
    check(lib().sa_colsum_bf16(_p(x), ld, M, N, _p(_req(out, F32, "out")), int(accumulate), ws, int(n_ranges), int(range_stride), _stream()),
          "sa_colsum_bf16")
    return out


def colsum_qv(dqkv, d, gq, gv):
    """q and v bias gradients (models/mae.py:125-128: k's bias is fixed at zero) out of the packed dqkv [M, 3 d]: ONE pair of launches when
    the two gradient buffers sit 2 d apart like the columns do (the flat [q | 0 | v] layout of train.FlatState), else two."""
    if gv.data_ptr() == gq.data_ptr() + 8 * d and gq.is_contiguous() and gv.is_contiguous():
        colsum_bf16(dqkv[:, :d], gq.view(-1), accumulate=True, n_ranges=2, range_stride=2 * d)
    else:
        colsum_bf16(dqkv[:, :d], gq, accumulate=True)
        colsum_bf16(dqkv[:, 2 * d:], gv, accumulate=True)


# ------------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, eps, *, y_bf16=None, y_f32=None, mean=None, rstd=None):
    M, D, ldx = _rows(_req(x, F32, "x"), "x")
    y = y_bf16 if y_bf16 is not None else y_f32
    ldy = _rows(y, "y")[2]
    if y_bf16 is not None and y_f32 is not None and _rows(y_f32, "y_f32")[2] != ldy:
        raise ValueError("layernorm_fwd: both outputs must share a leading dimension")
    # algorithmic bytes: x read once (4 B), each output written once
    _timed("ln_fwd_kernel", float(M) * D * (4 + (2 if y_bf16 is not None else 0) + (4 if y_f32 is not None else 0)), lambda: check(
        lib().sa_layernorm_fwd(_p(x), ldx, _p(gamma), _p(beta), _p(y_bf16), _p(y_f32), ldy, _p(mean), _p(rstd), M, D, float(eps), _stream()),
        "sa_layernorm_fwd"))


def layernorm_bwd(dy, x, gamma, mean, rstd, *, dres=None, dx_f32=None, dx_bf16=None, dgamma=None, dbeta=None, dxsum=None):
    M, D, ldx = _rows(_req(x, F32, "x"), "x")
    lddy = _rows(dy, "dy")[2]
    dxo = dx_f32 if dx_f32 is not None else dx_bf16
    lddx = _rows(dxo, "dx")[2]
    lddres = _rows(dres, "dres")[2] if dres is not None else 0
    ws = None
    if dgamma is not None or dbeta is not None or dxsum is not None:      # scratch for the two-stage column reduction
        ws = _workspace(lib().sa_layernorm_bwd_workspace_bytes(M, D), x.device, "ln_bwd")
    nbytes = float(M) * D * (4 + (2 if dy.dtype == BF16 else 4) + (4 if dres is not None else 0) + (4 if dx_f32 is not None else 0) + (2 if dx_bf16 is not None else 0))
    _timed("ln_bwd_kernel", nbytes, lambda: check(
        lib().sa_layernorm_bwd(_p(dy), int(dy.dtype == BF16), lddy, _p(x), ldx, _p(gamma), _p(mean), _p(rstd), _p(dres), lddres,
                               _p(dx_f32), _p(dx_bf16), lddx, _p(dgamma), _p(dbeta), _p(dxsum), _p(ws), M, D, _stream()), "sa_layernorm_bwd"))


# ------------------------------------------------------------------------------------------------ attention
def attention_fwd(qkv, H, N, scale, out, lse=None, n_query=0):
    rows, w, ld = _rows(_req(qkv, BF16, "qkv"), "qkv")
    Cdim = w // 3
    nq = N if not n_query or n_query > N else int(n_query)
    # algorithmic bytes: q (the queried rows), k, v read once, the output rows written once (+ lse); MFMA flops 4 nq N 64 per (sequence, head)
    _timed("attn_fwd_kernel", (rows // N) * (2.0 * Cdim * (2 * N + 2 * nq) + 4.0 * H * nq), lambda: check(
        lib().sa_attention_fwd(_p(qkv), rows, ld, Cdim, H, N, int(n_query), float(scale), _p(_req(out, BF16, "out")),
                               _rows(out, "out")[2], _p(lse), _stream()), "sa_attention_fwd"), mfma_flops=4.0 * (rows // N) * H * nq * N * 64)


def attention_bwd(qkv, H, N, scale, out, dout, lse, dqkv, n_query=0):
    rows, w, ld = _rows(_req(qkv, BF16, "qkv"), "qkv")
    Cdim = w // 3
    if _rows(dqkv, "dqkv")[2] != ld or _rows(dout, "dout")[2] != _rows(out, "out")[2]:
        raise ValueError("attention_bwd: dqkv/qkv and dout/out must share leading dimensions")
    nq = N if not n_query or n_query > N else int(n_query)
    # algorithmic bytes: qkv, out, dout read once, dqkv written once; MFMA flops 2.5 x the forward's (two passes recompute S and dP)
    _timed("attn_bwd_kernel", (rows // N) * 2.0 * Cdim * (3 * N + 2 * nq + 3 * N), lambda: check(
        lib().sa_attention_bwd(_p(qkv), rows, ld, Cdim, H, N, int(n_query), float(scale), _p(out), _p(_req(dout, BF16, "dout")), _rows(out, "out")[2],
                               _p(_req(lse, F32, "lse")), _p(_req(dqkv, BF16, "dqkv")), _stream()), "sa_attention_bwd"),
           mfma_flops=10.0 * (rows // N) * H * nq * N * 64)


# ------------------------------------------------------------------------------------------------ BatchNorm pieces
def bn_colstats(x, mean, m2):
    B, Cn, ld = _rows(_req(x, F32, "x"), "x")
    check(lib().sa_bn_colstats(_p(x), ld, B, Cn, _p(mean), _p(m2), _stream()), "sa_bn_colstats")


def bn_apply(x, mean, rstd, gamma=None, beta=None, relu=False, *, y_f32=None, y_bf16=None):
    B, Cn, ld = _rows(_req(x, F32, "x"), "x")
    y = y_f32 if y_f32 is not None else y_bf16
    check(lib().sa_bn_apply(_p(x), ld, B, Cn, _p(mean), _p(rstd), _p(gamma), _p(beta), int(relu), _p(y_f32), _p(y_bf16),
                            _rows(y, "y")[2], _stream()), "sa_bn_apply")


def bn_bwd_stats(dy, x, mean, rstd, gamma, beta, relu, s1, s2):
    B, Cn, ld = _rows(_req(x, F32, "x"), "x")
    check(lib().sa_bn_bwd_stats(_p(dy), int(dy.dtype == BF16), _rows(dy, "dy")[2], _p(x), ld, B, Cn, _p(mean), _p(rstd), _p(gamma),
                                _p(beta), int(relu), _p(s1), _p(s2), _stream()), "sa_bn_bwd_stats")


def bn_finalize(stats, rows_per_rank, eps, momentum, mean, rstd, running_mean=None, running_var=None):
    """stats: [W, 2, C] per-rank (mean, M2); may be a strided slice [:, k] of a packed [W, n, 2, C] all-gather."""
    Wn, two, Cn = stats.shape
    if two != 2 or stats.stride(2) != 1 or stats.stride(1) != Cn:
        raise ValueError("bn_finalize: stats must be [W, 2, C] with contiguous (mean, M2) rows per rank")
    check(lib().sa_bn_finalize(_p(_req(stats, F32, "stats")), stats.stride(0) if Wn > 1 else 0, Wn, rows_per_rank, Cn, float(eps), float(momentum),
                               _p(mean), _p(rstd), _p(running_mean), _p(running_var), _stream()), "sa_bn_finalize")


def bn_bwd_apply(dy, x, mean, rstd, gamma, beta, relu, s1, s2, inv_n, *, out_scale=None, dx_f32=None, dx_bf16=None):
    B, Cn, ld = _rows(_req(x, F32, "x"), "x")
    dxo = dx_f32 if dx_f32 is not None else dx_bf16
    check(lib().sa_bn_bwd_apply(_p(dy), int(dy.dtype == BF16), _rows(dy, "dy")[2], _p(x), ld, B, Cn, _p(mean), _p(rstd), _p(gamma),
                                _p(beta), int(relu), _p(s1), _p(s2), float(inv_n), _p(out_scale), _p(dx_f32), _p(dx_bf16), _rows(dxo, "dx")[2],
                                _stream()), "sa_bn_bwd_apply")


# ------------------------------------------------------------------------------------------------ loss pieces
def matmul_f32(A, B, out, *, trans_a=False, trans_b=False, alpha=1.0):
    """out[M,N] = alpha * op(A) @ op(B) in exact fp32 (MFMA f32)."""
    _req(A, F32, "A"), _req(B, F32, "B"), _req(out, F32, "out")
    sam, sak = (A.stride(1), A.stride(0)) if trans_a else (A.stride(0), A.stride(1))
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    sbk, sbn = (B.stride(1), B.stride(0)) if trans_b else (B.stride(0), B.stride(1))
    Kb, N = (B.shape[1], B.shape[0]) if trans_b else (B.shape[0], B.shape[1])
    if K != Kb or tuple(out.shape) != (M, N) or out.stride(1) != 1:
        raise ValueError("matmul_f32: shape mismatch")
    check(lib().sa_matmul_f32(_p(A), sam, sak, _p(B), sbk, sbn, _p(out), out.stride(0), M, N, K, float(alpha), _stream()), "sa_matmul_f32")
    return out


def bt_loss_grad(c, alpha, lmbda, hsic, loss, G=None):
    D = c.shape[0]
    ws = _workspace(lib().sa_bt_loss_workspace_bytes(), c.device, "bt_loss")
    check(lib().sa_bt_loss_grad(_p(_req(c, F32, "c")), D, float(alpha), float(lmbda), int(bool(hsic)), _p(loss), _p(G), _p(ws), _stream()),
          "sa_bt_loss_grad")


def bt_stats2(z1, z2, stats):
    B, D = z1.shape
    check(lib().sa_bt_stats2(_p(_req(z1, F32, "z1")), _p(_req(z2, F32, "z2")), z1.stride(0), B, D, _p(_req(stats, F32, "stats")), _stream()),
          "sa_bt_stats2")


def bt_corr(z1, z2, all_stats, eps, momentum, inv_n, mean, rstd, running_mean, running_var, z1n, z2n, c):
    B, D = z1.shape
    W = all_stats.shape[0]
    if z1.stride(0) != z2.stride(0) or tuple(all_stats.shape[1:]) != (2, 2, D) or not all_stats.is_contiguous():
        raise ValueError("bt_corr: z1 / z2 must share a row stride, all_stats must be a contiguous [W, 2, 2, D]")
    check(lib().sa_bt_corr(_p(_req(z1, F32, "z1")), _p(_req(z2, F32, "z2")), z1.stride(0), B, D, _p(_req(all_stats, F32, "all_stats")), W, float(eps),
                           float(momentum), float(inv_n), _p(mean), _p(rstd), _p(running_mean), _p(running_var), _p(z1n), _p(z2n), _p(c), _stream()),
          "sa_bt_corr")


def bt_bwd_products(z1n, z2n, G, inv_n, dzn, sums):
    B, D = z1n.shape
    check(lib().sa_bt_bwd_products(_p(_req(z1n, F32, "z1n")), _p(_req(z2n, F32, "z2n")), B, D, _p(_req(G, F32, "G")), float(inv_n), _p(dzn), _p(sums),
                                   _stream()), "sa_bt_bwd_products")


def bt_bwd_apply(z1n, z2n, rstd, dzn, sums, inv_n, out_scale, dz1, dz2):
    B, D = z1n.shape
    check(lib().sa_bt_bwd_apply(_p(_req(z1n, F32, "z1n")), _p(_req(z2n, F32, "z2n")), B, D, _p(rstd), _p(dzn), _p(sums), float(inv_n), _p(out_scale),
                                _p(dz1), _p(dz2), _stream()), "sa_bt_bwd_apply")


# ------------------------------------------------------------------------------------------------ optimiser
def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, p_bf16=None):
    n = p.numel()
    # algorithmic bytes per parameter: p, g, m, v read (16 B), p, m, v written (12 B), bf16 copy written (2 B)
    _timed("adamw_vec4_kernel", float(n) * (28 + (2 if p_bf16 is not None else 0)), lambda: check(
        lib().sa_adamw_step(_p(p), _p(g), _p(m), _p(v), n, float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step),
                            float(grad_scale), _p(p_bf16), _stream()), "sa_adamw_step"))


def adamw_step_dev(p, g, m, v, hyper3, beta1, beta2, eps, wd, grad_scale=1.0, p_bf16=None, skip_flag=None):
    """adamw_step with {lr, 1/(1-b1^t), 1/sqrt(1-b2^t)} read from the device vector hyper3 (graph-replayable) and an optional gate word."""
    n = p.numel()
    _timed("adamw_vec4_kernel", float(n) * (28 + (2 if p_bf16 is not None else 0)), lambda: check(
        lib().sa_adamw_step_dev(_p(p), _p(g), _p(m), _p(v), n, _p(_req(hyper3, F32, "hyper3")), float(beta1), float(beta2), float(eps), float(wd),
                                float(grad_scale), _p(p_bf16), _p(skip_flag), _stream()), "sa_adamw_step_dev"))


def ema_update_gated(target, online, beta, skip_flag):
    check(lib().sa_ema_update_gated(_p(target), _p(online), target.numel(), float(beta), _p(skip_flag), _stream()), "sa_ema_update_gated")


def lars_step(p, g, mu, lr, wd, momentum, eta, adapt, scratch2=None, p_bf16=None):
    check(lib().sa_lars_step(_p(_req(p, F32, "p")), _p(_req(g, F32, "g")), _p(_req(mu, F32, "mu")), p.numel(), float(lr), float(wd), float(momentum),
                             float(eta), int(bool(adapt)), _p(scratch2), _p(p_bf16), _stream()), "sa_lars_step")


def ema_update(target, online, beta):
    check(lib().sa_ema_update(_p(target), _p(online), target.numel(), float(beta), _stream()), "sa_ema_update")


def axpy(y, x, a=1.0):
    check(lib().sa_axpy_f32(_p(_req(y, F32, "y")), _p(_req(x, F32, "x")), y.numel(), float(a), _stream()), "sa_axpy_f32")


def count_nonfinite(x, flag):
    """flag[0] (int32) += number of non-finite entries of the fp32 tensor x (no host synchronisation)."""
    if flag.dtype != torch.int32:
        raise TypeError("count_nonfinite: flag must be int32")
    check(lib().sa_count_nonfinite(_p(_req(x, F32, "x")), x.numel(), _p(flag), _stream()), "sa_count_nonfinite")


# ------------------------------------------------------------------------------------------------ frontend / augmentation
def logmel_fwd(wave, tables, out, T_out, start, mean, std, hop, starts=None, lengths=None, offsets=None):
    """starts / lengths / offsets: optional int32 device vectors [B] -- per-clip crop start (frames), clip length (samples) and first
    sample inside the row (include/ssl_audio_hip.h, ABI v6); None = the scalar `start`, the row length, 0."""
    B, L = wave.shape
    for name, t in (("starts", starts), ("lengths", lengths), ("offsets", offsets)):
        if t is not None and (_req(t, torch.int32, name).numel() != B or not t.is_contiguous()):
            raise ValueError(f"logmel_fwd: {name} must be a contiguous int32 vector of {B} entries")
    # algorithmic bytes: the waveform read once + the log-mel written once
    _timed("logmel_kernel", 4.0 * B * L + 4.0 * B * 64 * T_out, lambda: check(
        lib().sa_logmel_fwd(_p(_req(wave, F32, "wave")), wave.stride(0), B, L, _p(tables["window"]), _p(tables["twiddle"]),
                            _p(tables["mel_weights"]), _p(tables["mel_lo"]), _p(tables["mel_len"]), _p(_req(out, F32, "out")),
                            out.stride(0), T_out, int(start), _p(starts), _p(lengths), _p(offsets), float(mean), float(std), int(hop), _stream()),
              "sa_logmel_fwd"),
           # per frame: 1024-point FFT (5 N log2 N), window, 513 power bins (3 each), the <= 2 mel bands of every bin (2 x 2 each)
           flops=float(B) * T_out * (5.0 * 1024 * 10 + 1024 + 3 * 513 + 4 * 513))


def augment_views(lms, clip_stride, src_slot, mix_slot, params, out, F_in, T_in, canvas, max_w_ratio, do_fade, noise=None):
    V, F_out, T_out = out.shape[0], out.shape[-2], out.shape[-1]
    if noise is not None and (_req(noise, F32, "noise").numel() != V * F_in * T_in or not noise.is_contiguous()):
        raise ValueError(f"noise must be a contiguous [{V}, {F_in}, {T_in}] tensor of standard-normal draws")
    # algorithmic bytes per view: the clip and its mixup partner (and the noise grid) read once, the view written once
    _timed("augment_kernel", V * 4.0 * ((2 + (noise is not None)) * F_in * T_in + F_out * T_out), lambda: check(
        lib().sa_augment_views(_p(_req(lms, F32, "lms")), clip_stride, _p(src_slot), _p(mix_slot), _p(_req(params, F32, "params")),
                               _p(_req(out, F32, "out")), V, F_in, T_in, canvas[0], canvas[1], F_out, T_out, float(max_w_ratio),
                               int(do_fade), _p(noise), _stream()), "sa_augment_views"))


def normalize_batch(x, y, shift, workspace, eps, stat_div=1.0):
    check(lib().sa_normalize_batch(_p(_req(x, F32, "x")), _p(y), x.numel(), float(shift), _p(workspace), float(eps), float(stat_div), _stream()),
          "sa_normalize_batch")


def patchify_bf16(img, out, ph, pw):
    S, _, F_, T_ = img.shape
    check(lib().sa_patchify_bf16(_p(_req(img, F32, "img")), _p(_req(out, BF16, "out")), S, F_, T_, ph, pw, _stream()), "sa_patchify_bf16")


# ------------------------------------------------------------------------------------------------ token bookkeeping
def fill_cls(x, S, seq_stride, d, cls, pos0):
    check(lib().sa_fill_cls(_p(x), S, seq_stride, d, _p(cls), _p(pos0), _stream()), "sa_fill_cls")


def cls_grad(dx, S, seq_stride, d, dcls):
    check(lib().sa_cls_grad(_p(dx), S, seq_stride, d, _p(dcls), _stream()), "sa_cls_grad")


def gather_rows(src, src_seq_stride, src_row0, idx, dst, dst_seq_stride, dst_row0, S, d):
    check(lib().sa_gather_rows(_p(src), src_seq_stride, src_row0, _p(idx), idx.shape[1], _p(dst), dst_seq_stride, dst_row0, S, d, _stream()),
          "sa_gather_rows")


def scatter_add_rows(src, src_seq_stride, src_row0, idx, dst, dst_seq_stride, dst_row0, S, d):
    check(lib().sa_scatter_add_rows(_p(src), src_seq_stride, src_row0, _p(idx), idx.shape[1], _p(dst), dst_seq_stride, dst_row0, S, d,
                                    _stream()), "sa_scatter_add_rows")


def mix_gaussian_noise(x, normal, lambd, eps, out):
    check(lib().sa_mix_gaussian_noise(_p(_req(x, F32, "x")), _p(_req(normal, F32, "normal")), x.numel(), float(lambd), float(eps),
                                      _p(_req(out, F32, "out")), _stream()), "sa_mix_gaussian_noise")


def running_norm(x, state, n_seen, update, eps, out):
    Cn = x.shape[0]
    check(lib().sa_running_norm(_p(_req(x, F32, "x")), Cn, x.numel() // Cn, _p(_req(state, F32, "state")), int(n_seen), int(bool(update)), float(eps),
                                _p(_req(out, F32, "out")), _stream()), "sa_running_norm")


def token_group_sum(y, row0, G, group_stride, count, scale, out, accumulate=False):
    S, N, d = y.shape
    check(lib().sa_token_group_sum(_p(_req(y, F32, "y")), S, N, d, int(row0), int(G), int(group_stride), int(count), float(scale), int(accumulate),
                                   _p(_req(out, F32, "out")), _stream()), "sa_token_group_sum")


def mean_tokens_fwd(y, out):
    S, N, d = y.shape
    check(lib().sa_mean_tokens_fwd(_p(_req(y, F32, "y")), S, N, d, _p(_req(out, F32, "out")), _stream()), "sa_mean_tokens_fwd")


def mean_tokens_bwd(dout, dy):
    S, N, d = dy.shape
    check(lib().sa_mean_tokens_bwd(_p(_req(dout, F32, "dout")), S, N, d, _p(_req(dy, F32, "dy")), _stream()), "sa_mean_tokens_bwd")


def mae_unshuffle_fwd(x, mask_token, pos, ids_restore, out):
    B, kp1, d = x.shape
    L = ids_restore.shape[1]
    if ids_restore.dtype != torch.int32 or not ids_restore.is_contiguous():
        raise ValueError("mae_unshuffle: ids_restore must be a contiguous int32 [B, L] tensor")
    check(lib().sa_mae_unshuffle_fwd(_p(_req(x, F32, "x")), kp1 - 1, _p(_req(mask_token, F32, "mask_token")), _p(_req(pos, F32, "pos")),
                                     _p(ids_restore), B, L, d, _p(_req(out, F32, "out")), _stream()), "sa_mae_unshuffle_fwd")


def mae_unshuffle_bwd(dout, keep, ids_restore, dx, dmask_token):
    B, Lp1, d = dout.shape
    ws = _workspace(lib().sa_mae_unshuffle_bwd_workspace_bytes(), dout.device, "mae_unshuffle") if dmask_token is not None else None
    check(lib().sa_mae_unshuffle_bwd(_p(_req(dout, F32, "dout")), keep, _p(ids_restore), B, Lp1 - 1, d, _p(_req(dx, F32, "dx")),
                                     _p(dmask_token), _p(ws), _stream()), "sa_mae_unshuffle_bwd")


def mae_recon_loss_fwd(pred, row0, img, mask, ph, pw, acc2, loss, norm_pix=False):
    """pred: contiguous [B, row0 + L, P] (row0 = 1: decoder output with its CLS row, read in place)."""
    B, _, F_, T_ = img.shape
    check(lib().sa_mae_recon_loss_fwd(_p(_req(pred, F32, "pred")), pred.shape[1] * pred.shape[2], row0, _p(_req(img, F32, "img")),
                                      _p(_req(mask, F32, "mask")), B, F_, T_, ph, pw, int(bool(norm_pix)), _p(_req(acc2, F32, "acc2")),
                                      _p(_req(loss, F32, "loss")), _p(_workspace(lib().sa_mae_recon_loss_workspace_bytes(), pred.device, "mae_loss")),
                                      _stream()), "sa_mae_recon_loss_fwd")


def mae_recon_loss_finalize(acc2, loss):
    check(lib().sa_mae_recon_loss_finalize(_p(_req(acc2, F32, "acc2")), _p(_req(loss, F32, "loss")), _stream()), "sa_mae_recon_loss_finalize")


def mae_recon_loss_bwd(pred, row0, img, mask, ph, pw, acc2, gscale, dpred, norm_pix=False):
    B, _, F_, T_ = img.shape
    check(lib().sa_mae_recon_loss_bwd(_p(pred), pred.shape[1] * pred.shape[2], row0, _p(img), _p(mask), _p(acc2), _p(_req(gscale, F32, "gscale")),
                                      B, F_, T_, ph, pw, int(bool(norm_pix)), _p(_req(dpred, F32, "dpred")), _stream()), "sa_mae_recon_loss_bwd")


# ------------------------------------------------------------------------------------------------ convolutional stems (NHWC)
def conv_out_size(n, stride):
    """3x3 kernel, padding 1."""
    return (n - 1) // stride + 1


def conv3x3_c1_fwd(x, w, bias, stride, y):
    """x fp32 [B, 1, H, W] (or [B, H, W]) -> y fp32 [B*Ho*Wo, C_out] (NHWC rows)."""
    B, H, W = x.shape[0], x.shape[-2], x.shape[-1]
    check(lib().sa_conv3x3_c1_fwd(_p(_req(x, F32, "x")), B, H, W, stride[0], stride[1], _p(_req(w, F32, "w")), _p(bias), w.shape[0],
                                  _p(_req(y, F32, "y")), _stream()), "sa_conv3x3_c1_fwd")


def conv3x3_c1_wgrad(x, dy16, stride, dw, dbias=None):
    B, H, W = x.shape[0], x.shape[-2], x.shape[-1]
    check(lib().sa_conv3x3_c1_wgrad(_p(_req(x, F32, "x")), B, H, W, stride[0], stride[1], _p(_req(dy16, BF16, "dy")), dy16.shape[-1],
                                    _p(_req(dw, F32, "dw")), _p(dbias), _stream()), "sa_conv3x3_c1_wgrad")


def im2col3x3(x16, B, H, W, C, stride, out):
    check(lib().sa_im2col3x3_bf16(_p(_req(x16, BF16, "x")), B, H, W, C, stride[0], stride[1], _p(_req(out, BF16, "out")), out.shape[-1], _stream()),
          "sa_im2col3x3_bf16")


def col2im3x3(dP16, B, H, W, C, stride, dx):
    check(lib().sa_col2im3x3_f32(_p(_req(dP16, BF16, "dP")), B, H, W, C, stride[0], stride[1], dP16.shape[-1], _p(_req(dx, F32, "dx")), _stream()),
          "sa_col2im3x3_f32")


def bn_colstats_tall(x, mean, m2):
    M, Cn, ld = _rows(_req(x, F32, "x"), "x")
    ws = _workspace(lib().sa_bn_tall_workspace_bytes(M, Cn), x.device, "bn_tall")
    check(lib().sa_bn_colstats_tall(_p(x), ld, M, Cn, _p(ws), _p(mean), _p(m2), _stream()), "sa_bn_colstats_tall")


def bn_bwd_stats_tall(dy, x, mean, rstd, gamma, beta, relu, s1, s2):
    M, Cn, ld = _rows(_req(x, F32, "x"), "x")
    ws = _workspace(lib().sa_bn_tall_workspace_bytes(M, Cn), x.device, "bn_tall")
    check(lib().sa_bn_bwd_stats_tall(_p(dy), int(dy.dtype == BF16), _rows(dy, "dy")[2], _p(x), ld, M, Cn, _p(mean), _p(rstd), _p(gamma), _p(beta),
                                     int(relu), _p(ws), _p(s1), _p(s2), _stream()), "sa_bn_bwd_stats_tall")


def maxpool2_fwd(x16, B, H, W, C, y16, idx):
    check(lib().sa_maxpool2_fwd(_p(_req(x16, BF16, "x")), B, H, W, C, _p(_req(y16, BF16, "y")), _p(idx), _stream()), "sa_maxpool2_fwd")


def maxpool2_bwd(dy, idx, B, H, W, C, dx):
    check(lib().sa_maxpool2_bwd(_p(_req(dy, F32, "dy")), _p(idx), B, H, W, C, _p(_req(dx, F32, "dx")), _stream()), "sa_maxpool2_bwd")


def relu_mask_fwd(x, keep, scale, *, y_bf16=None, y_f32=None):
    M, Cn, ld = _rows(_req(x, F32, "x"), "x")
    check(lib().sa_relu_mask_fwd(_p(x), ld, M, Cn, _p(keep), float(scale), _p(y_bf16), _rows(y_bf16, "y")[2] if y_bf16 is not None else 0,
                                 _p(y_f32), _rows(y_f32, "y")[2] if y_f32 is not None else 0, _stream()), "sa_relu_mask_fwd")


def relu_mask_bwd(dy, x_pre, keep, scale, dx_bf16):
    M, Cn, ld = _rows(_req(x_pre, F32, "x_pre"), "x_pre")
    check(lib().sa_relu_mask_bwd(_p(_req(dy, F32, "dy")), _rows(dy, "dy")[2], _p(x_pre), ld, M, Cn, _p(keep), float(scale), _p(_req(dx_bf16, BF16, "dx")),
                                 _rows(dx_bf16, "dx")[2], _stream()), "sa_relu_mask_bwd")


def nhwc_to_frames(x16, B, H, W, C, frames_bf16=None, frames_f32=None):
    check(lib().sa_nhwc_to_frames(_p(_req(x16, BF16, "x")), B, H, W, C, _p(frames_bf16), _p(frames_f32),
                                  _rows(frames_f32, "frames_f32")[2] if frames_f32 is not None else 0, _stream()), "sa_nhwc_to_frames")


def frames_to_nhwc(da, db, B, H, W, C, dx):
    check(lib().sa_frames_to_nhwc(_p(da), _rows(da, "da")[2] if da is not None else 0, _p(db), _rows(db, "db")[2] if db is not None else 0, B, H, W, C,
                                  _p(_req(dx, F32, "dx")), _stream()), "sa_frames_to_nhwc")


def meanmax_time_fwd(x, out, arg):
    B, T_, D = x.shape
    check(lib().sa_meanmax_time_fwd(_p(_req(x, F32, "x")), B, T_, D, _p(_req(out, F32, "out")), _p(arg), _stream()), "sa_meanmax_time_fwd")


def meanmax_time_bwd(dout, arg, dx):
    B, T_, D = dx.shape
    check(lib().sa_meanmax_time_bwd(_p(_req(dout, F32, "dout")), _p(arg), B, T_, D, _p(_req(dx, F32, "dx")), _stream()), "sa_meanmax_time_bwd")


def se_fwd(x16, B, L, C, w1, w2, s, h, e, y16):
    """Squeeze-and-excitation gate on a channel-last bf16 map [B * L, C] (include/ssl_audio_hip.h: sa_se_fwd)."""
    check(lib().sa_se_fwd(_p(_req(x16, BF16, "x")), B, L, C, _p(_req(w1, F32, "w1")), _p(_req(w2, F32, "w2")), w1.shape[0], _p(_req(s, F32, "s")),
                          _p(_req(h, F32, "h")), _p(_req(e, F32, "e")), _p(_req(y16, BF16, "y")), _stream()), "sa_se_fwd")


def se_bwd(dy, x16, B, L, C, w1, w2, s, h, e, dx, dw1, dw2):
    ws = _workspace(2 * B * C * 4, dy.device, "se_block")
    check(lib().sa_se_bwd(_p(_req(dy, F32, "dy")), _p(_req(x16, BF16, "x")), B, L, C, _p(_req(w1, F32, "w1")), _p(_req(w2, F32, "w2")), w1.shape[0],
                          _p(s), _p(h), _p(e), _p(_req(dx, F32, "dx")), _p(_req(dw1, F32, "dw1")), _p(_req(dw2, F32, "dw2")), _p(ws), _stream()), "sa_se_bwd")


def device_info():
    name = C.create_string_buffer(128)
    cus = C.c_int32(0)
    check(lib().sa_device_info(name, 128, C.byref(cus)), "sa_device_info")
    return name.value.decode(), cus.value


# ------------------------------------------------------------------------------------------------ ResNet pieces (resnet.hip)
def pool_out_size(n):
    """MaxPool2d(3, stride 2, padding 1)."""
    return (n - 1) // 2 + 1


def maxpool3s2_fwd(x16, B, H, W, C, y16, y32, idx):
    check(lib().sa_maxpool3s2_fwd(_p(_req(x16, BF16, "x")), B, H, W, C, _p(_req(y16, BF16, "y")), _p(y32), _p(idx), _stream()), "sa_maxpool3s2_fwd")


def maxpool3s2_bwd(dy, idx, B, H, W, C, dx):
    check(lib().sa_maxpool3s2_bwd(_p(_req(dy, F32, "dy")), _p(idx), B, H, W, C, _p(_req(dx, F32, "dx")), _stream()), "sa_maxpool3s2_bwd")


def subsample_fwd(x16, B, H, W, C, stride, y16):
    check(lib().sa_subsample_fwd(_p(_req(x16, BF16, "x")), B, H, W, C, stride[0], stride[1], _p(_req(y16, BF16, "y")), _stream()), "sa_subsample_fwd")


def subsample_bwd_add(dy16, B, H, W, C, stride, dx):
    check(lib().sa_subsample_bwd_add(_p(_req(dy16, BF16, "dy")), _rows(dy16, "dy")[2], B, H, W, C, stride[0], stride[1], _p(_req(dx, F32, "dx")),
                                     _stream()), "sa_subsample_bwd_add")


def add_relu_fwd(z, identity, y_f32, y_bf16=None):
    check(lib().sa_add_relu_fwd(_p(_req(z, F32, "z")), _p(_req(identity, F32, "identity")), z.numel(), _p(_req(y_f32, F32, "y")), _p(y_bf16), _stream()),
          "sa_add_relu_fwd")


def relu_bwd(dy, dy2, y, ds):
    check(lib().sa_relu_bwd(_p(_req(dy, F32, "dy")), _p(dy2), _p(_req(y, F32, "y")), y.numel(), _p(_req(ds, F32, "ds")), _stream()), "sa_relu_bwd")


def avgpool_fwd(x, B, L, C, out):
    check(lib().sa_avgpool_fwd(_p(_req(x, F32, "x")), B, L, C, _p(_req(out, F32, "out")), _stream()), "sa_avgpool_fwd")


def avgpool_bwd(dout, B, L, C, dx):
    check(lib().sa_avgpool_bwd(_p(_req(dout, F32, "dout")), B, L, C, _p(_req(dx, F32, "dx")), _stream()), "sa_avgpool_bwd")


# ------------------------------------------------------------------------------------------------ routing through torch.ops
DISPATCH = "direct"          # "direct": ops.<name> IS the ctypes call; "torch_ops": it goes torch.ops.ssl_audio.<name> -> dispatcher -> ctypes call


def route_through_dispatcher(on=True):
    """Make every `ops.<name>(...)` of the engine / trainer go through `torch.ops.ssl_audio.<name>` (custom_ops.py: torch.library operators,
    BASELINE's "driven from Python via PyTorch-ROCm custom ops") instead of calling the ctypes wrapper directly.  Same kernels, same
    bits (tests/test_configs_gpu.py::test_step_through_the_dispatcher_equals_direct); what it costs is host time per launch (the
    dispatcher + the Python-registered implementation: measured in DESIGN.md §6), so "direct" stays the default and
    `SA_DISPATCH=torch_ops` (or bench.py --dispatch torch_ops) selects this route.  Idempotent; `on=False` restores the direct calls."""
    global DISPATCH
    import sys
    from . import custom_ops                     # registers the operators against the ORIGINAL functions (captured at import)
    mod = sys.modules[__name__]
    direct = mod.__dict__.setdefault("_DIRECT_FNS", {})
    if not on:
        for name, fn in direct.items():
            setattr(mod, name, fn)
        DISPATCH = "direct"
        return
    ns = getattr(torch.ops, custom_ops.NAMESPACE)
    for name in custom_ops.SCHEMAS:
        if name in custom_ops._ADAPTERS:         # (the dispatcher signature differs from the ops function's: keep the direct call)
            continue
        direct.setdefault(name, getattr(mod, name))
        op = getattr(ns, name)
        if name in ("transpose_bf16", "cast_bf16"):              # the two that allocate their output when none is given

            def alloc_wrapper(src, dst=None, _op=op, _name=name):
                if dst is None:
                    src = src if _name == "transpose_bf16" else src.contiguous()
                    shape = (src.shape[1], src.shape[0]) if _name == "transpose_bf16" else tuple(src.shape)
                    dst = torch.empty(shape, dtype=BF16, device=src.device)
                _op(src, dst)
                return dst
            setattr(mod, name, alloc_wrapper)
        elif name in ("colsum_bf16", "matmul_f32"):              # ... and the two that hand their output argument back

            def out_wrapper(*args, _op=op, _pos=(1 if name == "colsum_bf16" else 2), **kwargs):
                _op(*args, **kwargs)
                return args[_pos] if len(args) > _pos else kwargs["out"]
            setattr(mod, name, out_wrapper)
        else:
            setattr(mod, name, op)
    DISPATCH = "torch_ops"


if os.environ.get("SA_DISPATCH") == "torch_ops":
    route_through_dispatcher(True)
