"""Explicit forward / backward schedule of the ViT encoder on the HIP kernels.

This is the MI355X-side replacement for autograd tracing through models/mae.py:349-469: a fixed sequence of
C-ABI launches per transformer block with fused epilogues and a hand-planned set of saved activations,

  fwd block:  LN1 -> [qkv GEMM + (q,0,v) bias] -> fused attention -> [proj GEMM + bias + residual]
              -> LN2 -> [fc1 GEMM + bias + GELU (pre-activation kept)] -> [fc2 GEMM + bias + residual]
  bwd block:  mirrors it; every dgrad runs in the forward operand layout (NT) against a transposed bf16 copy of the
              weight (rebuilt once per optimiser step), every wgrad is a TN split-K GEMM whose slices are summed in slice
              order by a second launch (bit-reproducible; SA_DETERMINISTIC=0: fp32 atomics), GELU' -- stored by the fc1
              forward -- multiplied in by the fc2 dgrad epilogue, LayerNorm backward fused with the residual-gradient
              add and emitting the bf16 copy the next GEMMs consume.

Residual stream fp32, GEMM operands / attention bf16, accumulation fp32.  `torch` only provides device
memory here.  The autograd.Function at the bottom exposes the schedule to nn.Module callers.
"""
import os
import weakref

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
# fc1 forward stores GELU'(pre-activation) (act 3) and the fc2 dgrad multiplies by it (act 4); SA_ACT_PAIR=0 falls back to the
# pre-activation form (act 1 / 2) for A/B measurements
_ACT_PAIR = __import__("os").environ.get("SA_ACT_PAIR", "1") != "0"


class _Bf16Cache:
    """bf16 copies of fp32 parameters for the GEMMs, refreshed when the parameter's version changes."""

    def __init__(self):
        self._c = {}
        self._manual = {}
        self._pinned = {}

    def mark_modified(self, p):
        """Call after a HIP kernel rewrote `p` in place (those writes do not bump torch's version counter)."""
        self._manual[id(p)] = self._manual.get(id(p), 0) + 1

    def pin(self, p, w16):
        """`w16` (a view of train.FlatState's bf16 buffer) is kept current by the optimiser kernel, whose in-place writes do not
        touch torch's version counter.  Anything that DOES bump it (load_state_dict, an in-place torch op on the parameter)
        changed the master behind the optimiser's back: the copy is re-cast on the next use."""
        self._pinned[id(p)] = [w16, p.data_ptr(), weakref.ref(p), p._version]
        _forget_when_dead(p, self._pinned, self._manual)

    def get(self, p):
        # id() values are recycled once a parameter dies, so every entry carries a weak reference to ITS parameter
        key = id(p)
        pinned = self._pinned.get(key)
        if pinned is not None and pinned[2]() is p and pinned[1] == p.data_ptr():
            if pinned[3] != p._version:
                ops.cast_bf16(p.detach().reshape(pinned[0].shape), pinned[0])
                pinned[3] = p._version
            return pinned[0]
        ent = self._c.get(key)
        ver = (p._version, self._manual.get(key, 0))
        if ent is None or ent[3]() is not p or ent[0] != ver or ent[1].device != p.device or ent[2] != p.data_ptr():
            w = p.detach()
            w2 = w.reshape(w.shape[0], -1) if w.dim() > 1 else w
            ent = (ver, ops.cast_bf16(w2.contiguous()), p.data_ptr(), weakref.ref(p))
            if key not in self._c:
                _forget_when_dead(p, self._c, self._manual)
            self._c[key] = ent
        return ent[1]


def _forget_when_dead(p, *tables):
    """Entries of the id()-keyed tables die with their parameter: each one pins device memory (a bf16 copy of a weight, a view that
    keeps a whole flat buffer alive), and models are rebuilt many times in one process by tests and sweeps."""
    key = id(p)

    def drop():
        for t in tables:
            t.pop(key, None)
    weakref.finalize(p, drop)


BF16_WEIGHTS = _Bf16Cache()

# Bumped by whoever rewrites parameters with a HIP kernel (train.FlatState's optimiser step): the pinned bf16 copies are refreshed by
# that kernel itself, anything DERIVED from them (the transposed copies below) is rebuilt on its next use.  A flat state registers its own
# counter for its parameters (PARAM_EPOCH: id(parameter) -> that one-element list), so that rewriting ONE state -- BYOL's EMA of the target
# network, right after the online network's optimiser step rebuilt its transposed copies -- does not mark every other state's copies stale
# (45 single-matrix transposes per BYOL step); WEIGHT_EPOCH is the counter of parameters no state owns.
WEIGHT_EPOCH = [0]
PARAM_EPOCH = {}


def register_epoch(params, cell):
    for p in params:
        if id(p) not in PARAM_EPOCH:
            _forget_when_dead(p, PARAM_EPOCH)
        PARAM_EPOCH[id(p)] = cell


def _epoch_of(key):
    return PARAM_EPOCH.get(key, WEIGHT_EPOCH)[0]


class _Bf16TransposedCache:
    """[in, out] bf16 copies of the [out, in] Linear weights.  The data gradient dX = dY W then reads W^T k-contiguous, i.e. runs in
    the forward operand layout (NT) instead of through the transposing LDS reads of the k-strided layout (NN): measured 6-16 % faster on
    the ViT-B shapes (fc2 dgrad 367 -> 312 us), for one transpose of the weights per optimiser step (170 MB of traffic, < 0.1 ms)."""

    def __init__(self):
        self._c = {}

    def get(self, p):
        w16 = BF16_WEIGHTS.get(p)                    # current [out, in] copy (casts if the master changed)
        key = id(p)
        ver = (p._version, BF16_WEIGHTS._manual.get(key, 0), _epoch_of(key), w16.data_ptr())
        ent = self._c.get(key)
        if ent is None or ent[2]() is not p or ent[0] != ver:
            buf = ent[1] if ent is not None and ent[2]() is p and ent[1].shape == (w16.shape[1], w16.shape[0]) and ent[1].device == w16.device else None
            if key not in self._c:
                _forget_when_dead(p, self._c)
            ent = (ver, ops.transpose_bf16(w16.contiguous(), buf), weakref.ref(p))
            self._c[key] = ent
        return ent[1]


    def refresh_all(self, owner=None):
        """Rebuild the transposed copies this cache holds (`owner`: only those of the parameters whose ids it contains -- one flat
        optimiser state's) in one launch (sa_transpose_bf16_batch) -- called by the flat optimiser step
        right after it rewrote the weights, so that the next backward's get() finds all of them fresh instead of issuing one ~5 us
        launch per Linear (48 per ViT step).  Entries created later (first use) are transposed on their own once."""
        live = [(k, e) for k, e in self._c.items() if e[2]() is not None and (owner is None or k in owner)]
        if not live:
            return
        tag = id(owner) if owner is not None else 0
        cache = self.__dict__.setdefault("_batch", {})
        rows, sig, tile0 = [], [], 0
        for k, e in live:
            p = e[2]()
            w16 = BF16_WEIGHTS.get(p)
            if not w16.is_contiguous() or e[1].shape != (w16.shape[1], w16.shape[0]):
                return                                           # (a reshaped weight: leave it to get())
            R, Cn = w16.shape
            rows.append((w16.data_ptr(), e[1].data_ptr(), R | (Cn << 32), tile0))
            sig.append((w16.data_ptr(), e[1].data_ptr(), R, Cn))
            tile0 += ((R + 63) // 64) * ((Cn + 63) // 64)
        sig = tuple(sig)
        if cache.get(tag, (None,))[0] != sig:                    # descriptor table: built once per set of weights
            cache[tag] = (sig, torch.tensor(rows, dtype=torch.int64).to(live[0][1][1].device), tile0)
        ops.transpose_bf16_batch(cache[tag][1], cache[tag][2])
        for k, e in live:
            p = e[2]()
            w16 = BF16_WEIGHTS.get(p)
            self._c[k] = ((p._version, BF16_WEIGHTS._manual.get(k, 0), _epoch_of(k), w16.data_ptr()), e[1], e[2])


BF16_WEIGHTS_T = _Bf16TransposedCache()
DGRAD_NT = os.environ.get("SA_DGRAD_NT", "1") != "0"      # data gradients against the transposed weight copies (0: k-strided weights, NN)


def _dgrad_w(w):
    """(B operand, b_kmajor) of a data-gradient GEMM dX = dY W for the Linear weight w [out, in]."""
    # (a k-contiguous operand needs the reduction length -- here out_features -- to be a multiple of the 64-wide K-tile)
    if DGRAD_NT and w.shape[0] % 64 == 0 and w.shape[1] % 8 == 0:
        return BF16_WEIGHTS_T.get(w), True
    return BF16_WEIGHTS.get(w), False

# id(parameter) -> (weakref(parameter), fp32 buffer, weakref(owner)): the buffer is normally a view into the flat gradient buffer of
# the owning train.FlatState.  When a parameter has a sink, the backward kernels accumulate straight into it and autograd is handed
# None (no per-parameter allocation, no autograd accumulate pass, and the flat buffer is what the gradient all-reduce and the fused
# AdamW consume).
GRAD_SINK = {}
_PENDING_BWD = {}


def register_grad_sink(p, view, owner):
    GRAD_SINK[id(p)] = (weakref.ref(p), view, weakref.ref(owner))
    _forget_when_dead(p, GRAD_SINK)


def block_done_hook(p):
    """The callable to run once the gradients of the transformer block that owns parameter `p` are FINAL, or None: the
    `block_done` of the gradient synchroniser (train.GradSync) attached to the FlatState that holds p's gradient sink -- instance
    state of that trainer, so several live trainers (online / probe / a second model) never see each other's blocks.  A block can be
    visited by several encoder passes of one step (mode 'mae': masked view 1 and unmasked view 2; MultiCropWrapper with two crop
    widths): every forward pass that will be differentiated arms one pending backward per block, and the hook fires when the LAST of
    them has run."""
    ent = GRAD_SINK.get(id(p))
    if ent is None or ent[0]() is not p:
        return None
    owner = ent[2]()
    sync = getattr(owner, "sync", None) if owner is not None else None
    return sync.block_done if (sync is not None and sync.active) else None


def reset_pending_backward():
    """Forget armed passes (start of a step: a forward whose backward never ran must not hold a block's all-reduce back)."""
    _PENDING_BWD.clear()


# id(q_bias) -> (weakref(q_bias), [3d] view [q_bias | 0 | v_bias]) when train.FlatState laid the two biases out around a zero
# pad (the qkv GEMM's bias vector with the k-bias fixed at zero, models/mae.py:125-128) -- otherwise it is assembled per call
QKV_BIAS = {}


def qkv_bias(qb, vb):
    ent = QKV_BIAS.get(id(qb))
    if ent is not None and ent[0]() is qb and ent[1].data_ptr() == qb.data_ptr():
        return ent[1]
    return torch.cat((qb.detach(), torch.zeros_like(vb), vb.detach()))


def grad_target(p):
    """(buffer to accumulate d/dp into, value to return to autograd)."""
    ent = GRAD_SINK.get(id(p))
    if ent is not None and ent[0]() is p:
        return ent[1], None
    t = torch.zeros_like(p)
    return t, t


WGRAD256_MIN_TILES = int(os.environ.get("SA_WGRAD256_MIN", "9"))
WGRAD_STREAM_MAX_TILES = int(os.environ.get("SA_WGRAD_STREAM", "16"))      # 0: off


def stream_wgrad(N, K, rows):
    """Does the weight gradient [N, K] over `rows` rows take the 192 x 192 streaming split-K kernel (csrc/gemm_stream.hip)?  Narrow
    outputs whose operands should stream from HBM once: measured (scripts/bench_gemm.py + rocprofv3, 127 488 rows) d = 192: qkv 50 -> 38 us,
    fc1 / fc2 62 -> 54, proj equal; d = 384 (the MAE decoder): qkv 173 -> ~135, fc1 / fc2 202 -> 175, but 384 x 384 (four tiles, each
    operand fetched twice) 52 -> 58: stays on the 128^2 kernel; d = 768 is MFMA-bound and stays on the 256^2 kernel (276 vs 288 us)."""
    t192 = ((N + 191) // 192) * ((K + 191) // 192)
    if t192 > WGRAD_STREAM_MAX_TILES or 4 * N * K < 3 * t192 * 192 * 192 or rows < 4096:
        return False
    return min(N, K) <= 192 if t192 <= 4 else min(N, K) <= 384


def wide_wgrad(N, K, rows):
    """The weight gradients that take the 256 x 256 split-K tile: measured (scripts/bench_gemm.py) with >= 9 output tiles of 256 x 256
    (d = 768: every block weight) the one-workgroup-per-CU tile wins (halved operand traffic, the long reduction hides its epilogue)."""
    return ((N + 255) // 256) * ((K + 255) // 256) >= WGRAD256_MIN_TILES and rows >= 4096


ASUM_FUSE = os.environ.get("SA_WGRAD_BIAS_FUSE", "1") != "0"


class RowSums:
    """A bias gradient that may ride along with its Linear's weight-gradient launch: `out` [>= N] fp32 gets += the column sums of dY outside
    rows [skip_lo, skip_hi) when the product runs on a streaming split-K kernel (ops.gemm(asum_out=)); `fallback()` does the same as a
    pass of its own (ops.colsum_*) when it does not."""
    __slots__ = ("out", "skip_lo", "skip_hi", "fallback")

    def __init__(self, out, skip_lo, skip_hi, fallback):
        self.out, self.skip_lo, self.skip_hi, self.fallback = out, skip_lo, skip_hi, fallback


def qv_row_sums(dqkv, d, gq, gv):
    """The q / v bias gradient of a packed qkv Linear (models/mae.py:125-128; k's bias is fixed at zero) as a RowSums: fusable when the two
    buffers sit 2 d apart in one allocation (train.FlatState's [q | 0 | v] layout), so that one [3 d] vector with the k third skipped
    covers both."""
    fb = lambda: ops.colsum_qv(dqkv, d, gq, gv)
    # (ONE allocation, not two that happen to lie 2 d apart: separately allocated bias gradients of the per-module path can come out of the
    # caching allocator exactly that far apart, and a [3 d] view of the first one's 256-byte storage does not exist)
    same = gq.untyped_storage().data_ptr() == gv.untyped_storage().data_ptr() and \
        gq.untyped_storage().nbytes() >= (gq.storage_offset() + 3 * d) * 4
    if ASUM_FUSE and same and gv.data_ptr() == gq.data_ptr() + 8 * d and gq.is_contiguous() and gv.is_contiguous() and gq.dtype == torch.float32:
        return RowSums(gq.view(-1).as_strided((3 * d,), (1,)), d, 2 * d, fb)
    return RowSums(None, 0, 0, fb)


def _wgrad(dY16, X16, out, rs=None):
    """out[N,K] += dY^T X  (TN GEMM; split-K atomics when the output has too few tiles to fill the chip).  rs: an optional RowSums (the
    Linear's bias gradient): fused into the launch on the streaming kernels, else taken by its fallback."""
    N, K = out.shape
    rows = dY16.shape[0]
    kw = {}
    if rs is not None and rs.out is not None:
        kw = dict(asum_out=rs.out, asum_skip_lo=rs.skip_lo, asum_skip_hi=rs.skip_hi)
    # measured (scripts/bench_gemm.py): with >= 9 output tiles of 256 x 256 (d = 768: every block weight) the one-workgroup-per-CU
    # 256^2 split-K tile wins (halved operand traffic, the long reduction hides its epilogue; proj at 9 tiles x 28 slices: 102 vs
    # 110 us); narrower outputs (ViT-T) stay on the 128^2 tile
    if stream_wgrad(N, K, rows):
        split = ops.pick_split_k(N, K, rows, tile=192)
        if split > 1:
            ops.gemm(dY16, X16, a_kmajor=False, b_kmajor=False, out_f32=out, split_k=split, tile256=2, **kw)
            if rs is not None and not kw:
                rs.fallback()
            return
    if wide_wgrad(N, K, rows):
        split = ops.pick_split_k(N, K, rows, tile=256)
        if split > 1:
            if not (ops.STREAM256 and K % 4 == 0):       # (the two-stage kernel has no row sums)
                kw = {}
            ops.gemm(dY16, X16, a_kmajor=False, b_kmajor=False, out_f32=out, split_k=split, tile256=True, **kw)
            if rs is not None and not kw:
                rs.fallback()
            return
    split = ops.pick_split_k(N, K, rows)
    if split > 1:
        ops.gemm(dY16, X16, a_kmajor=False, b_kmajor=False, out_f32=out, split_k=split)
    else:
        ops.gemm(dY16, X16, a_kmajor=False, b_kmajor=False, out_f32=out, accumulate=True)
    if rs is not None:
        rs.fallback()


WGRAD_GROUP = os.environ.get("SA_WGRAD_GROUP", "1") != "0"
WGRAD_GROUP256 = os.environ.get("SA_WGRAD_GROUP256", "0") == "1"


class WgradGroup:
    """The weight gradients of one transformer block's backward, launched TOGETHER at its end when they are the streaming kernel's kind
    (narrow outputs over the same rows: ViT-T's qkv / proj / fc1 / fc2, the MAE decoder's wide ones).  Split-K costs one fp32 partial
    tile per workgroup, and a launch wants a workgroup per CU: four launches write and re-read four chips' worth of partials (148 MB
    per ViT-T block against 784 MB of operands), one group launch a quarter of that (ops.gemm_wgrad_group).  Everything else goes out
    at once through `_wgrad`.  The operands stay referenced until `flush()`.
    Wide outputs (d = 768 / 1024: the 256 x 256 ring) CAN be grouped the same way (SA_WGRAD_GROUP256=1: 27 + 9 + 36 + 36 tiles split 7
    ways are 756 partial tiles instead of the 999 of four separately split launches) but measured slower -- ViT-B step 39.2 -> 40.0 ms:
    756 workgroups are three unsynchronised rounds on 256 CUs, and the tiles of a K slice no longer run side by side to share their
    operand slabs in L2 -- so they stay four launches of one round each."""
    __slots__ = ("jobs", "tile")          # jobs: (dY, X, out, RowSums or None)

    def __init__(self):
        self.jobs, self.tile = [], 0

    def add(self, dY16, X16, out, rs=None):
        N, K = out.shape
        rows = dY16.shape[0]
        tile = 0
        if WGRAD_GROUP and stream_wgrad(N, K, rows):
            tile = 192
        elif WGRAD_GROUP256 and wide_wgrad(N, K, rows) and K % 4 == 0 and ops.STREAM256:
            tile = 256
        if tile and len(self.jobs) < 8 and (not self.jobs or (self.jobs[0][0].shape[0] == rows and self.tile == tile)):
            self.jobs.append((dY16, X16, out, rs))
            self.tile = tile
        else:
            _wgrad(dY16, X16, out, rs)

    def flush(self):
        jobs, self.jobs = self.jobs, []
        if len(jobs) > 1:
            rows, t = jobs[0][0].shape[0], self.tile
            tiles = sum(((o.shape[0] + t - 1) // t) * ((o.shape[1] + t - 1) // t) for _, _, o, _ in jobs)
            split = ops.pick_split_k(0, 0, rows, tile=t, tiles=tiles)
            if split > 1:
                kw, fused = {}, None
                for i, j in enumerate(jobs):              # (one product of a group may carry its bias gradient: the qkv Linear's)
                    if j[3] is not None and j[3].out is not None and fused is None:
                        fused, kw = i, dict(asum_out=j[3].out, asum_index=i, asum_skip_lo=j[3].skip_lo, asum_skip_hi=j[3].skip_hi)
                ops.gemm_wgrad_group([j[0] for j in jobs], [j[1] for j in jobs], [j[2] for j in jobs], split, tile=t, **kw)
                for i, j in enumerate(jobs):
                    if j[3] is not None and i != fused:
                        j[3].fallback()
                return
        for j in jobs:
            _wgrad(*j)


class BlockParams:
    """Views of one transformer block's parameters (fp32 masters) -- key names follow models/mae.py."""
    __slots__ = ("n1w", "n1b", "wqkv", "qb", "vb", "wp", "bp", "n2w", "n2b", "w1", "b1", "w2", "b2")

    ORDER = __slots__

    def __init__(self, tensors):
        for n, t in zip(self.__slots__, tensors):
            setattr(self, n, t)


def block_forward(x, p, H, N, eps, save):
    """x: fp32 [M, d] residual stream -> new fp32 [M, d].  `save` (list or None) receives the backward state."""
    M, d = x.shape
    dev = x.device
    W = BF16_WEIGHTS.get
    h1 = torch.empty(M, d, dtype=BF16, device=dev)
    mean1, rstd1 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(x, p.n1w, p.n1b, eps, y_bf16=h1, mean=mean1, rstd=rstd1)
    qkv = torch.empty(M, 3 * d, dtype=BF16, device=dev)
    ops.gemm(h1, W(p.wqkv), bias=qkv_bias(p.qb, p.vb), out_bf16=qkv)        # k-bias is identically zero
    ao = torch.empty(M, d, dtype=BF16, device=dev)
    lse = torch.empty(M // N * H, N, device=dev)
    ops.attention_fwd(qkv, H, N, (d // H) ** -0.5, ao, lse)
    x2 = torch.empty(M, d, device=dev)
    ops.gemm(ao, W(p.wp), bias=p.bp.detach(), residual=x, out_f32=x2)
    h2 = torch.empty(M, d, dtype=BF16, device=dev)
    mean2, rstd2 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(x2, p.n2w, p.n2b, eps, y_bf16=h2, mean=mean2, rstd=rstd2)
    hidden = p.w1.shape[0]
    pre = torch.empty(M, hidden, dtype=BF16, device=dev) if save is not None else None
    a = torch.empty(M, hidden, dtype=BF16, device=dev)
    # with a backward to come, `pre` receives GELU'(pre-activation) (act 3): all the backward needs, and free next to GELU itself
    ops.gemm(h2, W(p.w1), bias=p.b1.detach(), act=(3 if _ACT_PAIR else 1) if pre is not None else 1, aux_out=pre, out_bf16=a)
    x3 = torch.empty(M, d, device=dev)
    ops.gemm(a, W(p.w2), bias=p.b2.detach(), residual=x2, out_f32=x3)
    if save is not None:
        save.append((x, mean1, rstd1, h1, qkv, ao, lse, x2, mean2, rstd2, h2, pre, a))
    return x3


def block_backward(dx3, dx3_16, p, g, H, N, saved, b2_done=False, prev_b2=None):
    """dx3 fp32 / dx3_16 bf16: gradient of the block output.  g: BlockParams of fp32 gradient buffers (accumulated
    into).  Returns (dx fp32, dx bf16) w.r.t. the block input.
    Bias gradients that are column sums of a LayerNorm-backward OUTPUT are accumulated inside that kernel: proj's bias
    from LN2's dx, and the PREVIOUS block's fc2 bias (`prev_b2`) from LN1's dx; `b2_done` says the later block already
    did that for this block's fc2 bias."""
    x, mean1, rstd1, h1, qkv, ao, lse, x2, mean2, rstd2, h2, pre, a = saved
    M, d = x.shape
    dev = x.device
    W = BF16_WEIGHTS.get
    wg = WgradGroup()        # the four weight gradients leave together (before LN1's backward reuses dx2_16's storage)
    # fc2
    wg.add(dx3_16, a, g.w2)
    if not b2_done:
        ops.colsum_bf16(dx3_16, g.b2, accumulate=True)
    dpre = torch.empty_like(pre)
    fuse_b1 = pre.shape[1] % 64 == 0                  # fc1's bias gradient = column sums of dpre: taken in the epilogue that produces it
    w2b, w2k = _dgrad_w(p.w2)
    ops.gemm(dx3_16, w2b, b_kmajor=w2k, act=4 if _ACT_PAIR else 2, aux_in=pre, out_bf16=dpre, colsum_out=g.b1 if fuse_b1 else None)   # (dY W2) * GELU'(pre)
    # fc1
    wg.add(dpre, h2, g.w1)
    if not fuse_b1:
        ops.colsum_bf16(dpre, g.b1, accumulate=True)
    dh2 = torch.empty(M, d, dtype=BF16, device=dev)
    w1b, w1k = _dgrad_w(p.w1)
    ops.gemm(dpre, w1b, b_kmajor=w1k, out_bf16=dh2)
    del dpre
    # LN2 + residual
    dx2 = torch.empty(M, d, device=dev)
    dx2_16 = torch.empty(M, d, dtype=BF16, device=dev)
    ops.layernorm_bwd(dh2, x2, p.n2w, mean2, rstd2, dres=dx3, dx_f32=dx2, dx_bf16=dx2_16, dgamma=g.n2w, dbeta=g.n2b, dxsum=g.bp)
    # proj
    wg.add(dx2_16, ao, g.wp)
    dao = dh2  # reuse
    wpb, wpk = _dgrad_w(p.wp)
    ops.gemm(dx2_16, wpb, b_kmajor=wpk, out_bf16=dao)
    # attention
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(qkv, H, N, (d // H) ** -0.5, ao, dao, lse, dqkv)
    # qkv
    wg.add(dqkv, h1, g.wqkv, qv_row_sums(dqkv, d, g.qb, g.vb))       # (the q / v bias gradient rides along where the kernel allows)
    wg.flush()
    dh1 = dao
    wqb, wqk = _dgrad_w(p.wqkv)
    ops.gemm(dqkv, wqb, b_kmajor=wqk, out_bf16=dh1)
    del dqkv
    # LN1 + residual
    dx = torch.empty(M, d, device=dev)
    dx_16 = dx2_16
    ops.layernorm_bwd(dh1, x, p.n1w, mean1, rstd1, dres=dx2, dx_f32=dx, dx_bf16=dx_16, dgamma=g.n1w, dbeta=g.n1b, dxsum=prev_b2)
    return dx, dx_16


def block_forward_cls(x, p, H, N, eps, save):
    """LAST block when only the CLS row of its output is consumed (MaskedAutoencoderViT.forward returns x[:, 0],
    models/mae.py:463): every token still feeds K and V, but queries, proj and the MLP are needed for row 0 of each
    sequence only -- S rows instead of S*N for 18/24 of the block's linear FLOPs.  Mathematically identical.
    Returns the compact [S, d] fp32 output rows."""
    M, d = x.shape
    S = M // N
    dev = x.device
    W = BF16_WEIGHTS.get
    h1 = torch.empty(M, d, dtype=BF16, device=dev)
    mean1, rstd1 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(x, p.n1w, p.n1b, eps, y_bf16=h1, mean=mean1, rstd=rstd1)
    qkv = torch.empty(M, 3 * d, dtype=BF16, device=dev)
    ops.gemm(h1, W(p.wqkv), bias=qkv_bias(p.qb, p.vb), out_bf16=qkv)
    ao = torch.zeros(M, d, dtype=BF16, device=dev)            # only the CLS rows are written
    lse = torch.zeros(S * H, N, device=dev)
    ops.attention_fwd(qkv, H, N, (d // H) ** -0.5, ao, lse, n_query=1)
    xc, aoc = x.view(S, N * d)[:, :d], ao.view(S, N * d)[:, :d]      # CLS rows as strided [S, d] views
    x2 = torch.empty(S, d, device=dev)
    ops.gemm(aoc, W(p.wp), bias=p.bp.detach(), residual=xc, out_f32=x2)
    h2 = torch.empty(S, d, dtype=BF16, device=dev)
    mean2, rstd2 = torch.empty(S, device=dev), torch.empty(S, device=dev)
    ops.layernorm_fwd(x2, p.n2w, p.n2b, eps, y_bf16=h2, mean=mean2, rstd=rstd2)
    hidden = p.w1.shape[0]
    pre = torch.empty(S, hidden, dtype=BF16, device=dev) if save is not None else None
    a = torch.empty(S, hidden, dtype=BF16, device=dev)
    # with a backward to come, `pre` receives GELU'(pre-activation) (act 3): all the backward needs, and free next to GELU itself
    ops.gemm(h2, W(p.w1), bias=p.b1.detach(), act=(3 if _ACT_PAIR else 1) if pre is not None else 1, aux_out=pre, out_bf16=a)
    x3 = torch.empty(S, d, device=dev)
    ops.gemm(a, W(p.w2), bias=p.b2.detach(), residual=x2, out_f32=x3)
    if save is not None:
        save.append((x, mean1, rstd1, h1, qkv, ao, lse, x2, mean2, rstd2, h2, pre, a))
    return x3


def block_backward_cls(dx3, dx3_16, p, g, H, N, saved, prev_b2=None):
    """Backward of block_forward_cls: dx3 / dx3_16 are the [S, d] gradients of the CLS output rows."""
    x, mean1, rstd1, h1, qkv, ao, lse, x2, mean2, rstd2, h2, pre, a = saved
    M, d = x.shape
    S = M // N
    dev = x.device
    W = BF16_WEIGHTS.get
    _wgrad(dx3_16, a, g.w2)
    ops.colsum_bf16(dx3_16, g.b2, accumulate=True)
    dpre = torch.empty_like(pre)
    ops.gemm(dx3_16, W(p.w2), b_kmajor=False, act=4 if _ACT_PAIR else 2, aux_in=pre, out_bf16=dpre)
    _wgrad(dpre, h2, g.w1)
    ops.colsum_bf16(dpre, g.b1, accumulate=True)
    dh2 = torch.empty(S, d, dtype=BF16, device=dev)
    ops.gemm(dpre, W(p.w1), b_kmajor=False, out_bf16=dh2)
    dx2 = torch.zeros(M, d, device=dev)                       # residual-stream gradient: non-zero on the CLS rows only
    dx2c = dx2.view(S, N * d)[:, :d]
    dx2_16 = torch.empty(S, d, dtype=BF16, device=dev)
    ops.layernorm_bwd(dh2, x2, p.n2w, mean2, rstd2, dres=dx3, dx_f32=dx2c, dgamma=g.n2w, dbeta=g.n2b)
    ops.layernorm_bwd(dh2, x2, p.n2w, mean2, rstd2, dres=dx3, dx_bf16=dx2_16)          # bf16 copy (compact rows; S x d, trivial)
    aoc = ao.view(S, N * d)[:, :d]
    _wgrad(dx2_16, aoc, g.wp)
    ops.colsum_bf16(dx2_16, g.bp, accumulate=True)
    dao = torch.zeros(M, d, dtype=BF16, device=dev)
    ops.gemm(dx2_16, W(p.wp), b_kmajor=False, out_bf16=dao.view(S, N * d)[:, :d])
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(qkv, H, N, (d // H) ** -0.5, ao, dao, lse, dqkv, n_query=1)
    _wgrad(dqkv, h1, g.wqkv, qv_row_sums(dqkv, d, g.qb, g.vb))
    dh1 = dao
    wqb, wqk = _dgrad_w(p.wqkv)
    ops.gemm(dqkv, wqb, b_kmajor=wqk, out_bf16=dh1)
    del dqkv
    dx = torch.empty(M, d, device=dev)
    dx_16 = torch.empty(M, d, dtype=BF16, device=dev)
    ops.layernorm_bwd(dh1, x, p.n1w, mean1, rstd1, dres=dx2, dx_f32=dx, dx_bf16=dx_16, dgamma=g.n1w, dbeta=g.n1b, dxsum=prev_b2)
    return dx, dx_16


class EncoderFn(torch.autograd.Function):
    """tokens [S, N, d] fp32 (CLS + patch tokens, positional terms already added) -> transformer blocks -> final
    LayerNorm.  Output: CLS latent [S, d] (pool='cls'), mean of patch tokens [S, d] (pool='mean') or the whole
    normalised sequence [S, N, d] (pool='all').  Backward returns the gradient of the tokens and of every parameter.
    """

    @staticmethod
    def forward(ctx, tokens, H, eps, pool, n_blocks, *params):
        S, N, d = tokens.shape
        blocks = [BlockParams(params[13 * i:13 * (i + 1)]) for i in range(n_blocks)]
        nw, nb = params[13 * n_blocks], params[13 * n_blocks + 1]
        need_grad = any(ctx.needs_input_grad)
        save = [] if need_grad else None
        x = tokens.detach().reshape(S * N, d)
        cls_prune = pool == "cls" and n_blocks >= 1 and N > 1
        for bp in (blocks[:-1] if cls_prune else blocks):
            x = block_forward(x, bp, H, N, eps, save)
        dev = x.device
        if pool == "cls":
            rows = block_forward_cls(x, blocks[-1], H, N, eps, save) if cls_prune else x.view(S, N * d)[:, :d]
            out = torch.empty(S, d, device=dev)
            mean, rstd = torch.empty(S, device=dev), torch.empty(S, device=dev)
            ops.layernorm_fwd(rows, nw, nb, eps, y_f32=out, mean=mean, rstd=rstd)
            x = rows
        else:
            y = torch.empty(S * N, d, device=dev)
            mean, rstd = torch.empty(S * N, device=dev), torch.empty(S * N, device=dev)
            ops.layernorm_fwd(x, nw, nb, eps, y_f32=y, mean=mean, rstd=rstd)
            if pool == "mean":
                out = torch.empty(S, d, device=dev)
                ops.mean_tokens_fwd(y.view(S, N, d), out)
            else:
                out = y.view(S, N, d)
        if need_grad and n_blocks and block_done_hook(params[2]) is not None:
            for i in range(n_blocks):
                key = id(params[13 * i + 2])
                _PENDING_BWD[key] = _PENDING_BWD.get(key, 0) + 1
        ctx.cfg = (S, N, d, H, pool, n_blocks, cls_prune)
        ctx.saved = save
        ctx.final = (x, mean, rstd)
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, dout):
        S, N, d, H, pool, n_blocks, cls_prune = ctx.cfg
        params = ctx.params
        x, mean, rstd = ctx.final
        dev = x.device
        grads, gz = [], []
        for p in params:
            if p.requires_grad:
                buf, ret = grad_target(p)
            else:
                buf, ret = torch.zeros_like(p), None      # kernels always write somewhere
            gz.append(buf)
            grads.append(ret)
        nw = params[13 * n_blocks]
        gnw, gnb = gz[13 * n_blocks], gz[13 * n_blocks + 1]
        dout = dout.contiguous().float()
        M = S * N
        if pool == "cls" and cls_prune:
            dx = torch.empty(S, d, device=dev)                 # gradient of the CLS rows only (x is the compact [S, d] output)
            dx16 = torch.empty(S, d, dtype=BF16, device=dev)
            ops.layernorm_bwd(dout, x, nw, mean, rstd, dx_f32=dx, dx_bf16=dx16, dgamma=gnw, dbeta=gnb)
        elif pool == "cls":
            dx = torch.zeros(M, d, device=dev)
            dx16 = torch.zeros(M, d, dtype=BF16, device=dev)
            ops.layernorm_bwd(dout, x, nw, mean, rstd, dx_f32=dx.view(S, N * d)[:, :d],      # x: strided CLS-row view
                              dx_bf16=dx16.view(S, N * d)[:, :d], dgamma=gnw, dbeta=gnb)
        else:
            if pool == "mean":
                dy = torch.empty(S, N, d, device=dev)
                ops.mean_tokens_bwd(dout, dy)
                dy = dy.view(M, d)
            else:
                dy = dout.reshape(M, d)
            dx = torch.empty(M, d, device=dev)
            dx16 = torch.empty(M, d, dtype=BF16, device=dev)
            ops.layernorm_bwd(dy, x, nw, mean, rstd, dx_f32=dx, dx_bf16=dx16, dgamma=gnw, dbeta=gnb)
        for i in reversed(range(n_blocks)):
            bp = BlockParams(params[13 * i:13 * (i + 1)])
            bg = BlockParams(gz[13 * i:13 * (i + 1)])
            prev_b2 = gz[13 * (i - 1) + 12] if i > 0 else None          # fc2 bias gradient buffer of block i-1
            if cls_prune and i == n_blocks - 1:
                dx, dx16 = block_backward_cls(dx, dx16, bp, bg, H, N, ctx.saved[i], prev_b2=prev_b2)
            else:
                dx, dx16 = block_backward(dx, dx16, bp, bg, H, N, ctx.saved[i], b2_done=(i < n_blocks - 1), prev_b2=prev_b2)
            ctx.saved[i] = None
            hook = block_done_hook(params[13 * i + 2])
            if hook is not None:
                key = id(params[13 * i + 2])
                left = _PENDING_BWD.get(key, 1) - 1
                _PENDING_BWD[key] = left
                if left <= 0:
                    _PENDING_BWD.pop(key, None)
                    hook(params[13 * i:13 * (i + 1)])
        dtok = dx.view(S, N, d) if ctx.needs_input_grad[0] else None
        return (dtok, None, None, None, None, *grads)


def encoder_apply(tokens, block_params, norm_w, norm_b, H, eps, pool):
    flat = []
    for bp in block_params:
        flat.extend(bp)
    return EncoderFn.apply(tokens, H, eps, pool, len(block_params), *flat, norm_w, norm_b)
