"""ConvStem patch embedding of the `vitc_*` encoders (models/mae.py:46-99) on the MI355X kernels.

`ConvStem` owns the parameters with the reference's names (`proj.0.weight`, `proj.1.{weight,bias,running_*}`, ...,
`proj.<3n>.{weight,bias}`), so checkpoints carry over; its forward is never called.  Compute is `ConvStemTokensFn`:

  x [S,1,F,T] --conv3x3 s2 (C_in = 1: direct kernel)--> BN2d(batch stats) + ReLU --> [im2col -> bf16 MFMA GEMM -> BN2d + ReLU] x (n-1)
     --> 1x1 conv GEMM (+ bias + interpolated positional rows, scattered past the CLS row) --> CLS fill [--> keep-gather]

Feature maps are channel-last: a map IS the [B*H*W, C] matrix of the GEMM and of the BatchNorm kernels, and its row order is
the token order of `x.flatten(2).transpose(1, 2)`.  Unlike PatchEmbed the stem is TRAINED (models/mae.py:186-192 freezes only
the plain patch projection), so the Function has a backward: GEMM dgrad + col2im gather, split-K wgrad, BatchNorm backward with
the statistics / sums exchanged across data-parallel ranks (SyncBN, utils/utils.py:411) exactly like the projector's.
"""
import torch
import torch.nn as nn

from . import dist as sdist
from . import ops
from .engine import BF16_WEIGHTS, _wgrad, grad_target
from .functional import pos_table_grad

BF16 = torch.bfloat16
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def to_2tuple(x):
    return tuple(x) if isinstance(x, (list, tuple)) else (x, x)


def stem_strides(patch_size):
    """models/mae.py:58-67."""
    patch_size = tuple(patch_size)
    table = {(16, 16): [(2, 2)] * 4, (16, 8): [(2, 2)] * 3 + [(2, 1)], (8, 8): [(2, 2)] * 3 + [(1, 1)], (64, 2): [(2, 2)] + [(2, 1)] * 5}
    if patch_size not in table:
        raise ValueError(f'Patch size {patch_size[0]}x{patch_size[1]} is not supported by ConvStem')
    return table[patch_size]


class ConvStem(nn.Module):
    """Parameter holder with the reference's constructor and attribute names (models/mae.py:51-87)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        if in_chans != 1:
            raise NotImplementedError("audio spectrogram input only (in_chans=1)")
        if norm_layer is not None or not flatten:
            raise NotImplementedError("ConvStem: the reference call site uses norm_layer=None, flatten=True (models/mae.py:187)")
        self.img_size, self.patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.strides = stem_strides(self.patch_size)
        assert embed_dim % 8 == 0, 'Embed dimension must be divisible by 8 for ConvStem'
        if (embed_dim // 8) % 8 != 0:
            raise NotImplementedError("ConvStem on the MI355X path needs embed_dim % 64 == 0 (channel counts are multiples of 8)")
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        stem = []
        input_dim, output_dim = 1, embed_dim // 8
        for l in range(len(self.strides)):
            stem.append(nn.Conv2d(input_dim, output_dim, kernel_size=3, stride=self.strides[l], padding=1, bias=False))
            stem.append(nn.BatchNorm2d(output_dim))
            stem.append(nn.ReLU(inplace=True))
            input_dim = output_dim
            if output_dim < embed_dim:
                output_dim *= 2
        stem.append(nn.Conv2d(input_dim, embed_dim, kernel_size=1))
        self.proj = nn.Sequential(*stem)
        self.norm = nn.Identity()

    def engine_params(self):
        """(conv weights, BN gammas, BN betas, final 1x1 weight, final bias) + the BatchNorm modules (for their buffers)."""
        n = len(self.strides)
        convs = [self.proj[3 * l] for l in range(n)]
        bns = [self.proj[3 * l + 1] for l in range(n)]
        return convs, bns, self.proj[3 * n]


def _bn_forward(h, B_rows, bn, training=True):
    """Training: batch statistics of a tall [M, C] map (SyncBN over ranks) -> mean, rstd; running buffers updated like nn.BatchNorm2d.
    Eval (`bn.eval()`, nn.BatchNorm2d's inference form): the running statistics -- no column pass, no collective, buffers untouched."""
    M, Cn = h.shape
    dev = h.device
    if not training:
        return bn.running_mean.detach(), torch.rsqrt(bn.running_var.detach() + bn.eps)
    stats = torch.empty(2, Cn, device=dev)
    ops.bn_colstats_tall(h, stats[0], stats[1])
    allst = sdist.all_gather_rows(stats)
    mean, rstd = torch.empty(Cn, device=dev), torch.empty(Cn, device=dev)
    ops.bn_finalize(allst, M, BN_EPS, BN_MOMENTUM, mean, rstd, bn.running_mean, bn.running_var)
    with torch.no_grad():
        bn.num_batches_tracked += 1
    return mean, rstd


def bn_bwd_sums(s, training):
    """The two sums the BatchNorm input gradient subtracts: the cross-rank totals in training; zeros in eval, where the layer is the
    fixed affine map of its running statistics and dx = gamma * rstd * dy (the parameter gradients are the local sums either way)."""
    if training:
        sdist.all_reduce_sum_(s)
        return s
    return torch.zeros_like(s)


def _pack_conv_weight(w, kpad, rows=None):
    """[C_out, C_in, 3, 3] fp32 master -> bf16 [rows >= C_out, kpad] in im2col order (ky, kx, c_in), zero padded (the dgrad GEMM
    reduces over C_out, which the k-major operand rule wants as a multiple of 64).  Data movement + one cast."""
    co, ci = w.shape[0], w.shape[1]
    packed = torch.zeros(rows or co, kpad, device=w.device)
    packed[:co, :9 * ci] = w.detach().permute(0, 2, 3, 1).reshape(co, 9 * ci)
    return ops.cast_bf16(packed)


def _kpad(ci):
    return (9 * ci + 63) // 64 * 64


class ConvStemTokensFn(torch.autograd.Function):
    """imgs [S,1,F,T] -> tokens [S, 1 + L (or keep), d] fp32: the conv stem, + positional rows, CLS row, optional keep-gather
    (prepare_tokens, models/mae.py:349-365 with patch_embed = ConvStem).  Gradients: every stem parameter, the CLS token and -- with
    `--use_learned_pos_embd` (:198-199; `pos_param` = the live nn.Parameter, `pos_A` = its resampling matrix or None) -- the positional table."""

    @staticmethod
    def forward(ctx, imgs, cls_token, pos, ids_keep, stem, pos_param, pos_A, *params):
        convs, bns, last = stem.engine_params()
        n = len(convs)
        S, _, F_, T_ = imgs.shape
        dev = imgs.device
        x = imgs.contiguous()
        dims = [(F_, T_, 1)]
        saved = []
        a = None
        for l in range(n):
            H, W, Ci = dims[-1]
            sh, sw = stem.strides[l]
            Ho, Wo = ops.conv_out_size(H, sh), ops.conv_out_size(W, sw)
            Co = convs[l].weight.shape[0]
            M = S * Ho * Wo
            h = torch.empty(M, Co, device=dev)
            if l == 0:
                ops.conv3x3_c1_fwd(x, convs[0].weight.detach().reshape(Co, 9).contiguous(), None, (sh, sw), h)
            else:
                kp = _kpad(Ci)
                P = torch.empty(M, kp, dtype=BF16, device=dev)
                ops.im2col3x3(a, S, H, W, Ci, (sh, sw), P)
                ops.gemm(P, _pack_conv_weight(convs[l].weight, kp), out_f32=h)
                del P
            mean, rstd = _bn_forward(h, M, bns[l], stem.training)
            a_next = torch.empty(M, Co, dtype=BF16, device=dev)
            ops.bn_apply(h, mean, rstd, bns[l].weight.detach(), bns[l].bias.detach(), True, y_bf16=a_next)
            saved.append((a, h, mean, rstd))
            a = a_next
            dims.append((Ho, Wo, Co))
        Hl, Wl, Cl = dims[-1]
        L = Hl * Wl
        d = last.weight.shape[0]
        if pos.shape[-2] != 1 + L:
            raise ValueError(f"ConvStem produced {L} patch tokens but the positional table has {pos.shape[-2] - 1}: the input length must "
                             f"give floor(T / patch) columns through every stride-2 stage (T % {stem.patch_size[1]} == 0)")
        tok = torch.empty(S, 1 + L, d, device=dev)
        pos2 = pos.detach().reshape(1 + L, d)
        ops.gemm(a, BF16_WEIGHTS.get(last.weight), bias=last.bias.detach(), residual=pos2[1:], res_mod=L, row_group=L, out_f32=tok.view(S * (1 + L), d))
        ops.fill_cls(tok, S, (1 + L) * d, d, cls_token.detach().reshape(-1), pos2[0])
        ctx.keep_rows = None
        if ids_keep is not None:
            keep = ids_keep.shape[1]
            out = torch.empty(S, 1 + keep, d, device=dev)
            rows = torch.cat([torch.zeros(S, 1, dtype=torch.int32, device=dev), ids_keep + 1], dim=1).contiguous()
            ops.gather_rows(tok, (1 + L) * d, 0, rows, out, (1 + keep) * d, 0, S, d)
            ctx.keep_rows = rows
            tok = out
        ctx.stem, ctx.saved, ctx.dims, ctx.x, ctx.a_last = stem, saved, dims, x, a
        ctx.cls_param = cls_token
        ctx.pos_param = pos_param if (pos_param is not None and pos_param.requires_grad) else None
        ctx.pos_A = pos_A
        ctx.L = L
        ctx.train = stem.training
        return tok

    @staticmethod
    def backward(ctx, dtok):
        stem, saved, dims, x, a_last = ctx.stem, ctx.saved, ctx.dims, ctx.x, ctx.a_last
        convs, bns, last = stem.engine_params()
        n = len(convs)
        S = x.shape[0]
        L = ctx.L
        dev = x.device
        dtok = dtok.contiguous()
        d = dtok.shape[-1]
        if ctx.keep_rows is not None:                       # un-gather: rows that were dropped by the masking get zero gradient
            full = torch.zeros(S, 1 + L, d, device=dev)
            keep1 = ctx.keep_rows.shape[1]
            ops.scatter_add_rows(dtok, keep1 * d, 0, ctx.keep_rows, full, (1 + L) * d, 0, S, d)
            dtok = full
        dcls = None
        if ctx.needs_input_grad[1]:
            buf, dcls = grad_target(ctx.cls_param)
            ops.cls_grad(dtok, S, (1 + L) * d, d, buf.view(-1))
        dpos = pos_table_grad(dtok, ctx.pos_param, ctx.pos_A) if ctx.pos_param is not None else None
        dy16 = ops.cast_bf16(dtok[:, 1:].contiguous().view(S * L, d))
        # 1x1 conv
        dwb, dw_last = grad_target(last.weight)
        _wgrad(dy16, a_last, dwb.view(d, -1))
        dbb, db_last = grad_target(last.bias)
        ops.colsum_bf16(dy16, dbb, accumulate=True)
        Hl, Wl, Cl = dims[-1]
        da = torch.empty(S * L, Cl, device=dev)
        ops.gemm(dy16, BF16_WEIGHTS.get(last.weight), b_kmajor=False, out_f32=da)
        W_ = sdist.get_world_size()
        grads_w, grads_g, grads_b = [None] * n, [None] * n, [None] * n
        for l in reversed(range(n)):
            a_prev, h, mean, rstd = saved[l]
            H, W, Ci = dims[l]
            Ho, Wo, Co = dims[l + 1]
            sh, sw = stem.strides[l]
            M = S * Ho * Wo
            gamma, beta = bns[l].weight, bns[l].bias
            s = torch.empty(2, Co, device=dev)
            ops.bn_bwd_stats_tall(da, h, mean, rstd, gamma, beta, True, s[0], s[1])
            dgb, grads_g[l] = grad_target(gamma)
            dbb2, grads_b[l] = grad_target(beta)
            ops.axpy(dbb2, s[0])
            ops.axpy(dgb, s[1])
            s = bn_bwd_sums(s, ctx.train)
            cpad = Co if l == 0 else (Co + 63) // 64 * 64            # zero columns up to the dgrad GEMM's K granule
            dh_full = torch.zeros(M, cpad, dtype=BF16, device=dev) if cpad != Co else torch.empty(M, Co, dtype=BF16, device=dev)
            dh = dh_full[:, :Co]
            ops.bn_bwd_apply(da, h, mean, rstd, gamma, beta, True, s[0], s[1], 1.0 / (M * W_), dx_bf16=dh)
            dwbuf, grads_w[l] = grad_target(convs[l].weight)
            if l == 0:
                ops.conv3x3_c1_wgrad(x, dh, (sh, sw), dwbuf.view(Co, 9))
                break
            kp = _kpad(Ci)
            P = torch.empty(M, kp, dtype=BF16, device=dev)
            ops.im2col3x3(a_prev, S, H, W, Ci, (sh, sw), P)          # recomputed: cheaper than keeping 9x the activations
            dwp = torch.zeros(Co, kp, device=dev)
            _wgrad(dh, P, dwp)
            del P
            # back to the parameter's [C_out, C_in, 3, 3] layout: a permuted copy (data movement), accumulated by the axpy kernel
            ops.axpy(dwbuf.view(-1), dwp[:, :9 * Ci].reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2).contiguous().view(-1))
            dP = torch.empty(M, kp, dtype=BF16, device=dev)
            ops.gemm(dh_full, _pack_conv_weight(convs[l].weight, kp, rows=cpad), b_kmajor=False, out_bf16=dP)
            da = torch.empty(S * H * W, Ci, device=dev)
            ops.col2im3x3(dP, S, H, W, Ci, (sh, sw), da)
            del dP
        out = []
        for l in range(n):
            out += [grads_w[l], grads_g[l], grads_b[l]]
        return (None, dcls, None, None, None, dpos, None, *out, dw_last, db_last)


def stem_flat_params(stem):
    convs, bns, last = stem.engine_params()
    flat = []
    for c, b in zip(convs, bns):
        flat += [c.weight, b.weight, b.bias]
    return flat + [last.weight, last.bias]
