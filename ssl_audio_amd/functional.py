"""autograd.Function shells around the HIP schedules (hand-written backward passes; no autograd tracing inside).

  TokensFn       patch-embed GEMM (+bias +interpolated pos-embed, rows scattered past CLS) + CLS fill [+ keep-gather]
  LinearFn       y = x W^T (+ b): fp32 in/out, bf16 MFMA operands               (decoder_embed / decoder_pred)
  MlpBnReluFn    Linear(no bias) -> BatchNorm1d(train) -> ReLU -> Linear(no bias) (projector / predictor chunk)
  BTLossFn       BarlowTwinsLoss.forward_loss with global-batch-exact BN / cross-correlation
"""
import torch

from . import dist as sdist
from . import ops
from .engine import BF16_WEIGHTS, _wgrad, grad_target

BF16, F32 = torch.bfloat16, torch.float32
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def pos_table_grad(full, pos_param, pos_A):
    """Gradient of the TRAINED positional table (`--use_learned_pos_embd`, models/mae.py:198-199) from the token gradient `full`
    [S, 1 + L, d] (all token positions; rows dropped by the masking hold zeros): dP = sum over the sequences of dtok at the table's own
    grid, dP[0] = sum_s dtok[s][0] and dP[1:] = A^T sum_s dtok[s][1:] when the table was resampled by the matrix A (:370-392).  Shared by
    the patch-projection and the ConvStem token assembly.  Returns what autograd should see (None when accumulated into the flat sink)."""
    S, n_tok, d = full.shape
    buf, dpos = grad_target(pos_param)
    nd = n_tok * d
    if pos_A is None:
        ops.cls_grad(full, S, nd, nd, buf.view(-1))          # column sums over the sequences, in a fixed order
    else:
        dsum = torch.zeros(n_tok, d, device=full.device)
        ops.cls_grad(full, S, nd, nd, dsum.view(-1))
        b2 = buf.view(-1, d)
        ops.axpy(b2[0], dsum[0])
        dpatch = torch.empty(pos_A.shape[1], d, device=full.device)
        ops.matmul_f32(pos_A, dsum[1:], dpatch, trans_a=True)
        ops.axpy(b2[1:].reshape(-1), dpatch.view(-1))
    return dpos


class TokensFn(torch.autograd.Function):
    """imgs [S,1,F,T] -> tokens [S, 1+L(or keep), d] fp32 (prepare_tokens, models/mae.py:349-365).
    The patch projection is frozen in the reference (models/mae.py:190-192) and so is the sin-cos positional table (:202): the
    gradients are the CLS token's and, with `--use_learned_pos_embd` (:198-199; `pos_param` = the live nn.Parameter, used at its
    own grid), the positional table's: dpos[n] = sum over the sequences of dtok[s][n] (kept rows only under random masking).
    ids_keep [S, keep] int32 (or None) applies random_masking's keep-gather."""

    @staticmethod
    def forward(ctx, imgs, cls_token, w_pe, b_pe, pos, ids_keep, pos_param=None, pos_A=None):
        S, _, F_, T_ = imgs.shape
        d = w_pe.shape[0]
        ph, pw = w_pe.shape[-2:]
        L = (F_ // ph) * (T_ // pw)
        dev = imgs.device
        patches = torch.empty(S * L, ph * pw, dtype=BF16, device=dev)
        ops.patchify_bf16(imgs.contiguous(), patches, ph, pw)
        tok = torch.empty(S, 1 + L, d, device=dev)
        pos2 = pos.detach().reshape(1 + L, d)
        ops.gemm(patches, BF16_WEIGHTS.get(w_pe), bias=b_pe.detach(), residual=pos2[1:], res_mod=L, row_group=L,
                 out_f32=tok.view(S * (1 + L), d))
        ops.fill_cls(tok, S, (1 + L) * d, d, cls_token.detach().reshape(-1), pos2[0])
        rows = None
        if ids_keep is not None:
            keep = ids_keep.shape[1]
            out = torch.empty(S, 1 + keep, d, device=dev)
            rows = torch.cat([torch.zeros(S, 1, dtype=torch.int32, device=dev), ids_keep + 1], dim=1).contiguous()   # CLS row + kept patch rows
            ops.gather_rows(tok, (1 + L) * d, 0, rows, out, (1 + keep) * d, 0, S, d)
            tok = out
        ctx.shape = tok.shape
        ctx.cls_param = cls_token
        ctx.pos_param = pos_param if (pos_param is not None and pos_param.requires_grad) else None
        ctx.rows, ctx.full, ctx.pos_A = rows, (S, 1 + L, d), pos_A
        return tok

    @staticmethod
    def backward(ctx, dtok):
        S, N, d = ctx.shape
        dcls = dpos = None
        dtok = dtok.contiguous()
        if ctx.needs_input_grad[1]:
            buf, ret = grad_target(ctx.cls_param)
            ops.cls_grad(dtok, S, N * d, d, buf.view(-1))
            dcls = ret
        if ctx.pos_param is not None:
            full = dtok
            if ctx.rows is not None:                 # masked: put the kept rows back at their token positions (zeros elsewhere) first
                full = torch.zeros(ctx.full, device=dtok.device)
                ops.scatter_add_rows(dtok, N * d, 0, ctx.rows, full, ctx.full[1] * d, 0, S, d)
            dpos = pos_table_grad(full, ctx.pos_param, ctx.pos_A)
        return None, dcls, None, None, None, None, dpos, None


class MaeUnshuffleFn(torch.autograd.Function):
    """forward_decoder's token assembly (models/mae.py:413-420): append mask tokens, un-shuffle with ids_restore, add the
    decoder positional table.  x [B, 1+keep, d] fp32 -> [B, 1+L, d] fp32; gradients: x and mask_token."""

    @staticmethod
    def forward(ctx, x, mask_token, pos, ids_restore):
        x = x.contiguous()
        B, kp1, d = x.shape
        ids = ids_restore.to(torch.int32).contiguous()
        out = torch.empty(B, ids.shape[1] + 1, d, device=x.device)
        ops.mae_unshuffle_fwd(x, mask_token.detach().reshape(-1).contiguous(), pos.detach().reshape(-1, d).contiguous(), ids, out)
        ctx.save_for_backward(ids)
        ctx.keep = kp1 - 1
        ctx.mask_param = mask_token
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        dout = dout.contiguous()
        B, _, d = dout.shape
        dx = torch.empty(B, ctx.keep + 1, d, device=dout.device)
        dmask = None
        buf = None
        if ctx.needs_input_grad[1]:
            buf, dmask = grad_target(ctx.mask_param)
        ops.mae_unshuffle_bwd(dout, ctx.keep, ids, dx, buf.view(-1) if buf is not None else None)
        return dx, dmask, None, None


class MaeReconLossFn(torch.autograd.Function):
    """forward_loss (models/mae.py:437-453) with patchify folded in: masked mean of the per-patch MSE.  One input channel.
    Data parallel: the mean is over the GLOBAL batch's masked patches (acc2 is all-reduced), like every other term of the loss.
    `pred` is decoder_pred's full output [B, row0 + L, P]; row0 = 1 skips its CLS row in place (models/mae.py:433)."""

    @staticmethod
    def forward(ctx, pred, imgs, mask, ph, pw, row0, norm_pix=False):
        pred = pred.contiguous(); imgs = imgs.contiguous(); mask = mask.contiguous().float()
        if imgs.shape[1] != 1:
            raise ValueError("MaeReconLossFn: the audio path has one input channel")
        acc2 = torch.empty(2, device=pred.device)
        loss = torch.empty((), device=pred.device)
        ops.mae_recon_loss_fwd(pred, row0, imgs, mask, ph, pw, acc2, loss.view(1), norm_pix)
        if sdist.collectives_active():             # global masked mean: (sum of masked errors, mask count) summed over ranks
            sdist.all_reduce_sum_(acc2)
            ops.mae_recon_loss_finalize(acc2, loss.view(1))
        ctx.save_for_backward(pred, imgs, mask, acc2)
        ctx.cfg = (ph, pw, row0, norm_pix)
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, imgs, mask, acc2 = ctx.saved_tensors
        ph, pw, row0, norm_pix = ctx.cfg
        dpred = torch.empty_like(pred)
        ops.mae_recon_loss_bwd(pred, row0, imgs, mask, ph, pw, acc2, g.reshape(1).float().contiguous(), dpred, norm_pix)
        return dpred, None, None, None, None, None, None


class MeanTokensFn(torch.autograd.Function):
    """`x[:, 1:].mean(dim=1)` (models/mae.py:461-462) on a [S, N, d] fp32 sequence."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        out = torch.empty(x.shape[0], x.shape[2], device=x.device)
        ops.mean_tokens_fwd(x, out)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        dy = torch.empty(ctx.shape, device=dout.device)
        ops.mean_tokens_bwd(dout.contiguous().float(), dy)
        return dy


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        x16 = ops.cast_bf16(x2)
        out = torch.empty(x2.shape[0], weight.shape[0], device=x.device)
        ops.gemm(x16, BF16_WEIGHTS.get(weight), bias=bias.detach() if bias is not None else None, out_f32=out)
        ctx.save_for_backward(x16, weight, bias)
        ctx.xshape = x.shape
        return out.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x16, weight, bias = ctx.saved_tensors
        dy16 = ops.cast_bf16(dy.reshape(-1, dy.shape[-1]).contiguous())
        dwb, dw = grad_target(weight)
        _wgrad(dy16, x16, dwb)
        db = None
        if bias is not None:
            dbb, db = grad_target(bias)
            ops.colsum_bf16(dy16, dbb, accumulate=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x16.shape[0], weight.shape[1], device=dy.device)
            ops.gemm(dy16, BF16_WEIGHTS.get(weight), b_kmajor=False, out_f32=dx)
            dx = dx.view(ctx.xshape)
        return dx, dw, db


def _bn_forward_stats(h, eps, running_mean, running_var):
    """Local (mean, M2) -> all-gather -> combined mean / rstd (+ running buffers)."""
    B, Cn = h.shape
    dev = h.device
    stats = torch.empty(2, Cn, device=dev)
    ops.bn_colstats(h, stats[0], stats[1])
    allst = sdist.all_gather_rows(stats)
    mean, rstd = torch.empty(Cn, device=dev), torch.empty(Cn, device=dev)
    ops.bn_finalize(allst, B, eps, BN_MOMENTUM, mean, rstd, running_mean, running_var)
    return mean, rstd


def _gemm_f32_long_k(A16, W16, out, b_kmajor=True):
    """out = A W^T (or A W) in fp32 for the projector's small-batch GEMMs: a few hundred rows against an 8192-long reduction is one or
    two output tiles walking 128 K-steps on six CUs (130 us); splitting the reduction over the chip brings it to the launch floor."""
    M, K = A16.shape
    N = W16.shape[0] if b_kmajor else W16.shape[1]
    split = ops.pick_split_k(M, N, K)
    if split > 1:
        out.zero_()
        ops.gemm(A16, W16, b_kmajor=b_kmajor, out_f32=out, split_k=split)
    else:
        ops.gemm(A16, W16, b_kmajor=b_kmajor, out_f32=out)


class MlpBnReluFn(torch.autograd.Function):
    """One crop chunk through Linear(nb) -> BN1d(train, affine) -> ReLU -> Linear(nb)  (model.py:16-23,39-45).
    With a process group the BN is a SyncBN over the global chunk (utils/utils.py:411)."""

    @staticmethod
    def forward(ctx, x, w0, gamma, beta, w1, running_mean, running_var):
        B = x.shape[0]
        dev = x.device
        x16 = ops.cast_bf16(x.contiguous())
        h = torch.empty(B, w0.shape[0], device=dev)
        ops.gemm(x16, BF16_WEIGHTS.get(w0), out_f32=h)
        mean, rstd = _bn_forward_stats(h, BN_EPS, running_mean, running_var)
        a = torch.empty(B, w0.shape[0], dtype=BF16, device=dev)
        ops.bn_apply(h, mean, rstd, gamma.detach(), beta.detach(), True, y_bf16=a)
        z = torch.empty(B, w1.shape[0], device=dev)
        _gemm_f32_long_k(a, BF16_WEIGHTS.get(w1), z)
        ctx.save_for_backward(x16, h, mean, rstd, a, w0, gamma, beta, w1)
        return z

    @staticmethod
    def backward(ctx, dz):
        x16, h, mean, rstd, a, w0, gamma, beta, w1 = ctx.saved_tensors
        B, Hd = h.shape
        dev = h.device
        W = sdist.get_world_size()
        dz16 = ops.cast_bf16(dz.contiguous())
        dw1b, dw1 = grad_target(w1)
        _wgrad(dz16, a, dw1b)
        da = torch.empty(B, Hd, device=dev)      # fp32: BN backward subtracts batch means, keep its input unrounded
        ops.gemm(dz16, BF16_WEIGHTS.get(w1), b_kmajor=False, out_f32=da)
        s = torch.empty(2, Hd, device=dev)
        ops.bn_bwd_stats(da, h, mean, rstd, gamma, beta, True, s[0], s[1])
        dgb, dgamma = grad_target(gamma)                    # local contributions (summed over ranks with the other grads)
        dbb, dbeta = grad_target(beta)
        ops.axpy(dbb, s[0])
        ops.axpy(dgb, s[1])
        sdist.all_reduce_sum_(s)
        dh = torch.empty(B, Hd, dtype=BF16, device=dev)
        ops.bn_bwd_apply(da, h, mean, rstd, gamma, beta, True, s[0], s[1], 1.0 / (B * W), dx_bf16=dh)
        dw0b, dw0 = grad_target(w0)
        _wgrad(dh, x16, dw0b)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, w0.shape[1], device=dev)
            _gemm_f32_long_k(dh, BF16_WEIGHTS.get(w0), dx, b_kmajor=False)
        return dx, dw0, dgamma, dbeta, dw1, None, None


class LinearBnReluFn(torch.autograd.Function):
    """One hidden block Linear(nb) -> BN1d(train, affine) -> ReLU of a projector with `--projector_n_hidden_layers` > 1 (model.py:16-22):
    the front part of MlpBnReluFn on its own, fp32 out (the next block casts its own GEMM operand)."""

    @staticmethod
    def forward(ctx, x, w0, gamma, beta, running_mean, running_var):
        B = x.shape[0]
        dev = x.device
        x16 = ops.cast_bf16(x.contiguous())
        h = torch.empty(B, w0.shape[0], device=dev)
        ops.gemm(x16, BF16_WEIGHTS.get(w0), out_f32=h)
        mean, rstd = _bn_forward_stats(h, BN_EPS, running_mean, running_var)
        a = torch.empty(B, w0.shape[0], device=dev)
        ops.bn_apply(h, mean, rstd, gamma.detach(), beta.detach(), True, y_f32=a)
        ctx.save_for_backward(x16, h, mean, rstd, w0, gamma, beta)
        return a

    @staticmethod
    def backward(ctx, da):
        x16, h, mean, rstd, w0, gamma, beta = ctx.saved_tensors
        B, Hd = h.shape
        dev = h.device
        W = sdist.get_world_size()
        da = da.contiguous().float()
        s = torch.empty(2, Hd, device=dev)
        ops.bn_bwd_stats(da, h, mean, rstd, gamma, beta, True, s[0], s[1])
        dgb, dgamma = grad_target(gamma)                    # local contributions (summed over ranks with the other grads)
        dbb, dbeta = grad_target(beta)
        ops.axpy(dbb, s[0])
        ops.axpy(dgb, s[1])
        sdist.all_reduce_sum_(s)
        dh = torch.empty(B, Hd, dtype=BF16, device=dev)
        ops.bn_bwd_apply(da, h, mean, rstd, gamma, beta, True, s[0], s[1], 1.0 / (B * W), dx_bf16=dh)
        dw0b, dw0 = grad_target(w0)
        _wgrad(dh, x16, dw0b)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, w0.shape[1], device=dev)
            _gemm_f32_long_k(dh, BF16_WEIGHTS.get(w0), dx, b_kmajor=False)
        return dx, dw0, dgamma, dbeta, None, None


class BTLossFn(torch.autograd.Function):
    """forward_loss(z1, z2) (utils/loss.py:15-30), fp32.  Data-parallel semantics: BN statistics and the
    cross-correlation are those of the GLOBAL batch; `literal_ddp=True` reproduces the reference's
    per-rank BN + division by the local batch + SUM all-reduce instead.

    Exchanges per loss term (global-exact mode), each ONE collective:
      forward   all-gather of the packed per-rank statistics [z1: mean, M2 | z2: mean, M2]  (4*D floats per rank)
                all-reduce SUM of the D x D cross-correlation partials                     (utils/loss.py:20-21)
      backward  all-reduce SUM of the packed BatchNorm backward sums [z1: s1, s2 | z2: s1, s2] (4*D floats)
    """

    @staticmethod
    def forward(ctx, z1, z2, alpha, lmbda, hsic, running_mean, running_var, literal_ddp):
        B, D = z1.shape
        dev = z1.device
        W = sdist.get_world_size()
        local_bn = bool(literal_ddp and W > 1)
        z1, z2 = z1.contiguous().float(), z2.contiguous().float()
        if z1.stride(0) != z2.stride(0):
            z2 = z2.clone(memory_format=torch.contiguous_format)
        # five launches around the three collectives (csrc/bt_fused.hip; the piecewise kernels stay for the projector's BatchNorm)
        stats = torch.empty(2, 2, D, device=dev)                 # [view][mean | M2][D]
        ops.bt_stats2(z1, z2, stats)
        allst = stats.unsqueeze(0) if local_bn else sdist.all_gather_rows(stats)        # [W, 2, 2, D]: one all-gather for both views
        n_eff = B if (literal_ddp or W == 1) else B * W
        mean, rstd = torch.empty(2, D, device=dev), torch.empty(2, D, device=dev)
        z1n, z2n = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
        c = torch.empty(D, D, device=dev)
        # bn(z1) then bn(z2): the running buffers see both, in this order (utils/loss.py:17); c = z1n^T z2n / n (:19)
        ops.bt_corr(z1, z2, allst.contiguous(), BN_EPS, BN_MOMENTUM, 1.0 / n_eff, mean, rstd, running_mean, running_var, z1n, z2n, c)
        sdist.all_reduce_sum_(c)                                 # utils/loss.py:20-21
        loss = torch.empty(1, device=dev)
        G = torch.empty(D, D, device=dev)
        ops.bt_loss_grad(c, alpha, lmbda, hsic, loss, G)
        ctx.save_for_backward(z1n, z2n, rstd, G)
        ctx.cfg = (n_eff, local_bn)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        z1n, z2n, rstd, G = ctx.saved_tensors
        n_eff, local_bn = ctx.cfg
        B, D = z1n.shape
        dev = z1n.device
        dzn = torch.empty(2, B, D, device=dev)
        s = torch.empty(2, 2, D, device=dev)                     # [view][s1 | s2][D]: one all-reduce for both views
        ops.bt_bwd_products(z1n, z2n, G, 1.0 / n_eff, dzn, s)    # dz1n = z2n G^T / n, dz2n = z1n G / n and their column sums
        if not local_bn:
            sdist.all_reduce_sum_(s)
        scale = dloss.reshape(1).float().contiguous()
        dz1, dz2 = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
        ops.bt_bwd_apply(z1n, z2n, rstd, dzn, s, 1.0 / (B if local_bn else n_eff), scale, dz1, dz2)
        return dz1, dz2, None, None, None, None, None, None
