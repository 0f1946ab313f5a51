"""TEST INFRASTRUCTURE: a torch-CPU stand-in for the subset of `ssl_audio_amd.ops` that the projector / predictor / loss
schedules call (functional.MlpBnReluFn, functional.BTLossFn, engine._wgrad, engine.BF16_WEIGHTS).

The 2-rank gloo tests (no GPU in the build container) patch `functional.ops` / `engine.ops` with this module so that the
REAL exchange code of the product -- which statistics travel, how they are packed into collectives, how the results are
combined, in which order running statistics are updated -- runs under a real process group.  Each function restates the
contract of the `sa_*` entry point of the same name (include/ssl_audio_hip.h); the HIP kernels themselves are pinned by the
-m gpu tests.  Never imported by the package.
"""
import torch

BF16, F32 = torch.bfloat16, torch.float32


def cast_bf16(src, dst=None):
    out = src.to(BF16)
    if dst is not None:
        dst.copy_(out)
        return dst
    return out


def cast_f32_from_bf16(src, dst):
    dst.copy_(src.to(F32))
    return dst


def pick_split_k(M, N, K, cu_count=None, tile=128, tiles=None):
    return 1


def gemm_wgrad_group(dY, X, out, split_k, tile=192, asum_out=None, asum_index=-1, asum_skip_lo=0, asum_skip_hi=0):
    for dy, x, o in zip(dY, X, out):
        o += dy.float().t() @ x.float()
    if asum_out is not None:
        s = dY[asum_index].float().sum(0)
        s[asum_skip_lo:asum_skip_hi] = 0
        asum_out[:s.numel()] += s


def gemm(A, B, *, a_kmajor=True, b_kmajor=True, alpha=1.0, bias=None, act=0, aux_in=None, aux_out=None, residual=None, res_mod=0,
         out_f32=None, out_bf16=None, row_group=0, split_k=1, accumulate=False, tile256=False, colsum_out=None):
    assert act == 0 and aux_in is None and aux_out is None and residual is None and row_group == 0 and colsum_out is None
    a = A.float() if a_kmajor else A.float().t()
    b = B.float() if b_kmajor else B.float().t()
    d = alpha * (a @ b.t())
    if bias is not None:
        d = d + bias
    if out_f32 is not None:
        if accumulate or split_k > 1:
            out_f32.add_(d)
        else:
            out_f32.copy_(d)
    if out_bf16 is not None:
        out_bf16.copy_(d.to(BF16))


def colsum_bf16(x, out, accumulate=False, n_ranges=1, range_stride=0):
    assert n_ranges == 1
    s = x.float().sum(0)
    out.add_(s) if accumulate else out.copy_(s)
    return out


def colsum_qv(dqkv, d, gq, gv):
    colsum_bf16(dqkv[:, :d], gq, accumulate=True)
    colsum_bf16(dqkv[:, 2 * d:], gv, accumulate=True)


def axpy(y, x, a=1.0):
    y.add_(x, alpha=a)


def bn_colstats(x, mean, m2):
    mu = x.mean(0)
    mean.copy_(mu)
    m2.copy_(((x - mu) ** 2).sum(0))


def bn_finalize(stats, rows_per_rank, eps, momentum, mean, rstd, running_mean=None, running_var=None):
    W = stats.shape[0]
    mu = stats[:, 0].mean(0)
    m2 = (stats[:, 1] + rows_per_rank * (stats[:, 0] - mu) ** 2).sum(0)
    n = W * rows_per_rank
    mean.copy_(mu)
    rstd.copy_(torch.rsqrt(m2 / n + eps))
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(mu, alpha=momentum)
    if running_var is not None:
        running_var.mul_(1 - momentum).add_(m2 / max(n - 1, 1), alpha=momentum)


def _xhat(x, mean, rstd):
    return (x - mean) * rstd


def bn_apply(x, mean, rstd, gamma=None, beta=None, relu=False, *, y_f32=None, y_bf16=None):
    v = _xhat(x, mean, rstd)
    if gamma is not None:
        v = v * gamma + beta
    if relu:
        v = torch.relu(v)
    if y_f32 is not None:
        y_f32.copy_(v)
    if y_bf16 is not None:
        y_bf16.copy_(v.to(BF16))


def _g(dy, x, mean, rstd, gamma, beta, relu):
    g = dy.float()
    if relu:
        pre = _xhat(x, mean, rstd)
        if gamma is not None:
            pre = pre * gamma + beta
        g = g * (pre > 0)
    return g


def bn_bwd_stats(dy, x, mean, rstd, gamma, beta, relu, s1, s2):
    g = _g(dy, x, mean, rstd, gamma, beta, relu)
    s1.copy_(g.sum(0))
    s2.copy_((g * _xhat(x, mean, rstd)).sum(0))


def bn_bwd_apply(dy, x, mean, rstd, gamma, beta, relu, s1, s2, inv_n, *, out_scale=None, dx_f32=None, dx_bf16=None):
    g = _g(dy, x, mean, rstd, gamma, beta, relu)
    dx = rstd * (g - s1 * inv_n - _xhat(x, mean, rstd) * s2 * inv_n)
    if gamma is not None:
        dx = dx * gamma
    if out_scale is not None:
        dx = dx * out_scale
    if dx_f32 is not None:
        dx_f32.copy_(dx)
    if dx_bf16 is not None:
        dx_bf16.copy_(dx.to(BF16))


def matmul_f32(A, B, out, *, trans_a=False, trans_b=False, alpha=1.0):
    a = A.t() if trans_a else A
    b = B.t() if trans_b else B
    out.copy_(alpha * (a @ b))
    return out


def bt_loss_grad(c, alpha, lmbda, hsic, loss, G=None):
    D = c.shape[0]
    eye = torch.eye(D, dtype=torch.bool)
    off = (c + 1.0) if hsic else c
    loss.copy_((alpha * (torch.diagonal(c) - 1).pow(2).sum() + lmbda * off[~eye].pow(2).sum()).reshape(1))
    if G is not None:
        g = 2 * lmbda * off
        g[eye] = 2 * alpha * (torch.diagonal(c) - 1)
        G.copy_(g)


# ---- the fused loss-term entry points (csrc/bt_fused.hip), restated on the pieces above
def bt_stats2(z1, z2, stats):
    for v, z in enumerate((z1, z2)):
        bn_colstats(z, stats[v, 0], stats[v, 1])


def bt_corr(z1, z2, all_stats, eps, momentum, inv_n, mean, rstd, running_mean, running_var, z1n, z2n, c):
    B = z1.shape[0]
    for v, (z, zn) in enumerate(((z1, z1n), (z2, z2n))):       # bn(z1) then bn(z2): the running buffers see both, in this order
        bn_finalize(all_stats[:, v], B, eps, momentum, mean[v], rstd[v], running_mean, running_var)
        zn.copy_(_xhat(z, mean[v], rstd[v]))
    c.copy_(inv_n * (z1n.t() @ z2n))


def bt_bwd_products(z1n, z2n, G, inv_n, dzn, sums):
    dzn[0].copy_(inv_n * (z2n @ G.t()))
    dzn[1].copy_(inv_n * (z1n @ G))
    for v, zn in enumerate((z1n, z2n)):
        sums[v, 0].copy_(dzn[v].sum(0))
        sums[v, 1].copy_((dzn[v] * zn).sum(0))


def bt_bwd_apply(z1n, z2n, rstd, dzn, sums, inv_n, out_scale, dz1, dz2):
    for v, (zn, dz) in enumerate(((z1n, dz1), (z2n, dz2))):
        d = rstd[v] * (dzn[v] - sums[v, 0] * inv_n - zn * sums[v, 1] * inv_n)
        dz.copy_(d * out_scale if out_scale is not None else d)
STREAM256 = False
