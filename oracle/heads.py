"""Projector / predictor / Barlow Twins loss oracle (TEST INFRASTRUCTURE).  Plain PyTorch CPU ops.

Follows model.py:11-53 (BarlowTwinsHead, BarlowTwinsPredictor), utils/loss.py:8-48 (BarlowTwinsLoss),
utils/utils.py:23-27 (off_diagonal).  Batch-norm running statistics are returned explicitly.
"""
import torch
import torch.nn.functional as F

from . import rounding as R   # identity hooks unless rounding.mirror_hip_bf16() is active

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def off_diagonal(x):
    """utils/utils.py:23-27."""
    n, m = x.shape
    assert n == m
    return x.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten()


def batchnorm_train(x, weight=None, bias=None, eps=BN_EPS):
    """nn.BatchNorm1d in train mode: biased variance for normalisation; returns (y, mean, biased var)."""
    mu = x.mean(0)
    var = x.var(0, unbiased=False)
    y = (x - mu) * torch.rsqrt(var + eps)
    if weight is not None:
        y = y * weight + bias
    return y, mu, var


def running_update(rm, rv, mu, var, n, momentum=BN_MOMENTUM):
    """running stats use the UNBIASED variance (PyTorch BatchNorm semantics)."""
    return (1 - momentum) * rm + momentum * mu, (1 - momentum) * rv + momentum * var * n / max(n - 1, 1)


def mlp_bn_relu(x, w0, g, b, w1, ncrops):
    """Linear(no bias) -> BN1d(train) -> ReLU -> Linear(no bias), applied per crop chunk
    (BarlowTwinsHead.forward model.py:25-31 / BarlowTwinsPredictor.forward model.py:47-53)."""
    outs, stats = [], []
    for xc in x.chunk(ncrops):
        h = R.qb(F.linear(R.qf(xc), R.qw(w0)))
        hn, mu, var = batchnorm_train(h, g, b)
        stats.append((mu, var, h.shape[0]))
        outs.append(R.qb(F.linear(R.qf(F.relu(hn)), R.qw(w1))))
    return torch.cat(outs), stats


def head_forward(x, sd, ncrops=2, prefix="projector."):
    """BarlowTwinsHead.forward (model.py:25-31) for any `projector_n_hidden_layers` (model.py:16-22: n x [Linear, BatchNorm1d, ReLU] +
    [Linear], keys projector.{3i}.weight / projector.{3i+1}.{weight,bias}); returns (z, BatchNorm statistics in call order)."""
    n_hidden = 0
    while f"{prefix}{3 * n_hidden + 1}.weight" in sd:
        n_hidden += 1
    if n_hidden == 1:
        return mlp_bn_relu(x, sd[prefix + "0.weight"], sd[prefix + "1.weight"], sd[prefix + "1.bias"], sd[prefix + "3.weight"], ncrops)
    outs, stats = [], []
    for xc in x.chunk(ncrops):
        h = xc
        for i in range(n_hidden):
            h = R.qb(F.linear(R.qf(h), R.qw(sd[f"{prefix}{3 * i}.weight"])))
            hn, mu, var = batchnorm_train(h, sd[f"{prefix}{3 * i + 1}.weight"], sd[f"{prefix}{3 * i + 1}.bias"])
            stats.append((mu, var, h.shape[0]))
            h = F.relu(hn)
        outs.append(R.qb(F.linear(R.qf(h), R.qw(sd[f"{prefix}{3 * n_hidden}.weight"]))))
    return torch.cat(outs), stats


def predictor_forward(x, sd, ncrops=2):
    return mlp_bn_relu(x, sd["predictor.0.weight"], sd["predictor.1.weight"], sd["predictor.1.bias"], sd["predictor.3.weight"], ncrops)


def bt_cross_correlation(z1, z2, eps=BN_EPS):
    """c = BN(z1)^T BN(z2) / B  (utils/loss.py:17-19; affine-less BN, biased variance)."""
    n1, mu1, v1 = batchnorm_train(z1, eps=eps)
    n2, mu2, v2 = batchnorm_train(z2, eps=eps)
    return n1.T @ n2 / z1.shape[0], (mu1, v1, mu2, v2)


def bt_loss_from_c(c, alpha=1.0, lmbda=0.005, hsic=False):
    """utils/loss.py:23-30."""
    on = (torch.diagonal(c) - 1).pow(2).sum()
    off = (off_diagonal(c) + 1).pow(2).sum() if hsic else off_diagonal(c).pow(2).sum()
    return alpha * on + lmbda * off


def bt_forward_loss(z1, z2, alpha=1.0, lmbda=0.005, hsic=False):
    """BarlowTwinsLoss.forward_loss, single process (= global-batch-exact semantics)."""
    c, stats = bt_cross_correlation(z1, z2)
    return bt_loss_from_c(c, alpha, lmbda, hsic), stats


def bt_forward_loss_backward(z1, z2, alpha=1.0, lmbda=0.005, hsic=False, eps=BN_EPS):
    """Analytic forward + backward, no autograd: returns loss, dz1, dz2."""
    B = z1.shape[0]

    def norm(z):
        mu = z.mean(0)
        r = torch.rsqrt(z.var(0, unbiased=False) + eps)
        return (z - mu) * r, r

    n1, r1 = norm(z1)
    n2, r2 = norm(z2)
    c = n1.T @ n2 / B
    loss = bt_loss_from_c(c, alpha, lmbda, hsic)
    G = 2 * lmbda * (c + 1 if hsic else c)
    G = G - torch.diag(torch.diagonal(G)) + torch.diag(2 * alpha * (torch.diagonal(c) - 1))
    dn1, dn2 = n2 @ G.T / B, n1 @ G / B

    def bn_bwd(dn, n, r):
        return r * (dn - dn.mean(0) - n * (dn * n).mean(0))

    return loss, bn_bwd(dn1, n1, r1), bn_bwd(dn2, n2, r2)


def bt_forward(student_output, teacher_output, ncrops, ngcrops_each=1, alpha=1.0, lmbda=0.005, hsic=False):
    """BarlowTwinsLoss.forward (utils/loss.py:32-48): returns (loss, [stats per term in call order])."""
    student_out = student_output.chunk(ncrops - (2 - ngcrops_each))
    teacher_out = teacher_output.chunk(ngcrops_each)
    total, n_terms, all_stats = 0, 0, []
    for q in range(len(teacher_out)):
        for v in range(len(student_out)):
            if len(teacher_out) > 1 and q == v:
                continue
            l, st = bt_forward_loss(teacher_out[q], student_out[v], alpha, lmbda, hsic)
            total = total + l
            n_terms += 1
            all_stats.append((st, teacher_out[q].shape[0]))
    return total / n_terms, all_stats


def bt_running_stats(all_stats, D, dtype=torch.float32):
    """Replays the loss module's BatchNorm1d buffer updates: z1 then z2 per term."""
    rm, rv, nbt = torch.zeros(D, dtype=dtype), torch.ones(D, dtype=dtype), 0
    for (mu1, v1, mu2, v2), n in all_stats:
        rm, rv = running_update(rm, rv, mu1, v1, n)
        rm, rv = running_update(rm, rv, mu2, v2, n)
        nbt += 2
    return rm, rv, nbt
