"""Data-parallel protocol on CPU with 2 gloo ranks (no GPU): the exchanges ssl_audio_amd.dist provides, combined as
functional.BTLossFn / MlpBnReluFn / train.GradSync combine them, reproduce the SINGLE-process result on the global
batch (global-batch-exact semantics) -- and `literal_ddp` reproduces the reference's DDP quirk.

The per-rank arithmetic is written with plain torch CPU ops here (the HIP kernels need a GPU); what is under test is
the exchange pattern: which quantities travel, how they are combined, and that gradients are SUMMED over ranks.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import heads as oh
from ssl_audio_amd import dist as sdist

EPS = 1e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def chan_combine(allst, rows_per_rank):
    """What sa_bn_finalize does on device: combine per-rank (mean, M2) with equal row counts."""
    W = allst.shape[0]
    mean = allst[:, 0].mean(0)
    m2 = (allst[:, 1] + rows_per_rank * (allst[:, 0] - mean) ** 2).sum(0)
    return mean, m2 / (W * rows_per_rank)


def local_stats(z):
    mu = z.mean(0)
    return torch.stack([mu, ((z - mu) ** 2).sum(0)])


def dp_bt_loss(z1, z2, alpha, lmbda, literal):
    """The exchange sequence of functional.BTLossFn on one rank; returns loss, dz1, dz2 (local rows)."""
    B, D = z1.shape
    W = sdist.get_world_size()
    norm, rs = [], []
    for z in (z1, z2):
        if literal:
            mean, var = z.mean(0), z.var(0, unbiased=False)
        else:
            mean, var = chan_combine(sdist.all_gather_rows(local_stats(z)), B)
        r = torch.rsqrt(var + EPS)
        norm.append((z - mean) * r)
        rs.append(r)
    n_eff = B if literal else B * W
    c = norm[0].T @ norm[1] / n_eff
    sdist.all_reduce_sum_(c)
    loss = oh.bt_loss_from_c(c, alpha, lmbda)
    G = 2 * lmbda * c
    G = G - torch.diag(torch.diagonal(G)) + torch.diag(2 * alpha * (torch.diagonal(c) - 1))
    dn = [norm[1] @ G.T / n_eff, norm[0] @ G / n_eff]
    out = []
    for zn, d, r in zip(norm, dn, rs):
        s = torch.stack([d.sum(0), (d * zn).sum(0)])
        if not literal:
            sdist.all_reduce_sum_(s)
        n = B if literal else n_eff
        out.append(r * (d - s[0] / n - zn * s[1] / n))
    return loss, out[0], out[1]


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:  # surface the failure instead of leaving the parent waiting on the queue
        import traceback
        q.put((rank, False, False, False, False, repr(e) + traceback.format_exc(), 0.0))


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, l, w = sdist.init_from_env("gloo")
    assert (r, w) == (rank, world) and sdist.get_world_size() == world and sdist.is_main_process() == (rank == 0)
    torch.manual_seed(0)
    Bg, D = 16, 24
    z1g = torch.randn(Bg, D, dtype=torch.float64)
    z2g = z1g + 0.3 * torch.randn(Bg, D, dtype=torch.float64)
    sl = slice(rank * Bg // world, (rank + 1) * Bg // world)            # rank r owns a contiguous shard
    # ---- global-exact mode == single process on the global batch
    loss, dz1, dz2 = dp_bt_loss(z1g[sl], z2g[sl], 1.0, 0.005, literal=False)
    ref_l, ref_d1, ref_d2 = oh.bt_forward_loss_backward(z1g, z2g, 1.0, 0.005)
    ok = abs(float(loss) - float(ref_l)) < 1e-9 and torch.allclose(dz1, ref_d1[sl], atol=1e-12) and torch.allclose(dz2, ref_d2[sl], atol=1e-12)
    # ---- literal mode == the reference under DDP: per-rank BN, c / B_local, SUM all-reduce -> diagonal ~ world
    lit_loss, _, _ = dp_bt_loss(z1g[sl], z2g[sl], 1.0, 0.005, literal=True)
    parts = []
    for rr in range(world):
        s2 = slice(rr * Bg // world, (rr + 1) * Bg // world)
        cpart, _ = oh.bt_cross_correlation(z1g[s2], z2g[s2])
        parts.append(cpart)
    lit_ref = oh.bt_loss_from_c(sum(parts), 1.0, 0.005)
    ok_lit = abs(float(lit_loss) - float(lit_ref)) < 1e-9 and float(lit_loss) > 5 * float(loss)
    # ---- SyncBN statistics of the projector: Chan combination == global batch statistics
    h = torch.randn(Bg, 40, dtype=torch.float64) * 2 + 1
    mean, var = chan_combine(sdist.all_gather_rows(local_stats(h[sl])), Bg // world)
    ok_bn = torch.allclose(mean, h.mean(0), atol=1e-12) and torch.allclose(var, h.var(0, unbiased=False), atol=1e-12)
    # ---- gradient synchronisation: SUM over ranks of the local contributions, block ranges first, the rest in finish()
    from ssl_audio_amd import engine
    from ssl_audio_amd.train import GradSync
    import weakref

    class Flat:
        pass

    flat = Flat()
    flat.n_train = 100
    flat.grads = torch.arange(100, dtype=torch.float32) * (rank + 1)
    params = [torch.nn.Parameter(torch.zeros(10)) for _ in range(3)]
    for k, p in enumerate(params):                          # three "block" parameters living at [20,30), [30,40), [60,70)
        off = [20, 30, 60][k]
        engine.GRAD_SINK[id(p)] = (weakref.ref(p), flat.grads[off:off + 10])
    sync = GradSync(flat)
    sync.block_done(params[:2])                             # one contiguous run [20, 40)
    sync.block_done([params[2], params[0]])                 # scattered: [60, 70) and (again, idempotent ranges are NOT allowed) -> see below
    ok_sync = False
    # a block whose parameters are scattered must NOT sweep up what lies between them: gradients written there later
    # (by the remaining backward) have to be reduced exactly once, in finish()
    flat2 = Flat()
    flat2.n_train = 96
    flat2.grads = torch.zeros(96)
    q0, q1 = torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(8))
    engine.GRAD_SINK[id(q0)] = (weakref.ref(q0), flat2.grads[0:8])
    engine.GRAD_SINK[id(q1)] = (weakref.ref(q1), flat2.grads[80:88])
    flat2.grads[0:8] = rank + 1.0
    flat2.grads[80:88] = 10.0 * (rank + 1)
    sync2 = GradSync(flat2)
    sync2.block_done([q0, q1])
    flat2.grads[8:80] = 100.0 * (rank + 1)                  # "later layers" finish after the hook fired
    sync2.finish()
    tot = sum(range(1, world + 1))
    ok_sync = (torch.all(flat2.grads[0:8] == tot) and torch.all(flat2.grads[80:88] == 10.0 * tot)
               and torch.all(flat2.grads[8:80] == 100.0 * tot) and torch.all(flat2.grads[88:] == 0)).item()
    q.put((rank, ok, ok_lit, ok_bn, ok_sync, float(loss), float(lit_loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_protocol():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for r in res:
        assert not isinstance(r[5], str), r[5]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ok_lit, ok_bn, ok_sync, loss, lit in sorted(res):
        assert ok, f"rank {rank}: global-exact loss/grad mismatch"
        assert ok_lit, f"rank {rank}: literal-DDP quirk not reproduced (loss {loss}, literal {lit})"
        assert ok_bn, f"rank {rank}: SyncBN statistics combine"
        assert ok_sync, f"rank {rank}: gradient all-reduce ranges"
    assert abs(res[0][5] - res[1][5]) < 1e-12       # every rank holds the same global loss


def test_single_process_helpers():
    assert sdist.get_world_size() == 1 and sdist.get_rank() == 0 and sdist.is_main_process()
    t = torch.arange(6.0).reshape(2, 3)
    assert sdist.all_gather_rows(t).shape == (1, 2, 3)
    assert sdist.all_reduce_sum_(t) is t
    assert sdist.init_from_env() == (0, 0, 1) or "RANK" in os.environ
