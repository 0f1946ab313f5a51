"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU
for tests).  Only the exchanges SURVEY.md §8(e) lists exist:

  * BN statistics of the projector / predictor / loss  (all-gather of 2*C floats per rank, combined on device)
  * BN backward sums                                     (all-reduce of 2*C floats)
  * the D x D cross-correlation matrix                   (all-reduce SUM, utils/loss.py:20-21)
  * parameter gradients                                  (bucketed all-reduce SUM on a side stream, dp.py)

Semantics are *global-batch exact*: W ranks with B/W clips each compute what one process computes on B clips
(SURVEY.md F4); gradients are therefore SUMMED over ranks, not averaged.
"""
import os

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def init_from_env(backend=None):
    """torchrun-style initialisation (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, local_rank, world)."""
    if "RANK" not in os.environ:
        return 0, 0, 1
    rank, local, world = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ["WORLD_SIZE"])
    if not dist.is_initialized():
        backend = os.environ.get("SA_DIST_BACKEND", backend)    # e.g. gloo to rehearse several ranks on one GPU
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend)
    return rank, local, world


def collectives_active():
    """True when the exchanges must really be issued: more than one rank, or SA_DIST_FORCE=1 under an initialised process
    group (a one-rank RCCL rehearsal of every collective call site on a single GPU; results are unchanged)."""
    return get_world_size() > 1 or (os.environ.get("SA_DIST_FORCE") == "1" and is_dist_avail_and_initialized())


def all_gather_rows(t):
    """[*shape] -> [W, *shape] (W = 1 without a process group: a view, no copy)."""
    W = get_world_size()
    if not collectives_active():
        return t.unsqueeze(0)
    flat = t.contiguous().view(-1)
    out = torch.empty(W * flat.numel(), dtype=t.dtype, device=t.device)      # 1-D in / 1-D out: accepted by RCCL and gloo alike
    dist.all_gather_into_tensor(out, flat)
    return out.view((W,) + tuple(t.shape))


def all_reduce_sum_(t):
    if collectives_active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
