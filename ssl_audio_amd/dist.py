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
import torch.nn as nn


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def _ipc_env():
    """RCCL across PROCESSES shares device buffers through IPC handles, and this host driver stack only supports the dmabuf form: without
    HSA_ENABLE_IPC_MODE_LEGACY=0 the first communicator fails with `hipIpcGetMemHandle: invalid argument`.  The variable is read when the
    HSA runtime initialises, i.e. at the first GPU call of the process, so it is set (if unset) when this module is imported and again in
    init_from_env -- both ahead of any GPU call of the package; a launcher's or user's own value is never overridden."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


_ipc_env()


def init_from_env(backend=None):
    """torchrun-style initialisation (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), the counterpart of utils.init_distributed_mode
    (utils/utils.py:335-361).  Returns (rank, local_rank, world).  Works the same under an external `torch.distributed.run` as under
    bench.py's own child launcher: nothing here relies on the launcher having prepared the environment beyond torchrun's variables."""
    if "RANK" not in os.environ:
        return 0, 0, 1
    _ipc_env()
    rank, local, world = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ["WORLD_SIZE"])
    if not dist.is_initialized():
        backend = os.environ.get("SA_DIST_BACKEND", backend)    # e.g. gloo to rehearse several ranks on one GPU
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            local_dev = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_dev)
            # bind the communicator to this rank's device at creation: without device_id torch guesses the device from the GLOBAL rank
            # at the first collective ("Guessing device ID based on global rank. This can cause a hang ...")
            kw["device_id"] = torch.device("cuda", local_dev)
            if RCCL_CUS > 0:                # see reserve_cus_for_collectives
                os.environ.setdefault("NCCL_MAX_NCHANNELS", str(RCCL_CUS))
        dist.init_process_group(backend=backend, **kw)
    return rank, local, world


# RCCL's collective kernels hold one CU per channel for the length of an all-reduce, and a CU cannot host one of them next to
# a 248-VGPR GEMM workgroup.  A persistent GEMM grid sized to the whole chip that overlaps an all-reduce therefore finishes its
# last workgroups late (estimated: one ~290 us launch per gradient bucket delayed by up to the ~250 us all-reduce, ~6 % of a
# step).  SA_RCCL_CUS=n reserves n CUs instead (RCCL bounded to n channels, persistent grids and the split-K heuristic sized
# to the rest): measured cost 7.5 % of a step through tile-round quantisation (3012 tiles: 12 rounds on 256 CUs, 13 on 240),
# i.e. no better, so the default is 0 (off).  The fix built in round 3 is dynamic tile hand-out in the persistent kernels
# (sa_set_dynamic_tiles, switched on below whenever collectives are active).
RCCL_CUS = int(os.environ.get("SA_RCCL_CUS", "0"))


def reserve_cus_for_collectives():
    """Called by the trainer (and by GradSumParallel) once the process group exists.  With collectives active the persistent GEMM
    kernels hand their tiles out dynamically (sa_set_dynamic_tiles): a workgroup whose CU was held by an RCCL kernel then draws fewer
    tiles instead of finishing a static share late.  SA_RCCL_CUS > 0 additionally reserves CUs (the static alternative, off by
    default)."""
    from . import ops
    if collectives_active() and torch.cuda.is_available() and os.environ.get("SA_GEMM_DYNAMIC", "1") != "0":
        ops.set_dynamic_tiles(True)
    if RCCL_CUS > 0 and collectives_active() and torch.cuda.is_available() and dist.get_backend() == "nccl":
        _, cus = ops.device_info()
        ops.set_cu_budget(max(cus - RCCL_CUS, cus // 2))


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


class GradSumParallel(nn.Module):
    """What DistributedDataParallel is to the reference's drivers (utils/utils.py:412-416, applied main_bt_byol.py:440-444): the
    wrapper `model_setup_ddp` returns.  `.module` is the wrapped network (state_dict keys carry the same "module." prefix DDP
    gives the reference's checkpoints, main_bt_byol.py:494); forward delegates.  After every backward pass the gradients of
    the wrapped parameters are SUMMED over ranks -- summed, not averaged: losses here are global-batch exact (every rank holds
    the loss of the global batch and back-propagates its own rows' share of it).

    Mechanics: the trainable parameters are assigned ONCE, in reverse registration order (the order a backward pass finishes them
    in), to buckets of `bucket_bytes`; each bucket owns a flat fp32 buffer and every parameter's `.grad` is (re)pointed at its slice
    of it, so a bucket is all-reduced IN PLACE -- no flatten copy before the collective and no scatter copy after it (a gradient that
    autograd allocated afresh, e.g. after `zero_grad(set_to_none=True)`, costs one copy into its slice).  A post-accumulate hook per
    parameter counts arrivals; a complete bucket is reduced asynchronously on a side stream (RCCL) while the backward goes on; a
    callback queued on the autograd engine for the end of the pass reduces what is left (buckets that stayed incomplete because some
    parameter received no gradient) and joins the stream.  Static buckets make the collective sequence identical on every rank
    whatever order the hooks fire in.  One backward per optimiser step, like DDP without no_sync(): a second backward before the
    step would all-reduce the first one's (already summed) gradients again (x W) and raises instead.  "Before the step" is tracked
    per wrapper: the end of a reduced backward marks the bucket views as holding rank-summed gradients; the mark is dropped by any
    `torch.optim.Optimizer.step()` of an optimiser that holds one of the wrapper's parameters (a global post-step hook), by `grads_consumed()` for hand-written update loops, and per parameter
    when its `.grad` was replaced (`zero_grad(set_to_none=True)`, the default).
    """

    _live = None                                    # weak set of wrappers the global optimiser-step hook clears

    def __init__(self, module, bucket_bytes=64 << 20):
        super().__init__()
        self.module = module
        self.bucket_bytes = bucket_bytes
        self._stream = None
        self._buckets = []                          # [params], in reduction order
        self._where = {}                            # id(p) -> (bucket index, element offset)
        cur, cur_bytes = [], 0
        for p in reversed([q for q in module.parameters() if q.requires_grad]):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self._buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._buckets.append(cur)
        for b, params in enumerate(self._buckets):
            off = 0
            for p in params:
                self._where[id(p)] = (b, off)
                off += (p.numel() + 3) // 4 * 4     # 16-byte aligned slices
                p.register_post_accumulate_grad_hook(self._on_grad)
        self._flat = [None] * len(self._buckets)    # allocated on the first backward (the parameters' device is final by then)
        self._holds_sum = False                     # bucket views hold rank-summed gradients no optimiser step has consumed yet
        self._reset()
        if GradSumParallel._live is None:
            import weakref
            from torch.optim.optimizer import register_optimizer_step_post_hook
            GradSumParallel._live = weakref.WeakSet()
            register_optimizer_step_post_hook(GradSumParallel._after_optimizer_step)
        GradSumParallel._live.add(self)
        reserve_cus_for_collectives()

    @staticmethod
    def _after_optimizer_step(optimizer, *args, **kwargs):
        """Global post-step hook: the step of `optimizer` consumes the summed gradients of the wrappers whose parameters IT holds -- a step
        of an unrelated optimiser (another model, a probe, the predictor's own optimiser) leaves the other wrappers' guard armed
        (any step in the process used to clear every wrapper's mark)."""
        stepped = {id(p) for g in optimizer.param_groups for p in g["params"]}
        for w in list(GradSumParallel._live):
            if any(pid in stepped for pid in w._where):
                w.grads_consumed()

    def grads_consumed(self):
        """The summed gradients have been applied (called by the global optimiser-step hook; call it yourself after a hand-written
        parameter update that is not a `torch.optim.Optimizer`)."""
        self._holds_sum = False

    def _reset(self):
        self._arrived = [0] * len(self._buckets)
        self._seen = set()
        self._launched = [False] * len(self._buckets)
        self._inflight = []
        self._armed = False

    def forward(self, *args, **kwargs):
        if self._armed or self._inflight:           # a backward that raised never reached _finish: drop its half-built state
            for work in self._inflight:
                try:
                    work.wait()
                except Exception:
                    pass
            self._reset()
        return self.module(*args, **kwargs)

    def _bucket_buffer(self, b):
        if self._flat[b] is None:
            params = self._buckets[b]
            n = sum((p.numel() + 3) // 4 * 4 for p in params)
            self._flat[b] = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        return self._flat[b]

    def _on_grad(self, p):
        if not collectives_active() or p.grad is None:
            return
        if not self._armed:
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
            self._armed = True
        if id(p) in self._seen:
            raise RuntimeError("GradSumParallel: a parameter received a second gradient before the end of the backward pass was "
                               "reached -- one backward per optimiser step (the first pass's gradients are already summed over ranks)")
        self._seen.add(id(p))
        b, off = self._where[id(p)]
        flat = self._bucket_buffer(b)
        view = flat[off:off + p.numel()].view_as(p)
        if self._holds_sum and p.grad.data_ptr() == view.data_ptr():
            # autograd accumulated this pass's local gradient INTO the previous pass's rank-summed one: reducing the bucket again
            # would count the old sum W times
            raise RuntimeError("GradSumParallel: second backward before the optimiser step -- the gradients of the previous pass are "
                               "already summed over ranks and would be all-reduced again (gradient accumulation is not supported: "
                               "one backward per step; after a hand-written update call .grads_consumed()).  The gradients are INVALID "
                               "now (autograd has already added this pass's local gradient into the rank-summed one): zero_grad() "
                               "before the next backward, do not step on them")
        if p.grad.data_ptr() != view.data_ptr():    # autograd allocated this gradient: move it into the bucket, keep the view
            view.copy_(p.grad)
            p.grad = view
        self._arrived[b] += 1
        if self._arrived[b] == len(self._buckets[b]):
            self._launch(b)

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        flat = self._bucket_buffer(b)
        for q in self._buckets[b]:                  # a parameter without a gradient this pass contributes zeros (its slice may be stale)
            if id(q) not in self._seen:
                _, off = self._where[id(q)]
                flat[off:off + q.numel()].zero_()
        if flat.is_cuda and dist.get_backend() == "nccl":
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._inflight.append(work)

    def _finish(self):
        try:
            for b in range(len(self._buckets)):     # every rank reduces every bucket, in bucket order
                self._launch(b)
            for work in self._inflight:
                work.wait()
            if self._stream is not None:
                torch.cuda.current_stream().wait_stream(self._stream)
            self._holds_sum = True
        finally:
            self._reset()
