"""The pre-training step of the hot path, MI355X-first: the counterpart of train_one_epoch
(main_bt_byol.py:40-166) and of main.py:86-119, with everything that the reference leaves on the host moved
onto the GPU and everything launch-bound flattened:

  waveforms (HBM) --sa_logmel_fwd--> ring of normalised log-mels --sa_augment_views--> 2 views
     --> encoder + projector (engine.py) --> Barlow Twins loss (global-batch exact) --> backward
     --> gradient all-reduce per transformer block on a side stream (RCCL) overlapped with the remaining backward
     --> ONE fused AdamW launch per weight-decay group over flat fp32 state (+ bf16 weight refresh in the same pass)
     --> (BYOL-ish variant) ONE EMA launch over the flat parameter buffer.

Parameters, gradients and Adam moments live in flat buffers (`FlatState`); the nn.Module parameters are views
into them, so state_dict()/load_state_dict() keep the reference's key names.
"""
import math
import weakref

import torch
import torch.nn as nn

from . import dist as sdist
from . import engine, ops
from .augmentations import BatchedPairAugment, NormalizeBatch
from .frontend import MelSpectrogram
from .loss import BarlowTwinsLoss
from .model import BarlowTwinsHead, BarlowTwinsPredictor, ModelWrapper
from .utils import MultiCropWrapper

AUDIOSET_STATS = (-0.8294, 4.6230)   # main_bt_byol.py:288


class FlatState:
    """Flat fp32 parameter / gradient / Adam-moment buffers for a set of modules.

    Layout: [decayed params | un-decayed params | frozen params], each parameter padded to 8 elements so every view
    is 32-byte aligned.  get_param_groups' rule (utils/utils.py:136-147): no decay on '.bias' names and 1-D tensors.
    """

    def __init__(self, named_params, device):
        decay, nodecay, frozen = [], [], []
        for name, p in named_params:
            if not p.requires_grad:
                frozen.append((name, p))
            elif name.endswith(".bias") or p.dim() == 1:
                nodecay.append((name, p))
            else:
                decay.append((name, p))
        self.order = decay + nodecay + frozen
        pad8 = lambda n: (n + 7) // 8 * 8
        # layout entries: (name, parameter, elements) -- parameter None = a zero pad.  Each attention's q_bias and v_bias are placed
        # around a zero pad of the same length, so that [q_bias | 0 | v_bias] IS the qkv GEMM's bias vector (k-bias fixed at zero,
        # models/mae.py:125-128) and no per-call assembly is needed.  The pad has zero gradient and no weight decay: AdamW leaves it 0.
        by_name = dict(nodecay)
        taken, nodecay_layout = set(), []
        for name, p in nodecay:
            if name in taken:
                continue
            sib = name[:-len("q_bias")] + "v_bias" if name.endswith(".q_bias") else None
            if sib in by_name and p.numel() % 8 == 0 and by_name[sib].numel() == p.numel():
                nodecay_layout += [(name, p, p.numel()), (None, None, p.numel()), (sib, by_name[sib], p.numel())]
                taken.add(sib)
            else:
                nodecay_layout.append((name, p, p.numel()))
        layout = [(n, p, p.numel()) for n, p in decay] + nodecay_layout + [(n, p, p.numel()) for n, p in frozen]
        self.n_decay = sum(pad8(p.numel()) for _, p in decay)
        self.n_train = self.n_decay + sum(pad8(n) for _, _, n in nodecay_layout)
        total = self.n_train + sum(pad8(p.numel()) for _, p in frozen)
        self.params = torch.zeros(total, device=device)
        self.params_bf16 = torch.zeros(total, dtype=torch.bfloat16, device=device)
        self.grads = torch.zeros(self.n_train, device=device)
        self.m = torch.zeros(self.n_train, device=device)
        self.v = torch.zeros(self.n_train, device=device)
        self.offsets = {}
        self.qkv_fused = {n for n, _, _ in nodecay_layout if n is not None and n.endswith(".q_bias") and n[:-len("q_bias")] + "v_bias" in taken}
        off = 0
        for name, p, n in layout:
            if p is not None:
                self.params[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.params[off:off + n].view(p.shape)
                self.offsets[name] = (off, n)
                if off < self.n_train:
                    engine.register_grad_sink(p, self.grads[off:off + n].view(p.shape), self)
                if name in self.qkv_fused:
                    engine.QKV_BIAS[id(p)] = (weakref.ref(p), self.params[off:off + 3 * n])
                    engine._forget_when_dead(p, engine.QKV_BIAS)
            off += pad8(n)
        ops.cast_bf16(self.params, self.params_bf16)
        self.bind_bf16()
        self.step_count = 0
        self._param_ids = frozenset(id(p) for _, p in self.order)
        self.epoch = [0]                                  # bumped whenever a kernel rewrites this state's weights (engine.PARAM_EPOCH)
        engine.register_epoch([p for _, p in self.order], self.epoch)
        # {lr, 1/(1-b1^t), 1/sqrt(1-b2^t)} per parameter group, read by the AdamW launches from DEVICE memory: what changes from step to
        # step is data, not a kernel argument, so a captured step (HIP graph) replays with the current values.  Staged through pinned host
        # memory by one small async copy per step.
        self.hyper = torch.zeros(2, 4, device=device)
        self.sync = None           # the GradSync that reduces self.grads over ranks (engine.block_done_hook finds it through the sinks)

    def bind_bf16(self):
        """Point the engine's bf16 weight cache at views of the flat bf16 buffer (kept fresh by the AdamW kernel)."""
        for name, p in self.order:
            off, n = self.offsets[name]
            w = self.params_bf16[off:off + n]
            w = w.view(p.shape[0], -1) if p.dim() > 1 else w
            engine.BF16_WEIGHTS.pin(p, w)

    def zero_grad(self):
        self.grads.zero_()

    def stage_hyper(self, lr, betas=(0.9, 0.999), lr_nodecay=None):
        """Host side of one optimiser step, OUTSIDE any captured region: advance the step count and send this step's learning rates and
        bias corrections to the device vector the AdamW launches read."""
        self.step_count += 1
        t = self.step_count
        c1, c2 = 1.0 - betas[0] ** t, 1.0 - betas[1] ** t
        # a FRESH pinned staging block per step: the copy is asynchronous and the host runs steps ahead of the stream, so one reused
        # block would be overwritten with step t + 1's values before step t's copy has read it (the caching host allocator keeps a
        # block away from reuse until the copy that reads it has completed)
        h = torch.tensor([[lr, 1.0 / c1, 1.0 / math.sqrt(c2), 0.0],
                          [lr if lr_nodecay is None else lr_nodecay, 1.0 / c1, 1.0 / math.sqrt(c2), 0.0]], dtype=torch.float32)
        if self.hyper.is_cuda:
            h = h.pin_memory()
        self.hyper.copy_(h, non_blocking=True)

    def adamw_staged(self, wd, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, skip_flag=None):
        """The two AdamW launches (decayed / un-decayed range) with their per-step scalars read from self.hyper (stage_hyper): capturable.
        skip_flag: device word; non-zero = leave everything untouched (non-finite loss, main_bt_byol.py:116-118)."""
        nd, nt = self.n_decay, self.n_train
        if nd:
            ops.adamw_step_dev(self.params[:nd], self.grads[:nd], self.m[:nd], self.v[:nd], self.hyper[0], betas[0], betas[1], eps, wd, grad_scale,
                               self.params_bf16[:nd], skip_flag)
        if nt > nd:
            ops.adamw_step_dev(self.params[nd:nt], self.grads[nd:nt], self.m[nd:nt], self.v[nd:nt], self.hyper[1], betas[0], betas[1], eps, 0.0,
                               grad_scale, self.params_bf16[nd:nt], skip_flag)
        self.epoch[0] += 1
        engine.BF16_WEIGHTS_T.refresh_all(self._param_ids)   # this state's transposed weight copies (the dgrad operands) in one launch

    def adamw(self, lr, wd, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, lr_nodecay=None):
        """One AdamW step: a launch for the decayed range (lr, wd) and one for the un-decayed range (lr_nodecay or lr, no decay) -- the
        two parameter groups of utils.get_param_groups, each with the learning rate its group holds."""
        self.step_count += 1
        nd, nt = self.n_decay, self.n_train
        if nd:
            ops.adamw_step(self.params[:nd], self.grads[:nd], self.m[:nd], self.v[:nd], lr, betas[0], betas[1], eps, wd, self.step_count,
                           grad_scale, self.params_bf16[:nd])
        if nt > nd:
            ops.adamw_step(self.params[nd:nt], self.grads[nd:nt], self.m[nd:nt], self.v[nd:nt], lr if lr_nodecay is None else lr_nodecay, betas[0], betas[1], eps, 0.0,
                           self.step_count, grad_scale, self.params_bf16[nd:nt])
        self.epoch[0] += 1                    # copies derived from the bf16 weights (transposed dgrad operands) are stale now
        engine.BF16_WEIGHTS_T.refresh_all(self._param_ids)   # ... and rebuilt here, all of this state's in one launch

    # ------------------------------------------------------------------ checkpoint / resume (main_bt_byol.py:492-503, utils/utils.py:37-46)
    def _trainable(self):
        return [(n, p) for n, p in self.order if p.requires_grad]      # decayed first, then un-decayed: utils.get_param_groups' order

    def optim_state_dict(self, lr, wd, betas=(0.9, 0.999), eps=1e-8, index_base=0):
        """The Adam moments in `torch.optim.AdamW.state_dict()` form over `utils.get_param_groups(module)`'s parameter order: two groups
        (decayed, un-decayed), parameter indices counted from `index_base`.  main.py's optimiser is exactly this dictionary; main_bt_byol.py's
        get_optimizer (:301-305) concatenates the encoder's and the predictor's groups -- `merge_optim_state_dicts` builds that form."""
        state, idx = {}, index_base
        groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=wd, amsgrad=False, maximize=False, foreach=None, capturable=False,
                       differentiable=False, fused=None, params=[]) for _ in range(2)]
        groups[1]["weight_decay"] = 0.0
        for n, p in self._trainable():
            off, cnt = self.offsets[n]
            state[idx] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[off:off + cnt].view(p.shape).clone(),
                          "exp_avg_sq": self.v[off:off + cnt].view(p.shape).clone()}
            groups[0 if off < self.n_decay else 1]["params"].append(idx)
            idx += 1
        return {"state": state, "param_groups": groups}

    def load_optim_state_dict(self, sd, first_group=0):
        """Inverse of optim_state_dict: reads the two parameter groups `first_group`, `first_group + 1` of a torch.optim.AdamW state dict
        (indices are consecutive over the groups in order, as torch numbers them)."""
        named = self._trainable()
        groups = sd["param_groups"]
        if len(groups) < first_group + 2:
            raise ValueError(f"optimizer state has {len(groups)} parameter groups, need groups {first_group} and {first_group + 1}")
        base = sum(len(g["params"]) for g in groups[:first_group])
        n_saved = len(groups[first_group]["params"]) + len(groups[first_group + 1]["params"])
        if n_saved != len(named):
            raise ValueError(f"optimizer state groups {first_group}-{first_group + 1} hold {n_saved} parameters, the module has {len(named)} trainable ones")
        steps = set()
        self.m.zero_()
        self.v.zero_()
        for i, (n, p) in enumerate(named):
            st = sd["state"].get(base + i)
            if st is None:                                   # a parameter that never received a gradient has no state entry
                continue
            off, cnt = self.offsets[n]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {base + i} has shape {tuple(st['exp_avg'].shape)}, parameter {n} has {tuple(p.shape)}")
            self.m[off:off + cnt].copy_(st["exp_avg"].reshape(-1).to(self.m.device, torch.float32))
            self.v[off:off + cnt].copy_(st["exp_avg_sq"].reshape(-1).to(self.v.device, torch.float32))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused AdamW keeps one count")
        self.step_count = steps.pop() if steps else 0

    def refresh_from_parameters(self):
        """After load_state_dict wrote the (flat-backed) parameters: re-cast the bf16 copies the GEMMs read."""
        ops.cast_bf16(self.params, self.params_bf16)
        self.epoch[0] += 1

    def ema_from(self, other, beta, skip_flag=None):
        """self = beta * self + (1 - beta) * other over all parameters (utils/utils.py:328-331), one launch (a no-op while the device word
        skip_flag is non-zero)."""
        assert self.params.numel() == other.params.numel()
        if skip_flag is not None:
            ops.ema_update_gated(self.params, other.params, beta, skip_flag)
        else:
            ops.ema_update(self.params, other.params, beta)
        ops.cast_bf16(self.params, self.params_bf16)
        self.epoch[0] += 1


def merge_optim_state_dicts(*sds):
    """Concatenate torch-style optimiser state dicts whose parameter indices already follow one another (optim_state_dict's index_base):
    the dictionary `torch.optim.AdamW(groups_a + groups_b + ...).state_dict()` would hold."""
    out = {"state": {}, "param_groups": []}
    for sd in sds:
        out["state"].update(sd["state"])
        out["param_groups"] += sd["param_groups"]
    return out


def _empty_groups(lr, wd, betas=(0.9, 0.999), eps=1e-8):
    """The two (empty) groups get_param_groups yields for a parameter-free predictor (BarlowTwinsPredictor(use=False) is an Identity)."""
    g = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=wd, amsgrad=False, maximize=False, foreach=None, capturable=False,
              differentiable=False, fused=None, params=[]) for _ in range(2)]
    g[1]["weight_decay"] = 0.0
    return {"state": {}, "param_groups": g}


class GradSync:
    """Gradient SUM all-reduce over ranks, bucketed per transformer block and overlapped with backward on a side
    HIP stream (collective site C3).  Buckets are contiguous ranges of FlatState.grads.

    grad_dtype "bf16" (default: $SA_GRAD_DTYPE, else "fp32"): a range travels as bf16 -- cast into a bf16 staging buffer on the side
    stream (sa_cast_f32_to_bf16), all-reduced there at half the bytes (187 instead of 374 MB per step and GPU for ViT-B; xGMI rings are
    per-link bound), widened back into the fp32 gradient buffer (sa_cast_bf16_to_f32) -- the moments and the update stay fp32.  The
    reference's DDP does the same under its fp16 autocast (gradient buckets in the parameters' reduced type); here it is an option
    because the sum of W bf16-rounded shares differs from the fp32 sum by ~1e-3 relative (tests/test_dp_gloo.py bounds it)."""

    def __init__(self, flat, bucket_bytes=64 << 20, min_block_bytes=1 << 20, grad_dtype=None):
        import os
        grad_dtype = grad_dtype or os.environ.get("SA_GRAD_DTYPE", "fp32")
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError(f"grad_dtype must be 'fp32' or 'bf16', got {grad_dtype!r}")
        self.grad_dtype = grad_dtype
        self.flat = flat
        self.world = sdist.get_world_size()
        self.active = sdist.collectives_active()
        self.stream = torch.cuda.Stream() if (self.active and flat.grads.is_cuda) else None
        self.pending = []
        self.bucket_bytes = bucket_bytes
        self._stage = None         # bf16 image of flat.grads' trainable range (grad_dtype bf16), allocated at the first exchange
        # a block's un-decayed vectors (LayerNorm weights, biases: a few KB) are NOT worth a collective of their own -- twelve
        # latency-bound all-reduces per step, each holding CUs next to the backward's GEMMs: runs below this size are left to finish(),
        # where the whole un-decayed region (contiguous in the flat buffer) goes out as one range
        self.min_block_bytes = min_block_bytes
        self._ready_ranges = []

    def block_done(self, params):
        """engine.block_done_hook(p) of this flat state: the gradients of `params` are final -> reduce their flat range now."""
        if not self.active:
            return
        spans = []
        for p in params:
            ent = engine.GRAD_SINK.get(id(p))
            if ent is None or ent[0]() is not p:
                continue
            g = ent[1]
            a = (g.data_ptr() - self.flat.grads.data_ptr()) // 4
            fused = engine.QKV_BIAS.get(id(p))                       # [q_bias | zero pad | v_bias]: one run, pad included
            n = 3 * g.numel() if (fused is not None and fused[0]() is p) else g.numel()
            spans.append((a, a + (n + 7) // 8 * 8))
        # a block's parameters sit in two places of the flat buffer (decayed weights | un-decayed vectors): reduce each
        # contiguous run on its own -- a single min..max range would sweep up other layers' unfinished gradients
        spans.sort()
        run = None
        runs = []
        for a, b in spans:
            if run is not None and a <= run[1]:
                run = (run[0], max(run[1], b))
            else:
                if run is not None:
                    runs.append(run)
                run = (a, b)
        if run is not None:
            runs.append(run)
        for lo, hi in runs:
            hi = min(hi, self.flat.n_train)
            if (hi - lo) * 4 >= self.min_block_bytes:        # (small runs: reduced once, with their neighbours, by finish())
                self._launch(lo, hi)

    def _launch(self, lo, hi):
        self._ready_ranges.append((lo, hi))
        view = self.flat.grads[lo:hi]
        if self.grad_dtype == "bf16":
            if lo % 8:                                                # (FlatState pads every parameter to 8 elements: ranges start 16-byte aligned in bf16)
                raise RuntimeError(f"GradSync: gradient range [{lo}, {hi}) does not start on an 8-element boundary")
            if self._stage is None:
                self._stage = torch.empty(self.flat.n_train, dtype=torch.bfloat16, device=self.flat.grads.device)
            stage = self._stage[lo:hi]
        if self.stream is None:
            if self.grad_dtype == "bf16":
                ops.cast_bf16(view, stage)
                sdist.all_reduce_sum_(stage)
                ops.cast_f32_from_bf16(stage, view)
            else:
                sdist.all_reduce_sum_(view)
            return
        ev = torch.cuda.Event()
        ev.record()
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            if self.grad_dtype == "bf16":
                # cast -> all-reduce -> widen, all ordered on the side stream (work.wait() makes the side STREAM wait, not the host); the
                # next range's cast queues behind this range's collective, which RCCL would serialise anyway
                ops.cast_bf16(view, stage)
                work = torch.distributed.all_reduce(stage, async_op=True)
                work.wait()
                ops.cast_f32_from_bf16(stage, view)
            else:
                self.pending.append(torch.distributed.all_reduce(view, async_op=True))

    def finish(self):
        """Reduce whatever no block hook covered (head, cls token, final norm, ...) and join the side stream."""
        if not self.active:
            return
        n = self.flat.n_train
        gaps, cur = [], 0
        for lo, hi in sorted(self._ready_ranges) + [(n, n)]:      # complement of what the block hooks already reduced
            if lo > cur:
                gaps.append((cur, lo))
            cur = max(cur, hi)
        for lo, hi in gaps:
            self._launch(lo, hi)
        for w in self.pending:
            w.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.pending.clear()
        self._ready_ranges.clear()


class BarlowTwinsTrainer:
    """One object = the whole step.  mode='bt': single network, two views, one loss term (main.py:86-119, BASELINE
    configs 2-3).  mode='byol': online (+predictor) / EMA target, two cross terms (main_bt_byol.py --stop_gradient
    --predictor, BASELINE config 4).  mode='mae': main.py:69-125 with `--mask --mask_ratio r --masked_recon` (BASELINE config 5):
    view 1 goes through the masked encoder + MAE decoder (teacher side, adds the reconstruction loss), view 2 through the
    unmasked encoder, one BT term between them."""

    def __init__(self, cfg, device, mode="bt", batch_per_rank=None, clip_samples=160000, seed=0, from_waveform=True,
                 ema_beta=0.99, grad_dtype=None):
        self.cfg, self.device, self.mode = cfg, device, mode
        self.world = sdist.get_world_size()
        self.B = batch_per_rank or cfg.batch_size // self.world
        self.clip_samples = clip_samples
        self.from_waveform = from_waveform
        torch.manual_seed(seed)
        self.online = MultiCropWrapper(ModelWrapper(cfg), BarlowTwinsHead(cfg, _feature_dim(cfg))).to(device)
        self.flat = FlatState(list(self.online.named_parameters()), device)
        # local crops (main.py:86-119 with ncrops = L + 2): the teacher sees global view 1, the student global view 2 and the L local views
        self.L = int(getattr(cfg, "local_crops_number", 0) or 0)
        if self.L and mode == "byol":
            raise NotImplementedError("local crops with the two-network form: main_bt_byol.py:96-107 chunks the online output (2 crops) into "
                                      "L + 2 pieces, which only lines up for L = 0")
        self.criterion = BarlowTwinsLoss(cfg, ncrops=self.L + 2).to(device)
        sdist.reserve_cus_for_collectives()
        self.sync = self.flat.sync = GradSync(self.flat, grad_dtype=grad_dtype)
        self.predictor = self.target = self.flat_pred = self.flat_target = None
        if mode == "byol":
            self.predictor = BarlowTwinsPredictor(cfg.projector_out_dim, use=True).to(device)
            self.flat_pred = FlatState(list(self.predictor.named_parameters()), device)
            self.sync_pred = self.flat_pred.sync = GradSync(self.flat_pred, grad_dtype=grad_dtype)
            self.target = MultiCropWrapper(ModelWrapper(cfg), BarlowTwinsHead(cfg, _feature_dim(cfg))).to(device)
            self.target.load_state_dict(self.online.state_dict())
            for p in self.target.parameters():
                p.requires_grad = False
            self.flat_target = _FrozenFlat(self.target, self.flat, device)
            self.ema_beta = ema_beta
        self.frontend = MelSpectrogram(cfg.sample_rate, cfg.n_fft, cfg.win_length, cfg.hop_length, cfg.n_mels, cfg.f_min, cfg.f_max)
        frames = self.frontend.n_frames(clip_samples) if from_waveform else cfg.crop_frames
        self.frames = frames
        self.augment = BatchedPairAugment(device, cfg.n_mels, cfg.crop_frames, cfg.crop_frames, cfg.mixup, cfg.RRC, cfg.RLF,
                                          cfg.mixup_ratio, virtual_crop_scale=tuple(cfg.virtual_crop_scale), seed=seed + 1000 * sdist.get_rank(),
                                          local_crops_number=self.L, local_crops_size=tuple(cfg.local_crops_size), gnoise=bool(cfg.Gnoise))
        self.post_norm = NormalizeBatch() if cfg.post_norm else None
        self.wd = cfg.wd
        # What `optimizer.param_groups` is to the reference's loop: utils.adjust_learning_rate(args, trainer, loader, iteration) --
        # main.py:52's call with this object in the optimiser's place -- writes the step's learning rates here, in get_optimizer's group
        # order (encoder decayed / un-decayed, then the predictor's two groups); the fused AdamW launches read them.
        self.param_groups = [{"lr": cfg.lr, "weight_decay": cfg.wd}, {"lr": cfg.lr, "weight_decay": 0.0}]
        if mode == "byol":
            self.param_groups += [{"lr": cfg.lr, "weight_decay": cfg.wd}, {"lr": cfg.lr, "weight_decay": 0.0}]
        self._graph = self._graph_views = self._graph_loss = None
        self._graph_mask_ratio = None                    # the mask ratio a captured step was recorded with (None: unmasked)
        self.use_graph = True                            # False: run eagerly although a graph exists (profiling passes with per-launch events)
        self.mask_ratio_schedule = None                  # per-iteration table (utils.sine_scheduler_increase, main.py:442), or None
        self.last_loss = None
        # main_bt_byol.py:116-118 stops on a non-finite loss with a host sync every step; here the test is a device-side counter that
        # the host reads every `finite_check_every` steps (and whenever assert_finite() is called)
        self._nonfinite = torch.zeros(1, dtype=torch.int32, device=device)
        self.finite_check_every = 100
        self._steps = 0

    @property
    def lr(self):
        """The encoder's decayed-group learning rate, i.e. what the AdamW launches use (`param_groups` is the single source)."""
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        for g in self.param_groups:
            g["lr"] = value

    # ------------------------------------------------------------------ data path
    def make_views(self, batch, lengths=None):
        """batch: waveforms [B, L] (from_waveform) or log-mels [B, 1, F, T_any]; -> [view1, view2] each [B,1,F,crop_frames].

        What `Dataset.__getitem__` does per sample (datasets.py:342-358) happens here per clip, in the reference's RNG order: a clip with
        more frames than cfg.crop_frames is cropped at its own `np.random.randint(l - crop_frames)`, a shorter one is right-padded, then the
        clip's views are drawn.  lengths (optional, host sequence of B ints): each clip's length in samples when the rows of `batch` are
        padded to a common length; default: every clip fills its row."""
        B = batch.shape[0]
        slots = self.augment.next_slots(B)
        if self.from_waveform:
            if lengths is not None:
                lengths = [int(n) for n in lengths]
                src_frames = [self.frontend.n_frames(n) for n in lengths]
            else:
                src_frames = self.frontend.n_frames(batch.shape[-1])
            drawn = self.augment.draw(B, src_frames=src_frames)          # crop starts first: the frontend launch needs them
            starts = self.augment.starts
            self.frontend(batch, crop_frames=self.cfg.crop_frames, start=starts if any(starts) else 0, norm_stats=AUDIOSET_STATS,
                          out=slots.view(B, 1, *slots.shape[1:]), lengths=lengths)
        else:
            slots.copy_(batch.view(B, *batch.shape[-2:]))
            drawn = None
        out = None
        if self._graph is not None and self.use_graph and self.post_norm is None and self._graph_views[0].shape[0] == B:
            out = self._graph_out()                          # the captured step's input buffers: the augmentation writes them in place
        views = self.augment(B, out=out, drawn=drawn)
        crops = [views[i] for i in range(2 + self.L)]        # [view1, view2, local_1 .. local_L] (utils/transforms.py:49-56, batched)
        if self.post_norm is not None:
            crops = [self.post_norm(c) for c in crops]       # main.py:60-65: NormalizeBatch per crop
        return crops

    def _graph_out(self):
        """[2, B, 1, F, T] tensor aliasing the two static view buffers when they are adjacent in memory (enable_graph allocates them so)."""
        return self._graph_pair

    # ------------------------------------------------------------------ one optimisation step
    def step(self, batch, iteration=None, loader_len=None, mask_ratio=None, lengths=None):
        """One optimisation step on a batch.  With `iteration` (the global training iteration, main.py:48) the per-iteration schedules of
        the reference's loop apply first: the learning rate (utils.adjust_learning_rate when cfg.lr_schedule, main.py:51-57; needs
        loader_len = iterations per epoch) and, in mode 'mae', the mask ratio (mask_ratio_for, main.py:71-81)."""
        if iteration is not None:
            self.apply_schedules(iteration, loader_len)
            if mask_ratio is None and (self.mode == "mae" or getattr(self.cfg, "mask", False)):
                mask_ratio = self.mask_ratio_for(iteration)
        return self.step_views(self.make_views(batch, lengths=lengths), mask_ratio=mask_ratio)

    def apply_schedules(self, iteration, loader_len):
        """main.py:51-57: `if args.lr_schedule: utils.adjust_learning_rate(args, optimizer, data_loader, iteration)`."""
        from . import utils
        if getattr(self.cfg, "lr_schedule", False):
            if loader_len is None:
                raise ValueError("the learning-rate schedule needs loader_len (iterations per epoch)")
            utils.adjust_learning_rate(self.cfg, self, range(loader_len), iteration)

    def mask_ratio_for(self, iteration):
        """The mask ratio main.py:71-81 picks for an iteration: the schedule table's entry, else (cfg.random_mask_ratio) 0 with
        probability 1/2 and U(0.05, cfg.mask_beta) otherwise, else cfg.mask_ratio; 0 when cfg.mask is off."""
        from . import utils
        if not getattr(self.cfg, "mask", True):
            return 0
        if self.mask_ratio_schedule is not None:
            return float(self.mask_ratio_schedule[iteration])
        if getattr(self.cfg, "random_mask_ratio", False):
            if getattr(self, "mode", None) == "byol":            # main_bt_byol.py:70-72: fixed U(0.02, 0.2)
                return utils.generate_random(l=0.02, h=0.2, p=0.5)
            return utils.generate_random(l=0.05, h=self.cfg.mask_beta, p=0.5)
        return self.cfg.mask_ratio

    def step_views(self, views, mask_ratio=None):
        """One optimisation step on two already-augmented views [B,1,F,T] (what train_one_epoch receives from its loader).  mask_ratio
        (mode 'mae'): this step's masking ratio, cfg.mask_ratio when None.

        The step is a host prologue (step counts, learning rates -> device, `stage`) followed by a device-only body (`_device_step`):
        forward, loss, backward, gradient all-reduce, AdamW [, EMA].  With `enable_graph()` the body is captured once into a HIP graph
        and replayed; the views are then copied into the graph's static input buffers."""
        g = self.param_groups
        self.flat.stage_hyper(g[0]["lr"], lr_nodecay=g[1]["lr"])
        if self.mode == "byol":
            self.flat_pred.stage_hyper(g[2]["lr"], lr_nodecay=g[3]["lr"])
        if self._graph is not None and self.use_graph:
            # the captured step holds ITS mask ratio (the token counts are baked into its launches): another ratio cannot be replayed
            if mask_ratio is not None and self._graph_mask_ratio is not None and float(mask_ratio) != float(self._graph_mask_ratio):
                raise RuntimeError(f"the captured step masks {self._graph_mask_ratio} of the tokens; a step at mask_ratio {mask_ratio} needs use_graph = False "
                                   "(per-iteration mask-ratio schedules change the launches' shapes)")
            for dst, src in zip(self._graph_views, views):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
            self._graph.replay()
            self.last_loss = self._graph_loss.clone()      # (the static tensor is overwritten by the next replay: callers keep their own)
        else:
            self.last_loss = self._device_step(views, mask_ratio)
        self._steps += 1
        if self._steps % self.finite_check_every == 0:
            self.assert_finite()
        return self.last_loss

    def _device_step(self, views, mask_ratio=None):
        """Everything of a step that runs on the device, with no host-side state that changes from step to step (capturable)."""
        self.flat.zero_grad()
        engine.reset_pending_backward()
        masked = self.mode == "mae" or bool(getattr(self.cfg, "mask", False))
        mr = (self.cfg.mask_ratio if mask_ratio is None else mask_ratio) if masked else 0
        recon_on = self.mode == "mae" or bool(getattr(self.cfg, "masked_recon", False))
        if self.mode == "bt" and self.L == 0 and not masked and not recon_on:
            z = self.online(views, ncrops=2)
            z1, z2 = z.chunk(2)
            loss = self.criterion.forward_loss(z1, z2)
        elif self.mode == "bt" and not masked and not recon_on:
            # main.py:86-119: teacher = model(images[:1], ncrops=1), student = model(images[1:], ncrops=L+1).  The two global views share
            # one encoder pass here (the encoder has no batch statistics and the head runs its BatchNorm per crop chunk, model.py:26-31,
            # in the same order: view 1, view 2, locals), the 16-wide local crops go through as their own width group
            z = self.online(views[:2], ncrops=2)
            t, s1 = z.chunk(2)
            sl = self.online(views[2:], ncrops=self.L)
            loss = self.criterion(torch.cat([s1, sl]), t, ngcrops_each=1)
        elif self.mode in ("bt", "mae"):
            # main.py:69-125 with `--mask` [--masked_recon]: ONLY the teacher's view is masked (and, with masked_recon, decoded: its
            # reconstruction loss is added, :119-122); mode 'mae' = these two flags forced on (BASELINE config 5)
            out = self.online(views[:1], ncrops=1, mask_ratio=mr, masked_recon=recon_on)
            t, recon = out if recon_on else (out, None)
            st = self.online(views[1:], ncrops=1 + self.L)
            loss = self.criterion(st, t, ngcrops_each=1)
            if recon is not None:
                loss = loss + recon
        else:
            # main_bt_byol.py:79-114: `--mask` masks BOTH views of the online encoder (:83-88), the target sees them whole (:97-101)
            self.flat_pred.zero_grad()
            out = self.online(views[:2], ncrops=2, mask_ratio=mr, masked_recon=recon_on) if (masked or recon_on) else self.online(views[:2], ncrops=2)
            o, recon = out if recon_on else (out, None)
            o = self.predictor(o, ncrops=1)
            with torch.no_grad():
                t = self.target(views, ncrops=2)
            loss = self.criterion(o, t, ngcrops_each=2)
            if recon is not None:
                loss = loss + recon                              # main_bt_byol.py:112-114
        ops.count_nonfinite(loss.detach().reshape(1), self._nonfinite)
        if self.mode == "byol":
            # before the optimiser step (main_bt_byol.py:121-126); behind the finite-loss gate like the optimiser (:116-118)
            self.flat_target.ema_from(self.flat, self.ema_beta, skip_flag=self._nonfinite)
        loss.backward()
        self.sync.finish()
        # a non-finite loss (counted just above) turns the optimiser launches into no-ops: the reference exits before optimizer.step()
        # (main_bt_byol.py:116-118); here the host reads the counter lazily, so the weights and moments are kept clean on the device
        self.flat.adamw_staged(self.wd, skip_flag=self._nonfinite)
        if self.mode == "byol":
            self.sync_pred.finish()
            self.flat_pred.adamw_staged(self.wd, skip_flag=self._nonfinite)
        return loss.detach()

    # ------------------------------------------------------------------ HIP graph
    def enable_graph(self, views=None):
        """Capture `_device_step` (encoder + projector forward, loss, backward, gradient all-reduce, AdamW [, EMA]) into ONE HIP graph that
        every later `step_views` / `step` replays: ~600 launches and ~150 allocator calls per step leave the host's critical path.  Call
        after at least one eager step (workspaces sized, transposed weight copies allocated, BatchNorm buffers touched).  What changes from
        step to step reaches the graph through device memory only: the views (static buffers, filled by the augmentation launch or a
        copy), the learning rates and Adam bias corrections (FlatState.hyper), the non-finite gate.  Host-sampled augmentation
        parameters, the frontend and the augmentation launch stay outside the graph.  Any failure of the capture raises -- there is no
        silent eager fallback."""
        if self._graph is not None:
            return
        if self._steps < 1:
            raise RuntimeError("enable_graph(): run at least one eager step first (workspaces and weight copies are allocated lazily)")
        masked = self.mode == "mae" or bool(getattr(self.cfg, "mask", False))
        if masked and (getattr(self.cfg, "random_mask_ratio", False) or self.mask_ratio_schedule is not None):
            raise NotImplementedError("graph capture with a mask ratio that changes from step to step (random_mask_ratio / a schedule table): the number "
                                      "of kept tokens is baked into the captured launches")
        # Random masking itself is capturable: the indices are device-side index bookkeeping (torch.rand from the device generator, whose
        # Philox offset a replay advances like an eager step, argsort, gather -- models/mae.py:332-339) and every token movement they drive
        # is a sa_* launch; with a fixed ratio all shapes are static.
        self._graph_mask_ratio = float(self.cfg.mask_ratio) if masked else None
        pair = torch.empty(2, self.B, 1, self.cfg.n_mels, self.cfg.crop_frames, device=self.device)
        if views is not None:
            pair[0].copy_(views[0]); pair[1].copy_(views[1])
        self._graph_pair = pair
        self._graph_views = [pair[0], pair[1]]
        if self.L:                                           # static buffers for the L local crops too (utils/transforms.py:38-47)
            lh, lw = tuple(self.cfg.local_crops_size)
            loc = torch.empty(self.L, self.B, 1, lh, lw, device=self.device)
            if views is not None:
                for l in range(self.L):
                    loc[l].copy_(views[2 + l])
            self._graph_views += [loc[l] for l in range(self.L)]
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._graph_loss = self._device_step(self._graph_views)
        self.last_loss = self._graph_loss
        self._graph = graph
        # the capture itself launched nothing; the host-side bookkeeping it advanced (weight epoch) describes the replayed step too

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self, epoch=0, driver="main_bt_byol"):
        """The dictionary main_bt_byol.py:492-498 saves ('model', 'optimizer', 'epoch', 'barlow_twins_loss'; the reference's drop of the
        predictor / target there is SURVEY.md A.5's quirk -- they are included here so that a byol run resumes exactly).

        'optimizer' is what the reference driver's own optimiser would save: get_optimizer (main_bt_byol.py:301-305) builds ONE AdamW over
        get_param_groups(encoder) + get_param_groups(predictor) -- four groups, the last two empty without a predictor -- and
        `driver="main"` gives main.py's two-group form.  Refuses to save after a non-finite loss (the reference stops before the optimiser
        step, main_bt_byol.py:116-118; here the flag is read lazily, so it is read now)."""
        self.assert_finite()
        opt = self.flat.optim_state_dict(self.lr, self.wd)
        if driver == "main_bt_byol":
            n_enc = sum(len(g["params"]) for g in opt["param_groups"])
            pred = self.flat_pred.optim_state_dict(self.lr, self.wd, index_base=n_enc) if self.mode == "byol" else _empty_groups(self.lr, self.wd)
            opt = merge_optim_state_dicts(opt, pred)
        elif self.mode == "byol":
            raise ValueError("driver='main' has no predictor; a byol trainer saves main_bt_byol.py's form")
        for saved, live in zip(opt["param_groups"], self.param_groups):       # the learning rates the schedule last set
            saved["lr"] = live["lr"]
        sd = {"model": self.online.state_dict(), "optimizer": opt, "epoch": epoch,
              "barlow_twins_loss": self.criterion.state_dict(), "steps": self._steps}
        if self.mode == "byol":
            sd["predictor"] = self.predictor.state_dict()
            sd["target"] = self.target.state_dict()
        return sd

    def load_state_dict(self, ckpt):
        """Resume from state_dict()'s dictionary, or from a checkpoint of the reference driver (utils/utils.py:37-46: 'model', 'optimizer'
        [, 'predictor'], 'epoch'; DDP's 'module.' prefix accepted).  Returns the epoch to continue with."""
        strip = lambda sd: {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.online.load_state_dict(strip(ckpt["model"]))
        self.flat.refresh_from_parameters()
        opt = ckpt.get("optimizer")
        if opt is not None:
            if len(opt["param_groups"]) not in (2, 4):
                raise ValueError(f"optimizer state with {len(opt['param_groups'])} parameter groups: expected main.py's 2 (encoder) or "
                                 "main_bt_byol.py's 4 (encoder + predictor, --stop_gradient)")
            self.flat.load_optim_state_dict(opt, 0)
            if len(opt["param_groups"]) == 4:
                n_pred = len(opt["param_groups"][2]["params"]) + len(opt["param_groups"][3]["params"])
                if self.mode == "byol":
                    self.flat_pred.load_optim_state_dict(opt, 2)
                elif n_pred:
                    raise ValueError(f"the checkpoint's optimizer holds {n_pred} predictor parameters; this trainer (mode '{self.mode}') has no predictor")
            for live, saved in zip(self.param_groups, opt["param_groups"]):
                live["lr"] = saved["lr"]
        if "barlow_twins_loss" in ckpt:
            self.criterion.load_state_dict(ckpt["barlow_twins_loss"])
        if self.mode == "byol":
            if "predictor" in ckpt:
                self.predictor.load_state_dict(strip(ckpt["predictor"]))
                self.flat_pred.refresh_from_parameters()
            if "optimizer_predictor" in ckpt:                      # (round-2 checkpoints kept the predictor's moments under their own key)
                self.flat_pred.load_optim_state_dict(ckpt["optimizer_predictor"])
            self.target.load_state_dict(strip(ckpt["target"]) if "target" in ckpt else self.online.state_dict())
            self.flat_target.refresh_from_parameters()
        self._steps = int(ckpt.get("steps", 0))
        return int(ckpt.get("epoch", 0))

    def assert_finite(self):
        """Raises FloatingPointError if any step since the last call produced a non-finite loss (one host read of the device flag)."""
        bad = int(self._nonfinite.item())
        if bad:
            self._nonfinite.zero_()
            raise FloatingPointError(f"Loss was not finite in {bad} step(s). Stopping training (main_bt_byol.py:116-118)")


class _FrozenFlat:
    """Flat parameter buffer of the EMA target with the SAME layout as the online FlatState (so EMA is one launch)."""

    def __init__(self, module, like, device):
        self.params = torch.zeros_like(like.params)
        self.params_bf16 = torch.zeros_like(like.params_bf16)
        named = dict(module.named_parameters())
        for name, _ in like.order:
            p = named[name]
            off, n = like.offsets[name]
            self.params[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.params[off:off + n].view(p.shape)
            w = self.params_bf16[off:off + n]
            engine.BF16_WEIGHTS.pin(p, w.view(p.shape[0], -1) if p.dim() > 1 else w)
            if name in like.qkv_fused:
                engine.QKV_BIAS[id(p)] = (weakref.ref(p), self.params[off:off + 3 * n])
                engine._forget_when_dead(p, engine.QKV_BIAS)
        ops.cast_bf16(self.params, self.params_bf16)
        self.epoch = [0]                                  # (its own counter: an EMA step leaves the online network's transposed copies fresh)
        engine.register_epoch([named[name] for name, _ in like.order], self.epoch)

    ema_from = FlatState.ema_from
    refresh_from_parameters = FlatState.refresh_from_parameters


def _feature_dim(cfg):
    return {"tiny": 192, "small": 384, "base": 768, "large": 1024}[cfg.model_type.split("_")[-1]]
