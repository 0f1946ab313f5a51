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
                    engine.GRAD_SINK[id(p)] = (weakref.ref(p), self.grads[off:off + n].view(p.shape))
                if name in self.qkv_fused:
                    engine.QKV_BIAS[id(p)] = (weakref.ref(p), self.params[off:off + 3 * n])
            off += pad8(n)
        ops.cast_bf16(self.params, self.params_bf16)
        self.bind_bf16()
        self.step_count = 0

    def bind_bf16(self):
        """Point the engine's bf16 weight cache at views of the flat bf16 buffer (kept fresh by the AdamW kernel)."""
        for name, p in self.order:
            off, n = self.offsets[name]
            w = self.params_bf16[off:off + n]
            w = w.view(p.shape[0], -1) if p.dim() > 1 else w
            engine.BF16_WEIGHTS.pin(p, w)

    def zero_grad(self):
        self.grads.zero_()

    def adamw(self, lr, wd, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        self.step_count += 1
        nd, nt = self.n_decay, self.n_train
        if nd:
            ops.adamw_step(self.params[:nd], self.grads[:nd], self.m[:nd], self.v[:nd], lr, betas[0], betas[1], eps, wd, self.step_count,
                           grad_scale, self.params_bf16[:nd])
        if nt > nd:
            ops.adamw_step(self.params[nd:nt], self.grads[nd:nt], self.m[nd:nt], self.v[nd:nt], lr, betas[0], betas[1], eps, 0.0,
                           self.step_count, grad_scale, self.params_bf16[nd:nt])
        engine.WEIGHT_EPOCH[0] += 1                    # copies derived from the bf16 weights (transposed dgrad operands) are stale now

    # ------------------------------------------------------------------ checkpoint / resume (main_bt_byol.py:492-503, utils/utils.py:37-46)
    def _trainable(self):
        return [(n, p) for n, p in self.order if p.requires_grad]      # decayed first, then un-decayed: utils.get_param_groups' order

    def optim_state_dict(self, lr, wd, betas=(0.9, 0.999), eps=1e-8):
        """The Adam moments in `torch.optim.AdamW.state_dict()` form over `utils.get_param_groups(model)`'s parameter order, so that the
        'optimizer' entry of a checkpoint is interchangeable between this flat state and the reference driver's own optimiser."""
        state, idx = {}, 0
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

    def load_optim_state_dict(self, sd):
        """Inverse of optim_state_dict; accepts what torch.optim.AdamW(utils.get_param_groups(model)).state_dict() wrote."""
        named = self._trainable()
        n_saved = sum(len(g["params"]) for g in sd["param_groups"])
        if n_saved != len(named):
            raise ValueError(f"optimizer state has {n_saved} parameters, the model has {len(named)} trainable ones")
        steps = set()
        self.m.zero_()
        self.v.zero_()
        for i, (n, p) in enumerate(named):
            st = sd["state"].get(i)
            if st is None:                                   # a parameter that never received a gradient has no state entry
                continue
            off, cnt = self.offsets[n]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {i} has shape {tuple(st['exp_avg'].shape)}, parameter {n} has {tuple(p.shape)}")
            self.m[off:off + cnt].copy_(st["exp_avg"].reshape(-1).to(self.m.device, torch.float32))
            self.v[off:off + cnt].copy_(st["exp_avg_sq"].reshape(-1).to(self.v.device, torch.float32))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused AdamW keeps one count")
        self.step_count = steps.pop() if steps else 0

    def refresh_from_parameters(self):
        """After load_state_dict wrote the (flat-backed) parameters: re-cast the bf16 copies the GEMMs read."""
        ops.cast_bf16(self.params, self.params_bf16)
        engine.WEIGHT_EPOCH[0] += 1

    def ema_from(self, other, beta):
        """self = beta * self + (1 - beta) * other over all parameters (utils/utils.py:328-331), one launch."""
        assert self.params.numel() == other.params.numel()
        ops.ema_update(self.params, other.params, beta)
        ops.cast_bf16(self.params, self.params_bf16)
        engine.WEIGHT_EPOCH[0] += 1


class GradSync:
    """Gradient SUM all-reduce over ranks, bucketed per transformer block and overlapped with backward on a side
    HIP stream (collective site C3).  Buckets are contiguous ranges of FlatState.grads."""

    def __init__(self, flat, bucket_bytes=64 << 20):
        self.flat = flat
        self.world = sdist.get_world_size()
        self.active = sdist.collectives_active()
        self.stream = torch.cuda.Stream() if (self.active and flat.grads.is_cuda) else None
        self.pending = []
        self.bucket_bytes = bucket_bytes
        self._ready_ranges = []

    def block_done(self, params):
        """engine.BLOCK_DONE_HOOK: the gradients of `params` are final -> reduce their flat range now."""
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
        for a, b in spans:
            if run is not None and a <= run[1]:
                run = (run[0], max(run[1], b))
            else:
                if run is not None:
                    self._launch(run[0], min(run[1], self.flat.n_train))
                run = (a, b)
        if run is not None:
            self._launch(run[0], min(run[1], self.flat.n_train))

    def _launch(self, lo, hi):
        self._ready_ranges.append((lo, hi))
        view = self.flat.grads[lo:hi]
        if self.stream is None:
            sdist.all_reduce_sum_(view)
            return
        ev = torch.cuda.Event()
        ev.record()
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
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
                 ema_beta=0.99):
        self.cfg, self.device, self.mode = cfg, device, mode
        self.world = sdist.get_world_size()
        self.B = batch_per_rank or cfg.batch_size // self.world
        self.clip_samples = clip_samples
        self.from_waveform = from_waveform
        torch.manual_seed(seed)
        self.online = MultiCropWrapper(ModelWrapper(cfg), BarlowTwinsHead(cfg, _feature_dim(cfg))).to(device)
        self.flat = FlatState(list(self.online.named_parameters()), device)
        self.criterion = BarlowTwinsLoss(cfg, ncrops=2).to(device)
        sdist.reserve_cus_for_collectives()
        self.sync = GradSync(self.flat)
        engine.BLOCK_DONE_HOOK = self.sync.block_done
        self.predictor = self.target = self.flat_pred = self.flat_target = None
        if mode == "byol":
            self.predictor = BarlowTwinsPredictor(cfg.projector_out_dim, use=True).to(device)
            self.flat_pred = FlatState(list(self.predictor.named_parameters()), device)
            self.sync_pred = GradSync(self.flat_pred)
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
                                          cfg.mixup_ratio, virtual_crop_scale=tuple(cfg.virtual_crop_scale), seed=seed + 1000 * sdist.get_rank())
        self.post_norm = NormalizeBatch() if cfg.post_norm else None
        self.lr, self.wd = cfg.lr, cfg.wd
        self.last_loss = None
        # main_bt_byol.py:116-118 stops on a non-finite loss with a host sync every step; here the test is a device-side counter that
        # the host reads every `finite_check_every` steps (and whenever assert_finite() is called)
        self._nonfinite = torch.zeros(1, dtype=torch.int32, device=device)
        self.finite_check_every = 100
        self._steps = 0

    # ------------------------------------------------------------------ data path
    def make_views(self, batch):
        """batch: waveforms [B, L] (from_waveform) or log-mels [B, 1, F, T_any]; -> [view1, view2] each [B,1,F,crop_frames]."""
        B = batch.shape[0]
        slots = self.augment.next_slots(B)
        if self.from_waveform:
            self.frontend(batch, crop_frames=self.cfg.crop_frames, start=0, norm_stats=AUDIOSET_STATS, out=slots.view(B, 1, *slots.shape[1:]))
        else:
            slots.copy_(batch.view(B, *batch.shape[-2:]))
        views = self.augment(B)
        v1, v2 = views[0], views[1]
        if self.post_norm is not None:
            v1, v2 = self.post_norm(v1), self.post_norm(v2)
        return [v1, v2]

    # ------------------------------------------------------------------ one optimisation step
    def step(self, batch):
        return self.step_views(self.make_views(batch))

    def step_views(self, views):
        """One optimisation step on two already-augmented views [B,1,F,T] (what train_one_epoch receives from its loader)."""
        self.flat.zero_grad()
        engine.reset_pending_backward()
        if self.mode == "bt":
            z = self.online(views, ncrops=2)
            z1, z2 = z.chunk(2)
            loss = self.criterion.forward_loss(z1, z2)
        elif self.mode == "mae":
            t, recon = self.online(views[:1], ncrops=1, mask_ratio=self.cfg.mask_ratio, masked_recon=True)
            st = self.online(views[1:], ncrops=1)
            loss = self.criterion(st, t, ngcrops_each=1) + recon
        else:
            self.flat_pred.zero_grad()
            o = self.online(views[:2], ncrops=2)
            o = self.predictor(o, ncrops=1)
            with torch.no_grad():
                t = self.target(views, ncrops=2)
            loss = self.criterion(o, t, ngcrops_each=2)
            self.flat_target.ema_from(self.flat, self.ema_beta)      # before the optimiser step (main_bt_byol.py:121-126)
        ops.count_nonfinite(loss.detach().reshape(1), self._nonfinite)
        loss.backward()
        self.sync.finish()
        self.flat.adamw(self.lr, self.wd)
        if self.mode == "byol":
            self.sync_pred.finish()
            self.flat_pred.adamw(self.lr, self.wd)
        self.last_loss = loss.detach()
        self._steps += 1
        if self._steps % self.finite_check_every == 0:
            self.assert_finite()
        return self.last_loss

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self, epoch=0):
        """The dictionary main_bt_byol.py:492-498 saves ('model', 'optimizer', 'epoch', 'barlow_twins_loss'; the reference's drop of the
        predictor / target there is SURVEY.md A.5's quirk -- they are included here so that a byol run resumes exactly)."""
        sd = {"model": self.online.state_dict(), "optimizer": self.flat.optim_state_dict(self.lr, self.wd), "epoch": epoch,
              "barlow_twins_loss": self.criterion.state_dict(), "steps": self._steps}
        if self.mode == "byol":
            sd["predictor"] = self.predictor.state_dict()
            sd["optimizer_predictor"] = self.flat_pred.optim_state_dict(self.lr, self.wd)
            sd["target"] = self.target.state_dict()
        return sd

    def load_state_dict(self, ckpt):
        """Resume from state_dict()'s dictionary, or from a checkpoint of the reference driver (utils/utils.py:37-46: 'model', 'optimizer'
        [, 'predictor'], 'epoch'; DDP's 'module.' prefix accepted).  Returns the epoch to continue with."""
        strip = lambda sd: {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.online.load_state_dict(strip(ckpt["model"]))
        self.flat.refresh_from_parameters()
        if "optimizer" in ckpt:
            self.flat.load_optim_state_dict(ckpt["optimizer"])
        if "barlow_twins_loss" in ckpt:
            self.criterion.load_state_dict(ckpt["barlow_twins_loss"])
        if self.mode == "byol":
            if "predictor" in ckpt:
                self.predictor.load_state_dict(strip(ckpt["predictor"]))
                self.flat_pred.refresh_from_parameters()
            if "optimizer_predictor" in ckpt:
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
        ops.cast_bf16(self.params, self.params_bf16)

    ema_from = FlatState.ema_from
    refresh_from_parameters = FlatState.refresh_from_parameters


def _feature_dim(cfg):
    return {"tiny": 192, "small": 384, "base": 768, "large": 1024}[cfg.model_type.split("_")[-1]]
