"""Spectrogram augmentations with the reference's class names and constructor arguments (augmentations.py), executed
by the fused HIP kernel sa_augment_views.

Two ways in:
  * per-sample modules (`MixupBYOLA`, `RandomResizeCrop`, `RandomLinearFader` on a [1, F, T] GPU tensor) -- drop-in
    for utils/transforms.py; they draw from the same global `np.random` / `random` streams in the same order as the
    reference, so a seeded run reproduces the reference's parameters exactly;
  * `BatchedPairAugment` -- the MI355X data path: B clips -> 2B views in ONE launch, the mixup memory bank kept as a
    device-resident ring of un-mixed log-mels (no per-sample Python, no host<->device traffic besides 40 B of
    parameters per view).  Its semantics are the reference's *sequential* ones: view e sees exactly the bank the
    reference's single worker would hold after e earlier views.

MixGaussianNoise (augmentations.py:125-141) and RunningNorm (:144-214) are provided too (off by default in the reference).
"""
import random

import numpy as np
import torch
import torch.nn as nn

from . import ops

EPS32 = 1.1920929e-07


def _to_dev(a, dtype, dev):
    """Host parameters -> device: lists / numpy arrays are staged through a fresh pinned block (the copy is asynchronous and the host runs
    ahead of the stream, so a reused block could be overwritten before its copy has read it)."""
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=dtype, non_blocking=True)
    h = torch.as_tensor(np.asarray(a, dtype={torch.float32: np.float32, torch.int32: np.int32}[dtype]))
    if torch.device(dev).type == "cuda":
        h = h.pin_memory()
    return h.to(dev, non_blocking=True)


def _launch(store, clip_stride, src, mix, params, out, F_in, T_in, canvas, do_fade, noise=None, max_w=None):
    dev = store.device
    p = _to_dev(params, torch.float32, dev).reshape(-1, 8)
    s = _to_dev(src, torch.int32, dev)
    m = _to_dev(mix, torch.int32, dev) if mix is not None else None
    # the kernel sizes its LDS source tile for the widest crop of the launch (include/ssl_audio_hip.h: max_w_ratio); the widest crop that was
    # DRAWN, not the canvas: 16-wide local crops of a 10 s clip are at most 0.6 of the canvas
    if max_w is None:
        max_w = float(np.asarray(params, dtype=np.float64).reshape(-1, 8)[:, 4].max()) if not isinstance(params, torch.Tensor) else canvas[1]
    ratio = min(max_w, canvas[1]) / max(out.shape[-1] - 1, 1)
    ops.augment_views(store, clip_stride, s, m, p, out, F_in, T_in, canvas, ratio, do_fade, noise)
    return out


def draw_rrc_params(canvas, in_size, time_scale, freq_scale, np_rng=np.random, py_rng=random):
    """RandomResizeCrop.get_params (augmentations.py:30-38): freq draw then time draw (numpy), then i, j (python
    `random`, inclusive bounds, drawn only when the canvas is larger than the crop)."""
    canvas_h, canvas_w = canvas
    src_h, src_w = in_size
    # np.random.uniform(a, b) IS a + (b - a) * random_sample() (one draw, the same two fp64 operations: checked draw for draw in
    # tests/test_boundary.py) at a tenth of the call cost; np.clip(int(.), 1, canvas) on plain ints; randint(0, n) = randrange(n + 1)
    rs = np_rng.random_sample
    h = min(max(int((freq_scale[0] + (freq_scale[1] - freq_scale[0]) * rs()) * src_h), 1), canvas_h)
    w = min(max(int((time_scale[0] + (time_scale[1] - time_scale[0]) * rs()) * src_w), 1), canvas_w)
    i = py_rng.randrange(canvas_h - h + 1) if canvas_h > h else 0
    j = py_rng.randrange(canvas_w - w + 1) if canvas_w > w else 0
    return i, j, h, w


class RandomResizeCrop(nn.Module):
    def __init__(self, out_size=(64, 96), virtual_crop_scale=(1.0, 1.5), freq_scale=(0.6, 1.5), time_scale=(0.6, 1.5)):
        super().__init__()
        self.out_size = out_size
        self.virtual_crop_scale = virtual_crop_scale
        self.freq_scale = freq_scale
        self.time_scale = time_scale
        self.interpolation = 'bicubic'

    @staticmethod
    def get_params(virtual_crop_size, in_size, time_scale, freq_scale):
        return draw_rrc_params(virtual_crop_size, in_size, time_scale, freq_scale)

    def forward(self, lms):
        F_in, T_in = lms.shape[-2:]
        canvas = [int(s * c) for s, c in zip((F_in, T_in), self.virtual_crop_scale)]
        i, j, h, w = self.get_params(canvas, (F_in, T_in), self.time_scale, self.freq_scale)
        out = torch.empty(1, self.out_size[0], self.out_size[1], device=lms.device)
        return _launch(lms.contiguous().float(), F_in * T_in, [0], None, [0.0, i, j, h, w, 0.0, 0.0, 0.0], out, F_in, T_in, canvas, False)

    def __repr__(self):
        format_string = self.__class__.__name__ + f'(virtual_crop_size={self.virtual_crop_scale}'
        format_string += ', time_scale={0}'.format(tuple(round(s, 4) for s in self.time_scale))
        format_string += ', freq_scale={0})'.format(tuple(round(r, 4) for r in self.freq_scale))
        return format_string


class RandomLinearFader(nn.Module):
    def __init__(self, gain=1.0):
        super().__init__()
        self.gain = gain

    def forward(self, lms):
        head, tail = self.gain * ((2.0 * np.random.rand(2)) - 1.0)
        F_in, T_in = lms.shape[-2:]
        out = torch.empty(1, F_in, T_in, device=lms.device)
        return _launch(lms.contiguous().float(), F_in * T_in, [0], None, [0.0, 0, 0, F_in, T_in, head, tail, 0.0], out, F_in, T_in,
                       (F_in, T_in), True)

    def __repr__(self):
        return self.__class__.__name__ + f'(gain={self.gain})'


def log_mixup_exp(xa, xb, alpha):
    """log(alpha * exp(xa) + (1 - alpha) * exp(xb) + eps) (augmentations.py:81-85) via the augmentation kernel."""
    F_in, T_in = xa.shape[-2:]
    store = torch.stack([xb.reshape(F_in, T_in), xa.reshape(F_in, T_in)]).contiguous().float()   # weight of slot 1 is "alpha"
    out = torch.empty(1, F_in, T_in, device=xa.device)
    _launch(store, F_in * T_in, [0], [1], [alpha, 0, 0, F_in, T_in, 0.0, 0.0, 0.0], out, F_in, T_in, (F_in, T_in), False)
    return out.view(xa.shape)


class MixupBYOLA(nn.Module):
    def __init__(self, ratio=0.2, n_memory=2048, log_mixup_exp=True):
        super().__init__()
        if not log_mixup_exp:
            raise NotImplementedError("only the log-mixup-exp variant (the reference default) is implemented")
        self.ratio = ratio
        self.n = n_memory
        self.log_mixup_exp = log_mixup_exp
        self.memory_bank = []

    def forward(self, x):
        alpha = self.ratio * np.random.random()
        if self.memory_bank:
            z = self.memory_bank[np.random.randint(len(self.memory_bank))]
            mixed = log_mixup_exp(z, x, alpha)         # weight alpha on z, 1 - alpha on x  (augmentations.py:110)
        else:
            mixed = x
        self.memory_bank = (self.memory_bank + [x])[-self.n:]
        return mixed.to(torch.float)

    def __repr__(self):
        return self.__class__.__name__ + f'(ratio={self.ratio},n={self.n},log_mixup_exp={self.log_mixup_exp})'


class MixGaussianNoise(nn.Module):
    """log((1 - l) exp(x) + exp(N(0, l)) + eps), l = ratio * np.random.rand() (augmentations.py:125-141).  The normal draws come
    from torch's generator of the input's device (`normal` overrides them, for parity tests with recorded draws)."""

    def __init__(self, ratio=0.2):
        super().__init__()
        self.ratio = ratio

    def forward(self, lms, normal=None):
        lambd = self.ratio * np.random.rand()
        x = lms.contiguous().float()
        if normal is None:
            normal = torch.randn(x.shape, device=x.device)
        out = torch.empty_like(x)
        ops.mix_gaussian_noise(x, normal.contiguous().float(), lambd, EPS32, out)
        return out

    def __repr__(self):
        return self.__class__.__name__ + f'(ratio={self.ratio})'


class RunningNorm(nn.Module):
    """Online normalisation with running mean / std over axis [1, 2] (augmentations.py:144-214): statistics are updated for the
    first epoch_samples * max_update_epochs calls, then frozen.  State lives on the device; one launch per call."""

    def __init__(self, epoch_samples, max_update_epochs=10, axis=[1, 2]):
        super().__init__()
        if list(axis) != [1, 2]:
            raise NotImplementedError("RunningNorm is implemented for the reference's axis=[1, 2]")
        self.max_update = epoch_samples * max_update_epochs
        self.axis = list(axis)
        self.n = 0
        self.state = None

    def forward(self, image):
        x = image.contiguous().float()
        if x.dim() != 3:
            raise ValueError("RunningNorm expects [C, F, T] inputs")
        if self.state is None:
            self.state = torch.zeros(x.shape[0], 2, device=x.device)
        out = torch.empty_like(x)
        update = self.n < self.max_update
        ops.running_norm(x, self.state, self.n, update, EPS32, out)
        if update:
            self.n += 1
        return out

    @property
    def mean(self):
        return self.state[:, 0].view(-1, 1, 1)

    @property
    def std(self):
        return self.state[:, 1].sqrt().clamp_min(EPS32).view(-1, 1, 1)

    def __repr__(self):
        return self.__class__.__name__ + f'(max_update={self.max_update},axis={self.axis})'


class NormalizeBatch(nn.Module):
    """(X - mean) / clamp(std_unbiased, eps) over the whole batch (augmentations.py:217-235, axis [0, 2, 3] with one channel)."""

    def __init__(self, axis=[0, 2, 3]):
        super().__init__()
        if list(axis) != [0, 2, 3]:
            raise NotImplementedError("NormalizeBatch is implemented for the reference's axis=[0, 2, 3]")
        self.axis = axis

    def forward(self, X):
        if X.shape[1] != 1:
            raise NotImplementedError("single-channel spectrogram batches only")
        X = X.contiguous().float()
        out = torch.empty_like(X)
        ws = torch.zeros(2, dtype=torch.float64, device=X.device)
        ops.normalize_batch(X, out, 0.0, ws, EPS32)
        return out


class BatchedPairAugment:
    """B clips -> views [2, B, 1, F, T_out] in one launch, with the reference's sequential mixup-bank semantics.

    The clip store is a device ring of normalised log-mels; `next_slots(B)` tells the caller (frontend) where to
    write the next batch so no copy is needed.  Event e = 2*clip + view appends clip e//2 to the virtual FIFO
    (augmentations.py:115, un-mixed input, capped at n_memory entries) after mixing with entry randint(len).

    `local_crops_number` = L > 0 adds the L local views of utils/transforms.py:38-47,54-55 (RandomResizeCrop of the UN-mixed clip to
    `local_crops_size`, virtual crop scale (1, 1), scales (0.05, 0.6); a second launch), drawn per clip after its two global views --
    the reference's RNG call order; `gnoise` inserts MixGaussianNoise (utils/transforms.py:21-22) between mixup and the crop of the
    global views, inside the same launch: lambda = gnoise_ratio * np.random.rand() per view from the numpy stream, the normal
    draws from a torch generator on the device (the reference draws them with torch.normal on the host: a different stream by
    construction; `normal_override` feeds recorded draws for parity tests)."""

    def __init__(self, device, n_mels=64, frames=1001, out_frames=None, mixup=True, rrc=True, rlf=True, mixup_ratio=0.2,
                 n_memory=2048, virtual_crop_scale=(1.0, 1.5), crop_scale=(0.6, 1.5), seed=None, local_crops_number=0,
                 local_crops_size=(16, 16), local_crop_scale=(0.05, 0.6), gnoise=False, gnoise_ratio=0.2):
        self.device = device
        self.F, self.T = n_mels, frames
        self.T_out = out_frames or frames
        self.mixup, self.rrc, self.rlf = mixup, rrc, rlf
        self.ratio, self.n = mixup_ratio, n_memory
        self.vcs, self.scale = tuple(virtual_crop_scale), tuple(crop_scale)
        self.L, self.local_size, self.local_scale = int(local_crops_number), tuple(local_crops_size), tuple(local_crop_scale)
        self.gnoise, self.gnoise_ratio = bool(gnoise), gnoise_ratio
        self.np_rng = np.random.RandomState(seed) if seed is not None else np.random
        self.py_rng = random.Random(seed) if seed is not None else random
        self.torch_gen = None
        if self.gnoise and seed is not None:
            self.torch_gen = torch.Generator(device=device)
            self.torch_gen.manual_seed(seed)
        self.normal_override = None   # [2B, F, T] standard-normal draws for the next call (parity tests), ordered like the views
        self.events = 0          # views processed so far == FIFO appends so far
        self.clips = 0           # clips processed so far
        self.store = None
        self.capacity = 0
        self._last = None        # the last batch's draws in event order (`records` builds its dictionaries from them on demand)
        self.starts = None       # the last batch's dataset crop starts (frames), one per clip; None when draw() was not given src_frames

    def _ensure_store(self, B):
        need = -(-(self.n // 2 + 1 + B) // B) * B           # multiple of B so a batch is always contiguous
        if self.store is None or self.capacity != need:
            if self.clips:
                raise RuntimeError("batch size changed mid-stream: the mixup ring would lose its history")
            self.capacity = need
            self.store = torch.zeros(need, self.F, self.T, device=self.device)

    def next_slots(self, B):
        """Ring region [B, F, T] (a view) the next batch of normalised log-mels must be written into."""
        self._ensure_store(B)
        if self.clips % B:
            raise RuntimeError("batches must keep a constant size")
        s0 = self.clips % self.capacity
        return self.store[s0:s0 + B]

    @property
    def records(self):
        """Explicit parameters of the last batch, one dictionary per view in DRAW order (parity tests and the smoke check read them; the
        step itself never does, so they are built on demand and cost the sampler nothing)."""
        if self._last is None:
            return []
        clip0, ev, loc = self._last
        L, out = self.L, []
        for e, (alpha, k, i, j, h, w, head, tail, lam) in enumerate(ev):
            out.append({"clip": clip0 + e // 2, "view": e % 2, "alpha": alpha, "bank_index": k, "rrc": (i, j, h, w), "head_tail": (head, tail),
                        "lambd": lam})
            if L and e % 2 == 1:
                for l in range(L):
                    out.append({"clip": clip0 + e // 2, "local": l, "rrc": loc[(e // 2) * L + l]})
        return out

    def draw(self, B, src_frames=None):
        """Host-side sampling in the reference's RNG call order; returns src, mix, params (arrays ordered
        [view0 of all clips..., view1 of all clips...]) and the canvas, while DRAWING in event order clip0v0, clip0v1, [clip0 locals,]
        clip1v0, ...  (`self.local_draw` = (src, params) of the L * B local views, ordered [local0 of all clips..., local1 ...]).

        src_frames (int, or one int per clip): frame count l of each clip's un-cropped log-mel.  The dataset's crop
        `start = np.random.randint(l - crop_frames)` (datasets.py:342-345; drawn only when l > crop_frames) is then sampled per clip
        BEFORE that clip's view draws -- `Dataset.__getitem__` crops, then runs the transform (:356-358) -- and left in `self.starts`
        for the frontend launch that fills `next_slots(B)`."""
        F, T = self.F, self.T
        if not self.capacity:
            self._ensure_store(B)
        canvas = [int(s * c) for s, c in zip((F, T), self.vcs)] if self.rrc else [F, T]
        rs, ri, pr = self.np_rng.random_sample, self.np_rng.randint, self.py_rng
        cap, n_mem, ratio, gratio = self.capacity, self.n, self.ratio, self.gnoise_ratio
        mixup, gnoise, rrc, rlf, L = self.mixup, self.gnoise, self.rrc, self.rlf, self.L
        scale, lscale, in_size = self.scale, self.local_scale, (F, T)
        per_clip = src_frames is not None and not isinstance(src_frames, (int, np.integer))
        if per_clip and len(src_frames) != B:
            raise ValueError(f"src_frames: expected one frame count per clip ({B}), got {len(src_frames)}")
        starts = [0] * B if src_frames is not None else None
        ev, loc, mixv = [], [], []
        events = self.events
        for b in range(B):
            if starts is not None:
                l = int(src_frames[b]) if per_clip else int(src_frames)
                if l > T:
                    starts[b] = int(ri(l - T))                         # datasets.py:344
            for v in range(2):
                alpha, k, lam, m = 0.0, -1, 0.0, -1
                if mixup:
                    alpha = ratio * rs()                               # augmentations.py:105
                    c = events if events < n_mem else n_mem
                    if c > 0:
                        k = int(ri(c))                                 # :108
                        m = ((events - c + k) // 2) % cap              # global append index of FIFO entry k -> its clip's ring slot
                    events += 1
                if gnoise:
                    lam = gratio * rs()                                # augmentations.py:135 (np.random.rand() is one random_sample)
                if rrc:
                    i, j, h, w = draw_rrc_params(canvas, in_size, scale, scale, self.np_rng, pr)
                else:
                    i, j, h, w = 0, 0, F, T
                if rlf:
                    head = 2.0 * rs() - 1.0                            # augmentations.py:70: 1.0 * (2 * np.random.rand(2) - 1)
                    tail = 2.0 * rs() - 1.0
                else:
                    head = tail = 0.0
                ev.append((alpha, k, i, j, h, w, head, tail, lam))
                mixv.append(m)
            for l_ in range(L):                                        # utils/transforms.py:54-55: the un-mixed clip, no fade
                loc.append(draw_rrc_params((F, T), in_size, lscale, lscale, self.np_rng, pr))
        self.events = events
        self.starts = starts
        self._last = (self.clips, ev, loc)
        # event order [clip][view] -> launch order [view][clip]
        slot = (self.clips + np.arange(B)) % cap
        a = np.asarray(ev, dtype=np.float64).reshape(B, 2, 9)
        par = np.empty((2, B, 8), dtype=np.float32)
        par[:, :, 0] = np.where(a[:, :, 1] >= 0, a[:, :, 0], 0.0).T
        par[:, :, 1:7] = a[:, :, 2:8].transpose(1, 0, 2)
        par[:, :, 7] = a[:, :, 8].T
        src = np.broadcast_to(slot.astype(np.int32), (2, B)).reshape(-1).copy()
        mix = np.asarray(mixv, dtype=np.int32).reshape(B, 2).T.reshape(-1).copy()
        self._max_w = float(a[:, :, 5].max())
        if L:
            la = np.asarray(loc, dtype=np.float32).reshape(B, L, 4)
            lpar = np.zeros((L, B, 8), dtype=np.float32)
            lpar[:, :, 1:5] = la.transpose(1, 0, 2)
            self.local_draw = (np.broadcast_to(slot.astype(np.int32), (L, B)).reshape(-1).copy(), lpar.reshape(-1, 8))
            self._max_w_local = float(la[:, :, 3].max())
        else:
            self.local_draw = (np.zeros(0, dtype=np.int32), np.zeros((0, 8), dtype=np.float32))
        return src, mix, par.reshape(-1, 8), canvas

    def __call__(self, B, out=None, drawn=None):
        """Augment the batch previously written into `next_slots(B)`; returns views [2, B, 1, F, T_out] -- or, with local crops, the
        crop list the reference's transform yields per clip, batched: [view1, view2, local_1 .. local_L], each [B, 1, h, w].
        `drawn`: what `draw(B, ...)` returned, when the caller sampled ahead of the frontend launch (it needs the crop starts)."""
        self._ensure_store(B)
        src, mix, par, canvas = drawn if drawn is not None else self.draw(B)
        if out is None:
            out = torch.empty(2, B, 1, self.F, self.T_out, device=self.device)
        noise = None
        if self.gnoise:
            noise = self.normal_override
            self.normal_override = None
            if noise is None:
                noise = torch.randn(2 * B, self.F, self.T, device=self.device, generator=self.torch_gen)
        _launch(self.store, self.F * self.T, src, mix if self.mixup else None, par, out.view(2 * B, 1, self.F, self.T_out), self.F, self.T,
                canvas, self.rlf, noise, max_w=self._max_w)
        crops = out
        if self.L:
            lh, lw = self.local_size
            loc = torch.empty(self.L, B, 1, lh, lw, device=self.device)
            _launch(self.store, self.F * self.T, self.local_draw[0], None, self.local_draw[1], loc.view(self.L * B, 1, lh, lw), self.F, self.T,
                    (self.F, self.T), False, max_w=self._max_w_local)
            crops = [out[0], out[1]] + [loc[l] for l in range(self.L)]
        self.clips += B
        return crops
