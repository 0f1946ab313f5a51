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


def _launch(store, clip_stride, src, mix, params, out, F_in, T_in, canvas, do_fade, noise=None):
    dev = store.device
    p = torch.tensor(params, dtype=torch.float32).reshape(-1, 8).to(dev, non_blocking=True)
    s = torch.tensor(src, dtype=torch.int32).to(dev, non_blocking=True)
    m = torch.tensor(mix, dtype=torch.int32).to(dev, non_blocking=True) if mix is not None else None
    ratio = canvas[1] / max(out.shape[-1] - 1, 1)
    ops.augment_views(store, clip_stride, s, m, p, out, F_in, T_in, canvas, ratio, do_fade, noise)
    return out


def draw_rrc_params(canvas, in_size, time_scale, freq_scale, np_rng=np.random, py_rng=random):
    """RandomResizeCrop.get_params (augmentations.py:30-38): freq draw then time draw (numpy), then i, j (python
    `random`, inclusive bounds, drawn only when the canvas is larger than the crop)."""
    canvas_h, canvas_w = canvas
    src_h, src_w = in_size
    h = int(np.clip(int(np_rng.uniform(*freq_scale) * src_h), 1, canvas_h))
    w = int(np.clip(int(np_rng.uniform(*time_scale) * src_w), 1, canvas_w))
    i = py_rng.randint(0, canvas_h - h) if canvas_h > h else 0
    j = py_rng.randint(0, canvas_w - w) if canvas_w > w else 0
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
        self.records = []        # explicit parameters of the last batch (for parity tests)

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

    def draw(self, B):
        """Host-side sampling in the reference's RNG call order; returns src, mix, params lists ordered
        [view0 of all clips..., view1 of all clips...] while DRAWING in event order clip0v0, clip0v1, [clip0 locals,] clip1v0, ...
        (`self.local_draw` = (src, params) of the L * B local views, ordered [local0 of all clips..., local1 ...])."""
        canvas = [int(s * c) for s, c in zip((self.F, self.T), self.vcs)] if self.rrc else [self.F, self.T]
        src = [0] * (2 * B)
        mix = [-1] * (2 * B)
        par = [None] * (2 * B)
        lsrc, lpar = [0] * (self.L * B), [None] * (self.L * B)
        self.records = []
        for b in range(B):
            clip = self.clips + b
            for v in range(2):
                alpha, k, lam = 0.0, -1, 0.0
                if self.mixup:
                    alpha = self.ratio * self.np_rng.random_sample()
                    c = min(self.events, self.n)
                    if c > 0:
                        k = int(self.np_rng.randint(c))
                        g = self.events - c + k                       # global append index of FIFO entry k
                        mix[v * B + b] = (g // 2) % self.capacity
                    self.events += 1
                if self.gnoise:
                    lam = self.gnoise_ratio * self.np_rng.rand()      # augmentations.py:135
                if self.rrc:
                    i, j, h, w = draw_rrc_params(canvas, (self.F, self.T), self.scale, self.scale, self.np_rng, self.py_rng)
                else:
                    i, j, h, w = 0, 0, self.F, self.T
                head, tail = (1.0 * ((2.0 * self.np_rng.rand(2)) - 1.0)) if self.rlf else (0.0, 0.0)
                src[v * B + b] = clip % self.capacity
                par[v * B + b] = [alpha if k >= 0 else 0.0, i, j, h, w, head, tail, lam]
                self.records.append({"clip": clip, "view": v, "alpha": alpha, "bank_index": k, "rrc": (i, j, h, w),
                                     "head_tail": (head, tail), "lambd": lam})
            for l in range(self.L):                                   # utils/transforms.py:54-55: the un-mixed clip, no fade
                i, j, h, w = draw_rrc_params((self.F, self.T), (self.F, self.T), self.local_scale, self.local_scale, self.np_rng, self.py_rng)
                lsrc[l * B + b] = clip % self.capacity
                lpar[l * B + b] = [0.0, i, j, h, w, 0.0, 0.0, 0.0]
                self.records.append({"clip": clip, "local": l, "rrc": (i, j, h, w)})
        self.local_draw = (lsrc, lpar)
        return src, mix, par, canvas

    def __call__(self, B, out=None):
        """Augment the batch previously written into `next_slots(B)`; returns views [2, B, 1, F, T_out] -- or, with local crops, the
        crop list the reference's transform yields per clip, batched: [view1, view2, local_1 .. local_L], each [B, 1, h, w]."""
        self._ensure_store(B)
        src, mix, par, canvas = self.draw(B)
        if out is None:
            out = torch.empty(2, B, 1, self.F, self.T_out, device=self.device)
        noise = None
        if self.gnoise:
            noise = self.normal_override
            self.normal_override = None
            if noise is None:
                noise = torch.randn(2 * B, self.F, self.T, device=self.device, generator=self.torch_gen)
        _launch(self.store, self.F * self.T, src, mix if self.mixup else None, par, out.view(2 * B, 1, self.F, self.T_out), self.F, self.T,
                canvas, self.rlf, noise)
        crops = out
        if self.L:
            lh, lw = self.local_size
            loc = torch.empty(self.L, B, 1, lh, lw, device=self.device)
            _launch(self.store, self.F * self.T, self.local_draw[0], None, self.local_draw[1], loc.view(self.L * B, 1, lh, lw), self.F, self.T,
                    (self.F, self.T), False)
            crops = [out[0], out[1]] + [loc[l] for l in range(self.L)]
        self.clips += B
        return crops
