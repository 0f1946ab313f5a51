"""Spectrogram augmentation oracle (TEST INFRASTRUCTURE).  numpy fp64 arithmetic, explicit parameters.

Follows augmentations.py (RandomResizeCrop :12-61, RandomLinearFader :64-78, log_mixup_exp :81-85,
MixupBYOLA :88-122, NormalizeBatch :217-235) and utils/transforms.py:7-58 (AudioPairTransform).
Sampling is separated from arithmetic: `draw_*` functions reproduce the reference's RNG call order on
explicit `np.random.RandomState` / `random.Random` objects; `*_apply` functions are pure.
"""
import random as _pyrandom

import numpy as np

EPS32 = float(np.finfo(np.float32).eps)
CUBIC_A = -0.75  # PyTorch upsample_bicubic2d coefficient


# ------------------------------------------------------------------ mixup (augmentations.py:81-85,103-117)
def log_mixup_exp(xa, xb, alpha):
    """log(alpha*exp(xa) + (1-alpha)*exp(xb) + eps)   (augmentations.py:81-85)."""
    return np.log(alpha * np.exp(xa) + (1.0 - alpha) * np.exp(xb) + EPS32)


def mixup_apply(x, z, alpha):
    """MixupBYOLA.forward arithmetic: the call passes 1-alpha as the weight of x (augmentations.py:110)."""
    return log_mixup_exp(x, z, 1.0 - alpha)


# ------------------------------------------------------------------ RandomResizeCrop
def canvas_size(in_size, virtual_crop_scale):
    return [int(s * c) for s, c in zip(in_size, virtual_crop_scale)]  # augmentations.py:42


def draw_rrc_params(np_rng, py_rng, canvas, in_size, time_scale, freq_scale):
    """get_params (augmentations.py:30-38): freq draw, time draw (np), then i, j (python random, inclusive)."""
    canvas_h, canvas_w = canvas
    src_h, src_w = in_size
    h = int(np.clip(int(np_rng.uniform(*freq_scale) * src_h), 1, canvas_h))
    w = int(np.clip(int(np_rng.uniform(*time_scale) * src_w), 1, canvas_w))
    i = py_rng.randint(0, canvas_h - h) if canvas_h > h else 0
    j = py_rng.randint(0, canvas_w - w) if canvas_w > w else 0
    return i, j, h, w


def _cubic_weights(t, dtype):
    A = dtype(CUBIC_A)
    one, two = dtype(1), dtype(2)

    def c1(x):  # |x| <= 1
        return ((A + two) * x - (A + dtype(3))) * x * x + one

    def c2(x):  # 1 < |x| < 2
        return ((A * x - dtype(5) * A) * x + dtype(8) * A) * x - dtype(4) * A

    return [c2(t + one), c1(t), c1(one - t), c2(two - t)]


def _axis_taps(n_in, n_out, dtype):
    """align_corners=True source coords; taps floor-1..floor+2 clamped to [0, n_in-1].
    The coordinate (scale * dst, floor, fraction) is evaluated in fp32 exactly as PyTorch's CPU kernel
    does for float input -- at T=1001 an fp64 coordinate differs from the reference by up to 3e-4."""
    f32 = np.float32
    scale = f32(n_in - 1) / f32(n_out - 1) if n_out > 1 else f32(0)
    src = (scale * np.arange(n_out, dtype=f32)).astype(f32)
    fl = np.floor(src)
    t = (src - fl).astype(dtype)
    fl = fl.astype(np.int64)
    idx = np.stack([np.clip(fl + k, 0, n_in - 1) for k in (-1, 0, 1, 2)], 0)  # [4, n_out]
    w = np.stack(_cubic_weights(t, dtype), 0)  # [4, n_out]
    return idx, w


def bicubic_resize(crop, out_size, dtype=np.float64):
    """F.interpolate(mode='bicubic', align_corners=True) on [..., h, w] (augmentations.py:53-54)."""
    crop = np.asarray(crop, dtype=dtype)
    iy, wy = _axis_taps(crop.shape[-2], out_size[0], dtype)
    ix, wx = _axis_taps(crop.shape[-1], out_size[1], dtype)
    rows = sum(crop[..., iy[k], :] * wy[k][:, None] for k in range(4))  # [..., H_out, w]
    return sum(rows[..., :, ix[k]] * wx[k] for k in range(4))


def rrc_apply(x, params, out_size, virtual_crop_scale=(1.0, 1.5), dtype=np.float64):
    """RandomResizeCrop.forward with explicit (i, j, h, w)  (augmentations.py:40-55). x: [C, F, T]."""
    x = np.asarray(x, dtype=dtype)
    c, F_, T_ = x.shape
    lh, lw = canvas_size((F_, T_), virtual_crop_scale)
    canvas = np.zeros((c, lh, lw), dtype=dtype)
    x0, y0 = (lw - T_) // 2, (lh - F_) // 2
    canvas[:, y0:y0 + F_, x0:x0 + T_] = x
    i, j, h, w = params
    return bicubic_resize(canvas[:, i:i + h, j:j + w], out_size, dtype)


# ------------------------------------------------------------------ RandomLinearFader (augmentations.py:69-74)
def linear_fader_apply(x, head, tail):
    T_ = x.shape[-1]
    return x + np.linspace(head, tail, T_).reshape((1,) * (x.ndim - 1) + (T_,))


# ------------------------------------------------------------------ NormalizeBatch (augmentations.py:229-232)
def normalize_batch(X):
    X = np.asarray(X, dtype=np.float64)
    mean = X.mean(axis=(0, 2, 3), keepdims=True)
    std = np.maximum(X.std(axis=(0, 2, 3), keepdims=True, ddof=1), EPS32)
    return (X - mean) / std


# ------------------------------------------------------------------ AudioPairTransform (utils/transforms.py:7-58)
class PairTransformOracle:
    """Sequential, stateful restatement: one shared MixupBYOLA bank + shared RNG streams, as in the
    reference where the same module instance is applied twice per clip (utils/transforms.py:52-53).

    `records` keeps the explicit parameters of every view so the batched GPU path can be driven with
    exactly the same (alpha, bank index, i, j, h, w, head, tail).
    """

    def __init__(self, n_mels=64, crop_frames=96, local_crops_number=0, local_crops_size=(16, 16),
                 mixup=True, rrc=True, rlf=True, mixup_ratio=0.2, virtual_crop_scale=(1.0, 1.5),
                 global_crop_scale=(0.6, 1.5), local_crop_scale=(0.05, 0.6), n_memory=2048, seed=None,
                 np_rng=None, py_rng=None, gnoise=False, gnoise_ratio=0.2, normal_fn=None):
        self.out_size = (n_mels, crop_frames)
        self.L = local_crops_number
        self.local_size = tuple(local_crops_size)
        self.mixup, self.rrc, self.rlf = mixup, rrc, rlf
        self.ratio = mixup_ratio
        self.vcs = tuple(virtual_crop_scale)
        self.gscale, self.lscale = tuple(global_crop_scale), tuple(local_crop_scale)
        self.n = n_memory
        self.bank = []
        # --Gnoise (utils/transforms.py:21-22): MixGaussianNoise(ratio) between mixup and the crop of the global views.  The reference
        # draws its normals with torch.normal; here they come from `normal_fn(shape)` so a test can hand both sides the same draws
        self.gnoise, self.gnoise_ratio, self.normal_fn = gnoise, gnoise_ratio, normal_fn
        self.np_rng = np_rng if np_rng is not None else np.random.RandomState(seed)
        self.py_rng = py_rng if py_rng is not None else _pyrandom.Random(seed)
        self.records = []

    def _global(self, x):
        rec = {}
        y = x
        if self.mixup:
            alpha = self.ratio * self.np_rng.random_sample()            # augmentations.py:105
            rec["alpha"] = alpha
            if self.bank:
                k = int(self.np_rng.randint(len(self.bank)))            # :108
                rec["bank_index"] = k
                y = mixup_apply(x, self.bank[k], alpha)                 # :110
            else:
                rec["bank_index"] = -1
            self.bank = (self.bank + [x])[-self.n:]                     # :115 (stores the UN-mixed input)
        if self.gnoise:
            lambd = self.gnoise_ratio * self.np_rng.rand()              # :135
            rec["lambd"] = lambd
            y = mix_gaussian_noise(np.asarray(y, dtype=np.float32), lambd, self.normal_fn(x.shape)).astype(np.float64)
        if self.rrc:
            canvas = canvas_size(x.shape[-2:], self.vcs)
            p = draw_rrc_params(self.np_rng, self.py_rng, canvas, x.shape[-2:], self.gscale, self.gscale)
            rec["rrc"] = p
            y = rrc_apply(y, p, self.out_size, self.vcs)
        if self.rlf:
            head, tail = 1.0 * ((2.0 * self.np_rng.rand(2)) - 1.0)      # :70
            rec["head_tail"] = (head, tail)
            y = linear_fader_apply(y, head, tail)
        self.records.append(rec)
        return y

    def _local(self, x):
        canvas = canvas_size(x.shape[-2:], (1, 1))
        p = draw_rrc_params(self.np_rng, self.py_rng, canvas, x.shape[-2:], self.lscale, self.lscale)
        self.records.append({"rrc": p, "local": True})
        return rrc_apply(x, p, self.local_size, (1, 1))

    def __call__(self, x):
        """x: [1, F, T] -> [view1, view2, local_1..local_L]  (utils/transforms.py:49-56)."""
        x = np.asarray(x, dtype=np.float64)
        crops = [self._global(x), self._global(x)]
        for _ in range(self.L):
            crops.append(self._local(x))
        return crops

    def dataset_item(self, lms, norm_stats=None):
        """`Dataset.__getitem__` from the loaded log-mel on (datasets.py:342-358): lms [1, F, l] -> trim at
        `start = np.random.randint(l - crop_frames)` when l > crop_frames (the GLOBAL numpy stream: the one the transform's draws come
        from, so the crop draw precedes the clip's view draws) or right zero-pad when shorter (:346-350), normalise (:353-354), then the
        pair transform (:356-358).  Returns (crops, start); `self.starts` keeps every start drawn."""
        lms = np.asarray(lms, dtype=np.float64)
        l, crop = lms.shape[-1], self.out_size[1]
        start = 0
        if l > crop:
            start = int(self.np_rng.randint(l - crop))                  # datasets.py:344
            lms = lms[..., start:start + crop]
        elif l < crop:
            lms = np.pad(lms, [(0, 0)] * (lms.ndim - 1) + [(0, crop - l)])
        if norm_stats is not None:
            lms = (lms - norm_stats[0]) / norm_stats[1]
        if not hasattr(self, "starts"):
            self.starts = []
        self.starts.append(start)
        return self(lms), start


def mix_gaussian_noise(lms, lambd, normal):
    """MixGaussianNoise.forward (augmentations.py:132-141) with the draws made explicit: lambd = ratio * np.random.rand(),
    z = exp(N(0, lambd)) = exp(lambd * normal); returns log((1 - lambd) * exp(lms) + z + eps)."""
    x = np.exp(lms.astype(np.float32))
    z = np.exp((np.float32(lambd) * normal.astype(np.float32)).astype(np.float32))
    return np.log((np.float32(1.0 - lambd) * x + z + np.float32(EPS32)).astype(np.float32)).astype(np.float32)


class RunningNormOracle:
    """RunningNorm (augmentations.py:144-214, axis [1, 2]): incremental mean mu += (mean(x) - mu) / n with n = samples seen BEFORE
    this one (the first sample initialises), the same recurrence on mean((x - mu)^2), std clamped below by eps; frozen after
    max_update samples."""

    def __init__(self, epoch_samples, max_update_epochs=10):
        self.max_update = epoch_samples * max_update_epochs
        self.n = 0
        self.mu = None
        self.s2 = None

    def __call__(self, image):
        x = image.astype(np.float32)
        if self.n < self.max_update:
            m = x.mean(axis=(1, 2), keepdims=True, dtype=np.float32)
            self.mu = m if self.n == 0 else self.mu + (m - self.mu) / np.float32(self.n)
            v = ((x - self.mu) ** 2).mean(axis=(1, 2), keepdims=True, dtype=np.float32)
            self.s2 = v if self.n == 0 else self.s2 + (v - self.s2) / np.float32(self.n)
            self.n += 1
        std = np.maximum(np.sqrt(self.s2), np.float32(EPS32))
        return ((x - self.mu) / std).astype(np.float32)
