"""Log-mel frontend (host side): mirrors the reference's use of torchaudio at datasets.py:39-48,115 and the
dataset's crop / pad / normalise at datasets.py:342-354, executed by the HIP kernel sa_logmel_fwd.

`MelSpectrogram` keeps torchaudio's constructor argument names so the call site at datasets.py:39-48 reads
the same; only the subset of behaviour the reference uses is supported (power=2, periodic Hann, center,
reflect padding, HTK mel scale, no normalisation).
"""
import math

import numpy as np
import torch

from . import ops

EPS32 = 1.1920929e-07


def _mel_filterbank(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> [n_freqs, n_mels] (float64)."""
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    f_pts = 700.0 * (10.0 ** (np.linspace(m_min, m_max, n_mels + 2) / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    return np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))


def build_tables(device, n_fft=1024, n_mels=64, f_min=60.0, f_max=7800.0, sample_rate=16000, win_length=None):
    """Window, FFT twiddles and compact mel weights as device tensors (built once, in float64, stored fp32)."""
    if n_fft != 1024 or n_mels != 64:
        raise NotImplementedError("the HIP frontend is specialised for n_fft=1024, n_mels=64 (reference defaults)")
    win_length = win_length or n_fft
    k = np.arange(n_fft, dtype=np.float64)
    # torch.hann_window(win_length, periodic=True), centred in the n_fft frame when shorter (torch.stft pads it on both sides:
    # hear/config.yaml has win_length 400 against the training recipe's 1024)
    window = np.zeros(n_fft)
    left = (n_fft - win_length) // 2
    window[left:left + win_length] = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length, dtype=np.float64) / win_length)
    tw = np.stack([np.cos(2.0 * np.pi * k / n_fft), -np.sin(2.0 * np.pi * k / n_fft)], axis=1)
    fb = _mel_filterbank(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate)
    lo = np.zeros(n_mels, dtype=np.int32)
    ln = np.zeros(n_mels, dtype=np.int32)
    for m in range(n_mels):
        nz = np.nonzero(fb[:, m])[0]
        if len(nz):
            lo[m], ln[m] = nz[0], nz[-1] - nz[0] + 1
    maxlen = int(ln.max())
    w = np.zeros((maxlen, n_mels), dtype=np.float64)
    for m in range(n_mels):
        w[:ln[m], m] = fb[lo[m]:lo[m] + ln[m], m]
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(device)
    return {"window": t(window, torch.float32), "twiddle": t(tw, torch.float32), "mel_weights": t(w, torch.float32),
            "mel_lo": t(lo, torch.int32), "mel_len": t(ln, torch.int32)}


class MelSpectrogram(torch.nn.Module):
    """Drop-in for `AT.MelSpectrogram(sample_rate, n_fft, win_length, hop_length, n_mels, f_min, f_max, power=2)`
    followed by `(x + eps).log()` and optional crop/pad/normalise, fused into one HIP launch.

    forward(wave [B, L] fp32 on the GPU) -> log-mel [B, 1, n_mels, T].
    """

    def __init__(self, sample_rate=16000, n_fft=1024, win_length=1024, hop_length=160, n_mels=64, f_min=60, f_max=7800, power=2):
        super().__init__()
        if win_length > n_fft or power != 2:
            raise NotImplementedError("only win_length <= n_fft and power == 2 (the reference's settings) are supported")
        self.cfg = dict(n_fft=n_fft, n_mels=n_mels, f_min=float(f_min), f_max=float(f_max), sample_rate=sample_rate, win_length=win_length)
        self.hop = hop_length
        self._tables = None

    def n_frames(self, n_samples):
        return 1 + n_samples // self.hop

    @staticmethod
    def _per_clip(v, B, device, name):
        """int / None -> (scalar, None); a sequence or tensor of B ints -> (0, int32 device vector) (staged through pinned memory)."""
        if v is None or isinstance(v, (int, np.integer)):
            return int(v or 0), None
        if isinstance(v, torch.Tensor):
            t = v.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()
        else:
            h = torch.as_tensor(np.asarray(v, dtype=np.int32))
            t = (h.pin_memory() if device.type == "cuda" else h).to(device, non_blocking=True)
        if t.numel() != B:
            raise ValueError(f"{name}: expected one value per clip ({B}), got {t.numel()}")
        return 0, t.view(B)

    def forward(self, wave, crop_frames=None, start=0, norm_stats=None, out=None, lengths=None, offsets=None):
        """norm_stats=(mean, std) applies (x-mean)/std; crop_frames/start follow datasets.py:342-351.

        `start`, `lengths`, `offsets` may each be one value per clip (sequence or tensor): clip b's own crop start in frames
        (`np.random.randint(l - crop_frames)`, datasets.py:344), its own length in samples (its frames end and reflect there; the rest of
        the crop is the right zero pad of datasets.py:346-350) and its first sample inside row b (the waveform crop of datasets.py:108-112)
        -- what `Dataset.__getitem__` does per sample, in one launch."""
        if wave.dim() == 1:
            wave = wave[None]
        if self._tables is None or self._tables["window"].device != wave.device:
            self._tables = build_tables(wave.device, **self.cfg)
        B, L = wave.shape
        start, starts = self._per_clip(start, B, wave.device, "start")
        _, lens = self._per_clip(lengths, B, wave.device, "lengths") if lengths is not None else (0, None)
        _, offs = self._per_clip(offsets, B, wave.device, "offsets") if offsets is not None else (0, None)
        T = crop_frames if crop_frames is not None else self.n_frames(L)
        mean, std = norm_stats if norm_stats is not None else (0.0, 1.0)
        if out is None:
            out = torch.empty(B, 1, self.cfg["n_mels"], T, dtype=torch.float32, device=wave.device)
        ops.logmel_fwd(wave.contiguous(), self._tables, out.view(B, -1), T, start, mean, std, self.hop, starts, lens, offs)
        return out
