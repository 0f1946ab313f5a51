"""Log-mel frontend oracle (TEST INFRASTRUCTURE; parity UNPINNED -- see oracle/__init__.py).

Restates torchaudio.transforms.MelSpectrogram with the arguments the reference passes at
datasets.py:39-48 (sample_rate 16000, n_fft = win_length = 1024, hop 160, n_mels 64, f_min 60,
f_max 7800, power 2; everything else torchaudio default: periodic Hann, center=True,
pad_mode='reflect', onesided, normalized=False, mel_scale='htk', norm=None) followed by
`(mel + finfo(float32).eps).log()` (datasets.py:115) and the dataset's crop/pad + normalise
(datasets.py:342-354).
"""
import numpy as np

EPS32 = float(np.finfo(np.float32).eps)  # 1.1920929e-07, torch.finfo().eps


def hann_periodic(n, dtype=np.float64):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)).astype(dtype)


def mel_filterbank(n_freqs=513, f_min=60.0, f_max=7800.0, n_mels=64, sample_rate=16000):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> [n_freqs, n_mels] fp64."""
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def n_frames(n_samples, hop=160):
    return 1 + n_samples // hop


def stft_window(n_fft, win_length=None, dtype=np.float64):
    """torch.stft's window: periodic Hann of win_length, zero-padded on both sides to n_fft when shorter (hear/config.yaml: 400)."""
    win_length = win_length or n_fft
    w = np.zeros(n_fft, dtype=dtype)
    left = (n_fft - win_length) // 2
    w[left:left + win_length] = hann_periodic(win_length, dtype)
    return w


def power_spectrogram(wave, n_fft=1024, hop=160, dtype=np.float64, win_length=None):
    """wave [..., L] -> |STFT|^2 [..., n_fft//2+1, frames]; center=True reflect padding."""
    wave = np.asarray(wave, dtype=dtype)
    pad = n_fft // 2
    x = np.pad(wave, [(0, 0)] * (wave.ndim - 1) + [(pad, pad)], mode="reflect")
    T = n_frames(wave.shape[-1], hop)
    idx = np.arange(n_fft)[None, :] + hop * np.arange(T)[:, None]
    frames = x[..., idx] * stft_window(n_fft, win_length, dtype)
    spec = np.fft.rfft(frames, axis=-1)
    p = spec.real ** 2 + spec.imag ** 2
    return np.swapaxes(p, -1, -2)


def mel_power(wave, n_fft=1024, hop=160, n_mels=64, f_min=60.0, f_max=7800.0, sample_rate=16000, dtype=np.float64, win_length=None):
    """wave [..., L] -> mel power spectrogram [..., n_mels, frames]: what `MelSpectrogram(..., power=2)` returns (datasets.py:39-48)."""
    p = power_spectrogram(wave, n_fft, hop, dtype, win_length)
    fb = mel_filterbank(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate).astype(dtype)
    return np.einsum("...ft,fm->...mt", p, fb)


def logmel(wave, n_fft=1024, hop=160, n_mels=64, f_min=60.0, f_max=7800.0, sample_rate=16000, dtype=np.float64, win_length=None):
    """wave [..., L] -> log-mel [..., n_mels, frames]  (datasets.py:39-48,115)."""
    return np.log(mel_power(wave, n_fft, hop, n_mels, f_min, f_max, sample_rate, dtype, win_length) + EPS32)


def crop_pad_normalize(lms, crop_frames, start, mean, std):
    """datasets.py:342-354: random crop (start given explicitly) or right zero-pad, then (x-mean)/std."""
    l = lms.shape[-1]
    if l > crop_frames:
        lms = lms[..., start:start + crop_frames]
    elif l < crop_frames:
        lms = np.pad(lms, [(0, 0)] * (lms.ndim - 1) + [(0, crop_frames - l)])
    return (lms - mean) / std
