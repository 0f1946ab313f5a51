"""HEAR wrapper oracle (TEST INFRASTRUCTURE): restates hear/sample/vit.py:88-126,160-247 and hear/utils.py:36-106 with the oracle's
own frontend and encoder.  The mel arithmetic is torchaudio's in the reference (absent here): like oracle/frontend.py that part is
parity UNPINNED.  Everything around it -- log + eps, normalisation statistics, framing, unit chunking, averaging, timestamps -- is
pinned by tests/test_oracle_golden.py against tests/golden/hear.npz, produced by running the reference's wrapper with that one
class stood in for (tests/golden/make_golden.py::gen_hear)."""
import numpy as np
import torch

from . import frontend as ofe
from . import vit as ovit

CFG = dict(sample_rate=16000, n_fft=1024, win_length=400, hop_length=160, n_mels=64, f_min=60.0, f_max=7800.0)   # hear/config.yaml


def frame_audio(audio, frame_size, hop_size, sample_rate):
    """hear/utils.py:56-106, literally: python loop over frames."""
    audio = np.pad(audio, [(0, 0), (frame_size // 2, frame_size - frame_size // 2)])
    n = audio.shape[1]
    step = hop_size / 1000.0 * sample_rate
    frames, stamps = [], []
    k, start, end = 0, 0, frame_size
    while True:
        frames.append(audio[:, start:end])
        stamps.append(k * step / sample_rate * 1000.0)
        k += 1
        start = int(round(k * step))
        end = start + frame_size
        if not end <= n:
            break
    return np.stack(frames, 1), np.broadcast_to(np.asarray(stamps, dtype=np.float32), (audio.shape[0], len(stamps)))


def to_feature(batch_audio):
    """_to_feature (hear/sample/vit.py:88-92): log-mel with the HEAR config's window -> [B, 1, 64, T]."""
    lms = ofe.logmel(np.asarray(batch_audio, dtype=np.float64), CFG["n_fft"], CFG["hop_length"], CFG["n_mels"], CFG["f_min"], CFG["f_max"],
                     CFG["sample_rate"], win_length=CFG["win_length"])
    return torch.from_numpy(lms).float().unsqueeze(1)


def encode_lms(x, params, num_heads, grid, unit_frames, patch=None):
    """encode_lms (hear/sample/vit.py:109-126): CLS embedding of every unit_frames-wide chunk of the right-padded input -> [B, n, d]."""
    pad = unit_frames - (x.shape[-1] % unit_frames)
    xp = torch.nn.functional.pad(x, (0, pad))
    return torch.stack([ovit.forward(xp[..., i * unit_frames:(i + 1) * unit_frames], params, num_heads, grid, patch=patch)
                        for i in range(xp.shape[-1] // unit_frames)], 1)


def scene_embeddings(batch_audio, params, num_heads, grid, unit_frames, patch=None):
    """get_scene_embeddings (hear/sample/vit.py:228-247): normalise by the batch's own mean / (unbiased) std, mean of the unit embeddings."""
    x = to_feature(batch_audio)
    x = (x - x.mean()) / x.std()
    return encode_lms(x, params, num_heads, grid, unit_frames, patch).mean(1)


def timestamp_embeddings(batch_audio, params, num_heads, grid, unit_frames, frame_duration=950, hop_size=50, patch=None):
    """get_timestamp_embeddings (hear/sample/vit.py:160-225), including compute_timestamp_stats' division of both statistics by the
    number of frames (hear/utils.py:47-50)."""
    frames, ts = frame_audio(np.asarray(batch_audio), int(frame_duration / 1000 * CFG["sample_rate"]), hop_size, CFG["sample_rate"])
    nb, nf, _ = frames.shape
    mel = to_feature(frames.reshape(nb * nf, -1))[:, 0]
    mean, std = mel.mean() / len(mel), mel.std() / len(mel)
    mel = ((mel - mean) / std).unsqueeze(1)
    emb = encode_lms(mel, params, num_heads, grid, unit_frames, patch).mean(1)
    return emb.reshape(nb, nf, -1), ts
