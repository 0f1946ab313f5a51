"""hear/utils.py of the reference: YAML config, framing of audio for timestamp embeddings, normalisation statistics."""
from pathlib import Path
from typing import Tuple

import torch
import yaml
from torch import Tensor

from .. import ops


class AttrDict(dict):
    """EasyDict stand-in: keys readable as attributes (hear/utils.py:10-17 returns an EasyDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = dict.__setitem__


def load_yaml_config(path_to_config):
    path_to_config = Path(path_to_config)
    assert path_to_config.is_file()
    with open(path_to_config) as f:
        return AttrDict(yaml.safe_load(f))


def frame_audio(audio: Tensor, frame_size: int, hop_size: float, sample_rate: int) -> Tuple[Tensor, Tensor]:
    """hear/utils.py:56-106: frames of frame_size samples centred every hop_size ms (zero padding of half a frame on both sides);
    returns (frames [n_sounds, n_frames, frame_size], timestamps in ms [n_sounds, n_frames]).  Index bookkeeping + copies only."""
    audio = torch.nn.functional.pad(audio, (frame_size // 2, frame_size - frame_size // 2))
    num_padded = audio.shape[1]
    step = hop_size / 1000.0 * sample_rate
    starts, stamps, k = [], [], 0
    start = 0
    while start + frame_size <= num_padded:
        starts.append(start)
        stamps.append(k * step / sample_rate * 1000.0)
        k += 1
        start = int(round(k * step))
    idx = torch.tensor(starts, device=audio.device)[:, None] + torch.arange(frame_size, device=audio.device)[None, :]
    frames = audio[:, idx]                                               # [n_sounds, n_frames, frame_size]
    ts = torch.tensor(stamps, dtype=torch.float32).expand(audio.shape[0], -1)
    return frames, ts


def normalize_like_timestamp_stats(melspec: Tensor) -> Tensor:
    """hear/utils.py:36-53 + hear/sample/vit.py:203-205: mean and std of ALL frames, each divided by len(melspec) (the reference's
    statistics, kept as they are), then (x - mean) / std -- one pass of the normalise kernel with stat_div = number of frames."""
    x = melspec.contiguous().float()
    out = torch.empty_like(x)
    ws = torch.zeros(2, dtype=torch.float64, device=x.device)
    ops.normalize_batch(x, out, 0.0, ws, 0.0, stat_div=float(len(melspec)))
    return out
