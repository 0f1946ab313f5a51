"""HEAR common-API wrapper for the ViT / ViTC encoders (hear/sample/vit.py:41-247 of the reference), on the MI355X kernels:
`load_model`, `get_scene_embeddings`, `get_timestamp_embeddings`, and `ViTModelWrapper` with the reference's checkpoint-key mapping
(hear/sample/vit.py:64-77; linear.py:114-133 is `load_encoder_state_dict`).  The log-mel frontend is frontend.MelSpectrogram
(sa_logmel_fwd), the batch normalisation sa_normalize_batch, the encoder the engine; torch only moves data and averages nothing:
the means over units / frames go through sa_token_group_sum.
"""
import os
from typing import List, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import mae, ops
from ..augmentations import NormalizeBatch
from ..frontend import MelSpectrogram
from . import utils

TIMESTAMP_FRAME_DUR = 950     # ms
TIMESTAMP_HOP_SIZE = 50       # ms
BATCH_SIZE = 512              # frames per encoder batch for timestamp embeddings
DEFAULT_CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config.yaml")


def get_to_melspec(cfg):
    return MelSpectrogram(sample_rate=cfg.sample_rate, n_fft=cfg.n_fft, win_length=cfg.win_length, hop_length=cfg.hop_length, n_mels=cfg.n_mels,
                          f_min=cfg.f_min, f_max=cfg.f_max, power=2)


def clean_state_dict(sd, prefixes):
    """The reference's key clean-up: unwrap {'model': ...}, then keep the keys under the first prefix that matches, stripped of it;
    no match: the dict as it is.  DDP's `module.` wrapper (main_bt_byol.py:494 saves the DDP state_dict) is dropped first."""
    if 'model' in sd.keys():
        sd = sd.get('model')
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    for pre in prefixes:
        clean = {k.replace(pre, ""): v for k, v in sd.items() if pre in k}
        if clean:
            return clean
    return sd


def load_encoder_state_dict(model, sd):
    """linear.py:114-133: checkpoint of the pre-training driver -> the bare encoder (`ModelWrapper(args).encoder`)."""
    model.load_state_dict(clean_state_dict(sd, ("backbone.encoder.", "encoder.encoder.")), strict=True)
    return model


def mean_over_dim1(x: Tensor) -> Tensor:
    """torch.mean(x, dim=1) of a [B, n, d] fp32 tensor through sa_token_group_sum."""
    B, n, d = x.shape
    out = torch.empty(B, 1, d, device=x.device)
    ops.token_group_sum(x.contiguous(), 0, 1, 0, n, 1.0 / n, out)
    return out.view(B, d)


class ViTModelWrapper(nn.Module):
    def __init__(self, cfg, model_type, model_file_path, patch_size):
        super().__init__()
        self.cfg = cfg
        self.use_cls = True if self.cfg.use_cls is None else self.cfg.use_cls
        self.sample_rate = cfg.sample_rate
        embed_size = self._get_model(model_type, patch_size)
        if model_file_path != "":
            self._load_weights(model_file_path)
        self.scene_embedding_size = embed_size
        self.timestamp_embedding_size = embed_size * self.model.grid_size()[0]
        self.to_melspec = get_to_melspec(cfg)
        self._norm = NormalizeBatch()

    def _get_model(self, model_type, patch_size):
        c = "vitc" in model_type
        size = model_type.split('_')[-1]
        self.model = mae.get_mae_vit(size, patch_size, c)
        return self.model.embed_dim

    def _load_weights(self, model_file_path):
        sd = torch.load(model_file_path, map_location='cpu')
        self.model.load_state_dict(clean_state_dict(sd, ("backbone.encoder.encoder.", "encoder.encoder.")), strict=True)

    def _get_timestamps(self, batch_audio, x):
        audio_len = len(batch_audio[0])
        sec = audio_len / self.cfg.sample_rate
        x_len = len(x[0])
        step = sec / x_len
        ts = torch.tensor([step * i for i in range(x_len)]).unsqueeze(0)
        return ts.repeat(len(batch_audio), 1)

    def _to_feature(self, batch_audio):
        return self.to_melspec(batch_audio)                      # [B, 1, n_mels, T]: log(mel + eps) fused into the kernel

    def _normalize_batch(self, x):
        return self._norm(x)                                     # (x - mean) / std over the whole batch (unbiased std), one channel

    def _to_normalized_spec(self, batch_audio):
        return self._normalize_batch(self._to_feature(batch_audio))

    def encode_lms(self, x):
        """Right-pad to a multiple of the encoder's unit width (a whole extra unit when it already is one), CLS embedding per unit
        -> [b, n_units, d] (hear/sample/vit.py:109-126)."""
        unit_frames = self.model.img_size[1]
        pad_frames = unit_frames - (x.shape[-1] % unit_frames)
        xp = torch.zeros(*x.shape[:-1], x.shape[-1] + pad_frames, device=x.device, dtype=x.dtype)
        xp[..., :x.shape[-1]] = x
        n_units = xp.shape[-1] // unit_frames
        out = torch.empty(x.shape[0], n_units, self.model.embed_dim, device=x.device)
        for i in range(n_units):
            out[:, i] = self.model(xp[..., i * unit_frames:(i + 1) * unit_frames].contiguous())
        return out

    def encode(self, batch_audio):
        return self.encode_lms(self._to_normalized_spec(batch_audio))


def load_model(model_file_path: str = "", model_type: str = "vitc_base", patch_size: str = "16x8", cfg_path: str = DEFAULT_CFG) -> torch.nn.Module:
    cfg = utils.load_yaml_config(cfg_path)
    patch = [int(patch_size.split("x")[0]), int(patch_size.split("x")[-1])]
    model = ViTModelWrapper(cfg, model_type, model_file_path, patch)
    if torch.cuda.is_available():
        model.cuda()
    return model


def get_timestamp_embeddings(audio_list: List, model: torch.nn.Module, frame_duration: float = TIMESTAMP_FRAME_DUR,
                             hop_size: float = TIMESTAMP_HOP_SIZE, cfg_path: str = DEFAULT_CFG) -> Tuple[Tensor, Tensor]:
    """Embeddings at regular intervals centred on the returned timestamps (ms): (n_sounds, n_timestamps, d), (n_sounds, n_timestamps)."""
    cfg = utils.load_yaml_config(cfg_path)
    to_melspec = get_to_melspec(cfg)
    model = model.to(audio_list[0].device)
    frames, timestamps = utils.frame_audio(audio_list, frame_size=int((frame_duration / 1000) * cfg.sample_rate), hop_size=hop_size,
                                           sample_rate=cfg.sample_rate)
    audio_batches, num_frames, _ = frames.shape
    frames = frames.flatten(end_dim=1).contiguous()
    melspec_frames = to_melspec(frames)                                   # [n, 1, n_mels, T], log already applied
    melspec_frames = utils.normalize_like_timestamp_stats(melspec_frames[:, 0]).unsqueeze(1)
    model.eval()
    embs = []
    with torch.no_grad():
        for i in range(0, melspec_frames.shape[0], BATCH_SIZE):
            embs.append(mean_over_dim1(model.encode_lms(melspec_frames[i:i + BATCH_SIZE])))
    embeddings = torch.cat(embs, dim=0).unflatten(0, (audio_batches, num_frames))
    return embeddings, timestamps


def get_scene_embeddings(audio_list: List, model: torch.nn.Module) -> Tensor:
    """One embedding per clip: mean over units of the CLS embeddings (n_sounds, model.scene_embedding_size)."""
    device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    model = model.to(device)
    model.eval()
    with torch.no_grad():
        return mean_over_dim1(model.encode(audio_list.to(device) if torch.is_tensor(audio_list) else audio_list))
