"""Training-runtime helpers with the reference's names (utils/utils.py): off_diagonal :23-27, MultiCropWrapper
:98-133, get_param_groups :136-147, EMA / update_moving_average :317-331, init_distributed_mode :335-361,
model_setup_ddp :410-417 and the rank helpers."""
import torch
import torch.nn as nn

from . import dist as sdist
from . import ops
from .engine import BF16_WEIGHTS
from .dist import get_rank, get_world_size, is_dist_avail_and_initialized, is_main_process  # noqa: F401  (re-exported)


def off_diagonal(x):
    n, m = x.shape
    assert n == m
    return x.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten()


class MultiCropWrapper(nn.Module):
    """Same-resolution crops are concatenated and run through the backbone together; the head sees all outputs."""

    def __init__(self, backbone, head):
        super().__init__()
        self.backbone = backbone
        self.head = head

    def forward(self, x, ncrops=1, **kwargs):
        recon_loss = None
        if not isinstance(x, list):
            x = [x]
        widths = [inp.shape[-1] for inp in x]
        outs, start = [], 0
        for end in range(1, len(x) + 1):
            if end < len(x) and widths[end] == widths[start]:
                continue
            batch = torch.cat(x[start:end]) if end - start > 1 else x[start]
            _out = self.backbone(batch, **kwargs)
            if isinstance(_out, tuple):
                recon_loss = _out[1] if recon_loss is None else recon_loss + _out[1]
                _out = _out[0]
            outs.append(_out)
            start = end
        output = torch.cat(outs) if len(outs) > 1 else outs[0]
        if recon_loss is not None:
            return self.head(output, ncrops), recon_loss
        return self.head(output, ncrops)


def get_param_groups(model):
    regularized, not_regularized = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if name.endswith(".bias") or len(param.shape) == 1:
            not_regularized.append(param)
        else:
            regularized.append(param)
    return [{'params': regularized}, {'params': not_regularized, 'weight_decay': 0.}]


class EMA():
    def __init__(self, beta):
        super().__init__()
        self.beta = beta

    def update_average(self, old, new):
        if old is None:
            return new
        out = old.clone()
        ops.ema_update(out, new.contiguous(), self.beta)
        return out


def update_moving_average(ema_updater, ma_model, current_model):
    """target = beta * target + (1 - beta) * online over .parameters() (buffers untouched), in place on the GPU."""
    with torch.no_grad():
        for current_params, ma_params in zip(current_model.parameters(), ma_model.parameters()):
            ops.ema_update(ma_params.data, current_params.data, ema_updater.beta)
            BF16_WEIGHTS.mark_modified(ma_params)


def init_distributed_mode(cfg):
    """Sets cfg.rank / cfg.gpu / cfg.world_size from the torchrun environment (RCCL via backend 'nccl')."""
    cfg.rank, cfg.gpu, cfg.world_size = sdist.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(cfg.gpu)


def model_setup_ddp(gpu, model):
    """The reference converts BN to SyncBN and wraps in DistributedDataParallel.  Here both are already built in:
    the head's BN exchanges statistics itself (functional.MlpBnReluFn) and gradients are summed by dp.GradSync,
    so this returns the module unchanged, keeping the (wrapped, unwrapped) return shape."""
    return model, model


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)
