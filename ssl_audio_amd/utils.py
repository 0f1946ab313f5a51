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


def _cat_crops(crops):
    """torch.cat(crops) -- without the copy when the crops ARE one tensor already: `BatchedPairAugment` writes the two global views of a
    batch as [2, B, 1, F, T], so views[0] and views[1] are adjacent slices of one allocation and their concatenation is a view of it
    (65 MB per ViT-B step not copied).  Anything else (other strides, gaps, autograd history) takes torch.cat."""
    if len(crops) == 1:
        return crops[0]
    first = crops[0]
    ok = first.is_contiguous() and not first.requires_grad and first.dim() >= 1
    if ok:
        step = first.numel() * first.element_size()
        for k, c in enumerate(crops):
            if (c.shape != first.shape or c.dtype != first.dtype or c.device != first.device or not c.is_contiguous() or c.requires_grad
                    or c.untyped_storage().data_ptr() != first.untyped_storage().data_ptr() or c.data_ptr() != first.data_ptr() + k * step):
                ok = False
                break
    if not ok:
        return torch.cat(crops)
    return first.as_strided((len(crops) * first.shape[0],) + tuple(first.shape[1:]), first.stride(), first.storage_offset())


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
            batch = _cat_crops(x[start:end])
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
    """utils/utils.py:410-417: the reference converts BN to SyncBN and wraps the network in DistributedDataParallel.
    Here SyncBN is built in (functional.MlpBnReluFn exchanges the BatchNorm statistics itself), and the wrapper returned is
    dist.GradSumParallel: after each backward it all-reduces (SUM: the loss is global-batch exact) the wrapped parameters'
    gradients in buckets on a side stream, so the driver's own optimizer (torch.optim.AdamW, main_bt_byol.py:302-313) steps
    on global gradients and the replicas stay identical.  Returns (wrapper, wrapped) like the reference."""
    wrapped = sdist.GradSumParallel(model)
    return wrapped, wrapped.module


def load_checkpoint(ckpt_path, model, predictor, optimizer):
    """utils/utils.py:37-46 (`main.py --resume_path`): model, predictor and optimizer state from a checkpoint; returns the epoch to resume at."""
    ckpt = torch.load(ckpt_path, map_location=torch.device('cpu'))
    model.load_state_dict(ckpt['model'])
    predictor.load_state_dict(ckpt['predictor'])
    optimizer.load_state_dict(ckpt['optimizer'])
    return ckpt['epoch']


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


# ------------------------------------------------------------------------------------------------ optimiser family
class LARS(torch.optim.Optimizer):
    """`utils.LARS` (utils/utils.py:150-189; `--optimizer LARS`, main_bt_byol.py:326-345): SGD with momentum on
    dp = g + wd * p, rescaled per tensor by the trust ratio eta * |p| / |dp|.  Same constructor and group keys; the
    arithmetic is two HIP launches per tensor (`sa_lars_step`).  1-D tensors skip decay / adaptation when the filters ask."""

    def __init__(self, params, lr, weight_decay=0, momentum=0.9, eta=0.001, weight_decay_filter=False, lars_adaptation_filter=False):
        defaults = dict(lr=lr, weight_decay=weight_decay, momentum=momentum, eta=eta, weight_decay_filter=weight_decay_filter,
                        lars_adaptation_filter=lars_adaptation_filter)
        super().__init__(params, defaults)

    def exclude_bias_and_norm(self, p):
        return p.ndim == 1

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise ValueError("LARS: parameters and gradients must be contiguous fp32 device tensors")
                one_d = self.exclude_bias_and_norm(p)
                wd = 0.0 if (group["weight_decay_filter"] and one_d) else group["weight_decay"]
                adapt = not (group["lars_adaptation_filter"] and one_d)
                st = self.state[p]
                if "mu" not in st:
                    st["mu"] = torch.zeros_like(p)
                    st["scratch"] = torch.empty(2, dtype=torch.float32, device=p.device)
                ops.lars_step(p, p.grad, st["mu"], group["lr"], wd, group["momentum"], group["eta"], adapt, st["scratch"])
                BF16_WEIGHTS.mark_modified(p)


def generate_random(l, h, p):
    """`utils.generate_random` (utils/utils.py:30-33, main.py:76): with probability p the mask ratio is 0, otherwise U(l, h).  One
    draw from Python's `random` decides, and only the non-zero branch consumes a draw from numpy's global generator -- the call order
    a seeded reference run has."""
    import random
    import numpy as np
    keep_zero = not (random.random() > p)
    return 0 if keep_zero else np.random.uniform(l, h)


def adjust_learning_rate(args, optimizer, loader, step):
    """`utils.adjust_learning_rate` (utils/utils.py:47-65, used by main.py:52): linear warm-up over epochs // 100 epochs, then a
    cosine from base to base / 1000 stretched over 1.25 x the run; base = batch_size / 128 and the result multiplies args.lr
    (LARS: args.lr_weights / args.lr_biases for groups 0 / 1) -- the reference's double batch-size scaling, kept as is."""
    import math
    per_epoch = len(loader)
    horizon = args.epochs * per_epoch * 1.25
    warm = int(args.epochs / 100) * per_epoch
    base = args.batch_size / 128
    if step < warm:
        scale = base * step / warm
    else:
        frac = (step - warm) / (horizon - warm)
        w = 0.5 * (1.0 + math.cos(math.pi * frac))
        scale = base * w + base * 0.001 * (1.0 - w)
    if args.optimizer == "LARS":
        optimizer.param_groups[0]["lr"] = scale * args.lr_weights
        optimizer.param_groups[1]["lr"] = scale * args.lr_biases
    else:
        for group in optimizer.param_groups:
            group["lr"] = scale * args.lr


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """Per-iteration table (utils/utils.py:68-78): linear warm-up, then half a cosine from base_value to final_value."""
    import numpy as np
    n_warm = warmup_epochs * niter_per_ep
    n_rest = epochs * niter_per_ep - n_warm
    warm = np.linspace(start_warmup_value, base_value, n_warm) if warmup_epochs > 0 else np.array([])
    k = np.arange(n_rest)
    rest = final_value + 0.5 * (base_value - final_value) * (1.0 + np.cos(np.pi * k / n_rest))
    return np.concatenate((warm, rest))


def sine_scheduler_increase(final_value, epochs, niter_per_ep, warmup_epochs=0, warmup_value=0):
    """Mask-ratio table (utils/utils.py:81-91, main.py:442): a constant warm-up at warmup_value, then
    (final - warmup) * sin(pi/2 * k/n) -- which restarts from 0, not from warmup_value (reference behaviour, kept)."""
    import numpy as np
    n_warm = warmup_epochs * niter_per_ep
    n_rest = epochs * niter_per_ep - n_warm
    warm = np.full(n_warm, float(warmup_value)) if warmup_epochs > 0 else np.array([])
    k = np.arange(n_rest)
    rest = (final_value - warmup_value) * np.sin((np.pi / 2.0) * (k / n_rest))
    return np.concatenate((warm, rest))


# ------------------------------------------------------------------------------------------------ eval path
@torch.no_grad()
def encode_vit(model, x, split_frames=True, use_cls=True):
    """`utils.encode_vit` (utils/utils.py:278-314): embed a spectrogram of any length with a ViT trained on unit_frames-wide
    inputs.  The input is right-padded to a multiple of unit_frames (a FULL extra unit when it already is one -- reference
    behaviour, kept); each unit goes through the encoder; CLS embeddings are averaged over units (use_cls) or the patch tokens
    are laid out as [time, freq x d], the padded tail dropped and the rest averaged over time.  Encoder and reductions are HIP
    (sa_token_group_sum); padding / stacking are data movement."""
    grid_f, grid_t = model.grid_size()
    d = model.embed_dim
    unit = model.img_size[1]
    pad = unit - (x.shape[-1] % unit)
    xp = torch.zeros(*x.shape[:-1], x.shape[-1] + pad, device=x.device, dtype=x.dtype)
    xp[..., :x.shape[-1]] = x
    if not split_frames:
        return model(xp)
    n_units = xp.shape[-1] // unit
    B = x.shape[0]
    if use_cls:
        embs = torch.empty(B, n_units, d, device=x.device)
        for i in range(n_units):
            embs[:, i] = model(xp[..., i * unit:(i + 1) * unit].contiguous())
        out = torch.empty(B, 1, d, device=x.device)
        ops.token_group_sum(embs, 0, 1, 0, n_units, 1.0 / n_units, out)
        return out.view(B, d)
    dropped = int(grid_t * pad / unit)                      # embedding frames that came from padding (at most one unit's worth)
    total = n_units * grid_t - dropped
    out = torch.zeros(B, grid_f, d, device=x.device)
    for i in range(n_units):
        count = min(grid_t, max(0, total - i * grid_t))
        if count == 0:
            break
        tok = model(xp[..., i * unit:(i + 1) * unit].contiguous(), return_all=True)      # [B, 1 + grid_f * grid_t, d]
        ops.token_group_sum(tok.contiguous(), 1, grid_f, grid_t, count, 1.0 / total, out, accumulate=True)
    return out.view(B, grid_f * d)
