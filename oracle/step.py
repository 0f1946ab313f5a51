"""Whole-step oracle (TEST INFRASTRUCTURE): the training step of main_bt_byol.py:79-135 and the
single-network variant of main.py:86-119, on explicit state dicts, plain PyTorch CPU (autograd for
gradients), with a hand-written AdamW so the optimiser arithmetic is restated too.
"""
import torch

from . import vit, heads


def multicrop_groups(widths):
    """MultiCropWrapper.forward crop grouping (utils/utils.py:113-116): consecutive crops that share the
    last-dim length run through the backbone together.  Returns [(start, end), ...]."""
    groups, start = [], 0
    for k in range(1, len(widths) + 1):
        if k == len(widths) or widths[k] != widths[start]:
            groups.append((start, k))
            start = k
    return groups


def split_param_groups(named_params):
    """get_param_groups (utils/utils.py:136-147): no weight decay on '.bias' names and 1-D tensors;
    frozen parameters are skipped."""
    reg, noreg = [], []
    for name, p in named_params:
        if not p.requires_grad:
            continue
        (noreg if (name.endswith(".bias") or p.dim() == 1) else reg).append(name)
    return reg, noreg


FROZEN = ("pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias", "decoder_pos_embed")  # models/mae.py:190-192,202,218


def network_forward(sd, views, ncrops, num_heads, grid):
    """MultiCropWrapper(ModelWrapper(vit), BarlowTwinsHead).forward on a list of [B,1,F,T] views.
    `sd` uses the reference's key layout: backbone.encoder.* (micro fixture) or
    backbone.encoder.encoder.* (ModelWrapper->ViT->MaskedAutoencoderViT), and head.projector.*"""
    pre = "backbone.encoder.encoder." if any(k.startswith("backbone.encoder.encoder.") for k in sd) else "backbone.encoder."
    enc = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    head = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
    outs = []
    for s, e in multicrop_groups([v.shape[-1] for v in views]):
        outs.append(vit.forward(torch.cat(views[s:e]), enc, num_heads, grid))
    z, stats = heads.head_forward(torch.cat(outs), head, ncrops)
    return z, stats


class AdamW:
    """torch.optim.AdamW(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay) restated (decoupled decay)."""

    def __init__(self, lr, wd, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.wd, self.b1, self.b2, self.eps = lr, wd, betas[0], betas[1], eps
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, params, grads, decay_names):
        self.t += 1
        c1, c2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        with torch.no_grad():
            for n, g in grads.items():
                p = params[n]
                if n not in self.m:
                    self.m[n], self.v[n] = torch.zeros_like(p), torch.zeros_like(p)
                if n in decay_names:
                    p.mul_(1 - self.lr * self.wd)
                self.m[n].mul_(self.b1).add_(g, alpha=1 - self.b1)
                self.v[n].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                denom = (self.v[n].sqrt() / (c2 ** 0.5)).add_(self.eps)
                p.addcdiv_(self.m[n], denom, value=-self.lr / c1)


def lars_step(p, g, mu, lr, weight_decay, momentum=0.9, eta=0.001, weight_decay_filter=False, lars_adaptation_filter=False):
    """One LARS update of one tensor, restated from utils/utils.py:161-189 (returns new p, mu; torch CPU fp32):
    dp = g (+ wd p unless the filter excludes 1-D tensors); trust ratio q = eta |p| / |dp| when both norms are positive, else 1
    (skipped for 1-D tensors under lars_adaptation_filter); mu <- momentum mu + q dp; p <- p - lr mu."""
    one_d = p.dim() == 1
    dp = g.clone()
    if not (weight_decay_filter and one_d):
        dp = dp + weight_decay * p
    if not (lars_adaptation_filter and one_d):
        pn, un = float(torch.linalg.vector_norm(p)), float(torch.linalg.vector_norm(dp))
        q = eta * pn / un if (pn > 0.0 and un > 0.0) else 1.0
        dp = dp * q
    mu = mu * momentum + dp
    return p - lr * mu, mu


def ema_update(target, online, beta, param_names):
    """update_moving_average (utils/utils.py:328-331): parameters only, buffers untouched."""
    with torch.no_grad():
        for n in param_names:
            target[n].mul_(beta).add_(online[n], alpha=1 - beta)


def apply_bn_buffers(sd, prefix, stats):
    """Replay BatchNorm1d buffer updates (one per crop chunk, in call order) into a state dict."""
    rm, rv, nbt = sd[prefix + "running_mean"], sd[prefix + "running_var"], sd[prefix + "num_batches_tracked"]
    for mu, var, n in stats:
        rm_new, rv_new = heads.running_update(rm, rv, mu.detach(), var.detach(), n)
        rm.copy_(rm_new)
        rv.copy_(rv_new)
        nbt += 1


def _leafify(sd, trainable):
    out = {}
    for k, v in sd.items():
        t = v.clone()
        if k in trainable:
            t.requires_grad_(True)
        out[k] = t
    return out


def is_param(key):
    return not (key.endswith("running_mean") or key.endswith("running_var") or key.endswith("num_batches_tracked"))


def trainable_names(sd):
    return [k for k in sd if is_param(k) and not any(k.endswith(f) for f in FROZEN)]


def bt_byol_step(online_sd, target_sd, pred_sd, views, num_heads, grid, opt, stop_gradient, use_predictor,
                 alpha=1.0, lmbda=0.005, ema_beta=0.99):
    """One iteration of train_one_epoch (main_bt_byol.py:79-135) with L=0 local crops.
    Mutates the three state dicts in place; returns (loss, grads-of-online dict)."""
    tr_on = trainable_names(online_sd)
    on = _leafify(online_sd, set(tr_on))
    tr_pr = [k for k in pred_sd if is_param(k)] if use_predictor else []
    pr = _leafify(pred_sd, set(tr_pr))
    tr_tg = [] if stop_gradient else trainable_names(target_sd)
    tg = _leafify(target_sd, set(tr_tg))

    zo, st_on = network_forward(on, views[:2], 2, num_heads, grid)
    apply_bn_buffers(online_sd, "head.projector.1.", st_on)
    if use_predictor:
        zo, st_pr = heads.predictor_forward(zo, pr, ncrops=1)
        apply_bn_buffers(pred_sd, "predictor.1.", st_pr)
    zt, st_tg = network_forward(tg, views, 2, num_heads, grid)
    apply_bn_buffers(target_sd, "head.projector.1.", st_tg)
    loss, _ = heads.bt_forward(zo, zt, ncrops=2, ngcrops_each=2, alpha=alpha, lmbda=lmbda)
    if stop_gradient:  # EMA happens BEFORE the optimiser step (main_bt_byol.py:121-126)
        ema_update(target_sd, online_sd, ema_beta, [k for k in online_sd if is_param(k)])
    leaves = [on[k] for k in tr_on] + [pr[k] for k in tr_pr] + [tg[k] for k in tr_tg]
    gs = torch.autograd.grad(loss, leaves, allow_unused=True)
    names = [("online", k) for k in tr_on] + [("pred", k) for k in tr_pr] + [("target", k) for k in tr_tg]
    store = {"online": online_sd, "pred": pred_sd, "target": target_sd}
    params, grads, decay = {}, {}, set()
    for (which, k), g in zip(names, gs):
        if g is None:
            continue
        key = which + ":" + k
        params[key], grads[key] = store[which][k], g
        if not (k.endswith(".bias") or store[which][k].dim() == 1):
            decay.add(key)
    opt.step(params, grads, decay)
    return float(loss.detach()), {k: g for (w, k), g in zip(names, gs) if w == "online" and g is not None}


def bt_step(sd, views, num_heads, grid, opt, alpha=1.0, lmbda=0.005):
    """Single-network Barlow Twins step (main.py:86-119 with L=0): teacher = view 1, student = view 2,
    one loss term forward_loss(z_view1, z_view2).  This is BASELINE config 2/3's step."""
    tr = trainable_names(sd)
    leaf = _leafify(sd, set(tr))
    z, st = network_forward(leaf, views, 2, num_heads, grid)
    apply_bn_buffers(sd, "head.projector.1.", st)
    z1, z2 = z.chunk(2)
    loss, _ = heads.bt_forward_loss(z1, z2, alpha, lmbda)
    gs = torch.autograd.grad(loss, [leaf[k] for k in tr], allow_unused=True)
    params = {k: sd[k] for k, g in zip(tr, gs) if g is not None}
    grads = {k: g for k, g in zip(tr, gs) if g is not None}
    decay = {k for k in params if not (k.endswith(".bias") or sd[k].dim() == 1)}
    opt.step(params, grads, decay)
    return float(loss.detach()), grads


def mae_step(sd, views, num_heads, grid, opt, mask=None, noise=None, mask_ratio=0.75, dec_heads=6, alpha=1.0, lmbda=0.005):
    """main.py:69-125 with `--mask --masked_recon` and L = 0 (BASELINE config 5's step): view 1 through the masked encoder + MAE
    decoder (teacher side; adds the reconstruction loss, models/mae.py:464-468), view 2 through the unmasked encoder, one
    Barlow-Twins term forward_loss(teacher, student).  The masking randomness is explicit: `mask` [B, L] or `noise` [B, L]."""
    tr = trainable_names(sd)
    leaf = _leafify(sd, set(tr))
    pre = "backbone.encoder.encoder." if any(k.startswith("backbone.encoder.encoder.") for k in sd) else "backbone.encoder."
    enc = {k[len(pre):]: v for k, v in leaf.items() if k.startswith(pre)}
    head = {k[len("head."):]: v for k, v in leaf.items() if k.startswith("head.")}
    lat_t, recon = vit.forward(views[0], enc, num_heads, grid, mask=mask, noise=noise, mask_ratio=mask_ratio, masked_recon=True,
                               dec_heads=dec_heads)
    lat_s = vit.forward(views[1], enc, num_heads, grid)
    zt, st_t = heads.head_forward(lat_t, head, 1)
    zs, st_s = heads.head_forward(lat_s, head, 1)
    apply_bn_buffers(sd, "head.projector.1.", st_t + st_s)
    bt, _ = heads.bt_forward(zs, zt, 2, ngcrops_each=1, alpha=alpha, lmbda=lmbda)
    loss = bt + recon
    gs = torch.autograd.grad(loss, [leaf[k] for k in tr], allow_unused=True)
    params = {k: sd[k] for k, g in zip(tr, gs) if g is not None}
    grads = {k: g for k, g in zip(tr, gs) if g is not None}
    decay = {k for k in params if not (k.endswith(".bias") or sd[k].dim() == 1)}
    opt.step(params, grads, decay)
    return float(loss.detach()), grads
