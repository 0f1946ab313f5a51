"""ResNet-18 encoders (models/resnet.py; BASELINE config 1) on the HIP path.

  kernels of resnet.hip against torch (MaxPool2d(3, 2, 1) with overlapping windows, strided subsampling, add + ReLU, average pool)
  `resnet18` and `resnet18_ReGP_NRF` forward / backward against tests/golden/resnet.npz (outputs of the reference) and against the
     oracle in bf16-mirror mode (gradients of all 62 parameters)
  BASELINE config 1: ResNet-18, 1 s clips (64 x 96 log-mel crops), batch 32, Barlow Twins -- the reference's own CPU-runnable case --
     through ModelWrapper + MultiCropWrapper + BarlowTwinsHead + BarlowTwinsLoss + LARS, against the oracle on the same weights and views
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gradcheck import check_step_gradients, cosine  # noqa: E402  (tests/gradcheck.py)
from ssl_audio_amd import ops  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    ops.lib()
    return torch.device("cuda:0")


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(x):           # [B, C, H, W] -> [B*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def nchw(m, B, H, W):  # [B*H*W, C] -> [B, C, H, W]
    return m.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("shape", [(3, 16, 7, 10), (2, 64, 32, 48), (2, 8, 1, 5), (1, 24, 9, 9)])
def test_maxpool3s2_matches_torch(dev, shape):
    """Forward (bf16 in, bf16 + fp32 out) and the summing backward against F.max_pool2d(3, 2, 1); odd sizes, a single row, and an input
    with many exact ties (bf16-rounded small integers) so that the first-maximum rule decides where gradients go."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H * 100 + W)
    for ties in (False, True):
        x = torch.randn(B, C, H, W, generator=g)
        if ties:
            x = torch.randint(-2, 3, (B, C, H, W), generator=g).float()
        x = x.to(BF16).float().requires_grad_(True)
        ref = F.max_pool2d(x, 3, 2, 1)
        dyr = torch.randn(ref.shape, generator=g)
        ref.backward(dyr)
        Ho, Wo = ops.pool_out_size(H), ops.pool_out_size(W)
        assert ref.shape[-2:] == (Ho, Wo)
        x16 = nhwc(x.detach()).to(BF16).to(dev)
        y16 = torch.empty(B * Ho * Wo, C, dtype=BF16, device=dev)
        y32 = torch.empty(B * Ho * Wo, C, device=dev)
        idx = torch.empty(B * Ho * Wo, C, dtype=torch.uint8, device=dev)
        ops.maxpool3s2_fwd(x16, B, H, W, C, y16, y32, idx)
        assert torch.equal(nchw(y32.cpu(), B, Ho, Wo), ref.detach()) and torch.equal(y16.float().cpu(), y32.cpu())
        dx = torch.empty(B * H * W, C, device=dev)
        ops.maxpool3s2_bwd(nhwc(dyr).to(dev), idx, B, H, W, C, dx)
        assert rel(nchw(dx.cpu(), B, H, W), x.grad) < 1e-6


@pytest.mark.parametrize("stride", [(2, 2), (1, 2), (2, 1)])
def test_subsample_and_elementwise(dev, stride):
    B, C, H, W = 2, 64, 9, 12
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g).to(BF16).float()
    sh, sw = stride
    Ho, Wo = (H - 1) // sh + 1, (W - 1) // sw + 1
    y = torch.empty(B * Ho * Wo, C, dtype=BF16, device=dev)
    ops.subsample_fwd(nhwc(x).to(BF16).to(dev), B, H, W, C, stride, y)
    assert torch.equal(nchw(y.float().cpu(), B, Ho, Wo), x[:, :, ::sh, ::sw])
    # backward: adds into an existing gradient at the sampled pixels only (dy with a padded leading dimension)
    base = torch.randn(B * H * W, C, generator=g)
    dy = torch.randn(B * Ho * Wo, C + 8, generator=g).to(BF16)
    dx = base.clone().to(dev)
    ops.subsample_bwd_add(dy.to(dev)[:, :C], B, H, W, C, stride, dx)
    ref = nchw(base, B, H, W)
    ref[:, :, ::sh, ::sw] += nchw(dy[:, :C].float(), B, Ho, Wo)
    assert rel(nchw(dx.cpu(), B, H, W), ref) < 1e-6
    # add + ReLU, its mask, average pool
    z, idn = torch.randn(B * H * W, C, generator=g), torch.randn(B * H * W, C, generator=g)
    y32, y16 = torch.empty(B * H * W, C, device=dev), torch.empty(B * H * W, C, dtype=BF16, device=dev)
    ops.add_relu_fwd(z.to(dev), idn.to(dev), y32, y16)
    assert torch.equal(y32.cpu(), F.relu(z + idn)) and torch.equal(y16.cpu(), F.relu(z + idn).to(BF16))
    d1, d2 = torch.randn(B * H * W, C, generator=g), torch.randn(B * H * W, C, generator=g)
    ds = torch.empty(B * H * W, C, device=dev)
    ops.relu_bwd(d1.to(dev), d2.to(dev), y32, ds)
    assert torch.equal(ds.cpu(), torch.where(F.relu(z + idn) > 0, d1 + d2, torch.zeros(())))
    ops.relu_bwd(d1.to(dev), None, y32, ds)
    assert torch.equal(ds.cpu(), torch.where(F.relu(z + idn) > 0, d1, torch.zeros(())))
    out = torch.empty(B, C, device=dev)
    ops.avgpool_fwd(y32, B, H * W, C, out)
    assert rel(out, F.relu(z + idn).view(B, H * W, C).mean(1)) < 1e-6
    dxa = torch.empty(B * H * W, C, device=dev)
    ops.avgpool_bwd(out, B, H * W, C, dxa)
    assert rel(dxa.view(B, H * W, C), (out.cpu() / (H * W))[:, None, :].expand(B, H * W, C)) < 1e-6


# ------------------------------------------------------------------------------------------------ networks vs the reference's outputs
def load_net(variant, dev, seed, aseed):
    from oracle import resnet as oresnet
    from ssl_audio_amd import resnet
    sd = oresnet.init_state(variant, seed, aseed)
    net = (resnet.resnet18 if variant == "resnet18" else resnet.resnet18_ReGP_NRF)()
    full = dict(sd)
    for k in net.state_dict():
        if k.endswith("num_batches_tracked"):
            full[k] = torch.zeros((), dtype=torch.long)
    net.load_state_dict(full, strict=True)
    return net.to(dev).train(), sd


@pytest.mark.parametrize("variant", ["resnet18", "resnet18_ReGP_NRF"])
def test_resnet_golden(dev, variant):
    """Same weights and input as the reference run that produced tests/golden/resnet.npz: embedding rel <= 3e-2 against the reference (<= 4e-2 against the oracle in bf16-mirror mode)
    (bf16 operands through 20 convolutions), gradient norms of all 62 parameters within 5 %, small-parameter gradients by cosine, BatchNorm running buffers;
    then every gradient against the oracle in bf16-mirror mode with the sensitivity-relative bound of tests/gradcheck.py."""
    from oracle import resnet as oresnet, rounding as R
    g = np.load(os.path.join(GOLD, "resnet.npz"))
    seed, aseed = [int(v) for v in g[f"{variant}.affine_seed"]]
    net, sd = load_net(variant, dev, seed, aseed)
    x = torch.from_numpy(g[f"{variant}.x"])
    y = net(x.to(dev))
    ref = torch.from_numpy(g[f"{variant}.y"])
    assert y.shape == ref.shape
    e = rel(y, ref)
    w = torch.linspace(-1, 1, y.numel()).reshape(y.shape)
    (y * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    got = {n: p.grad.detach().float().cpu() for n, p in net.named_parameters()}
    assert all(torch.isfinite(v).all() for v in got.values())
    names = [str(n) for n in g[f"{variant}.grad_names"]]
    assert names == list(got.keys())
    # ---- all gradients against the oracle: fp32 (== the reference, tests/test_oracle_golden.py) and bf16-mirror mode
    def oracle_grads():
        p = {k: v.clone().requires_grad_("running" not in k) for k, v in sd.items()}
        yo = oresnet.forward(x, p, variant)
        (yo * w).sum().backward()
        return {k: v.grad for k, v in p.items() if v.grad is not None}, yo.detach()
    fgrads, _ = oracle_grads()
    with R.mirror_hip_bf16():
        mgrads, y_mirror = oracle_grads()
    em = rel(y, y_mirror)
    for n in names[:4] + names[-4:]:
        print(f"   {n:34s} rel(hip, mirror) {rel(got[n], mgrads[n]):.3f}  rel(mirror, fp32) {rel(mgrads[n], fgrads[n]):.3f}  cos(hip, fp32) {cosine(got[n], fgrads[n]):.4f}")
    worst = 0.0
    for n, nrm in zip(names, g[f"{variant}.grad_norms"]):
        worst = max(worst, abs(float(got[n].double().norm()) - nrm) / max(nrm, 1e-12))
    print(f"{variant}: embedding rel {e:.2e} vs the reference, {em:.2e} vs the bf16-mirror oracle; worst gradient-norm deviation vs the reference {worst:.3f}")
    # 20 convolutions deep on bf16 operands with BatchNorm over 4 clips: the mirror oracle itself sits 2e-2 from the fp32 reference, and
    # after a few layers two bf16 evaluations no longer round the same way (different fp32 summation orders), so the mirror is no closer
    # to the HIP result than the reference is -- both bounds are rounding-noise bounds; the discriminating check is the gradient one below
    assert e <= 3e-2 and em <= 4e-2, (e, em)
    check_step_gradients(variant, got, mgrads, fgrads, 55)
    assert worst <= 0.25, worst
    for k in g.files:                                              # BatchNorm buffers after one training-mode forward
        if k.startswith(f"{variant}.after.") and "running" in k:
            name = k[len(f"{variant}.after."):]
            assert rel(net.state_dict()[name], g[k]) <= 2e-2, name


# ------------------------------------------------------------------------------------------------ BASELINE config 1
@pytest.mark.parametrize("variant", ["resnet18", "resnet18_ReGP_NRF"])
def test_resnet_eval_mode_golden(dev, variant):
    """`net.eval()` = nn.BatchNorm2d on its RUNNING statistics (the reference's kNN / linear-eval / HEAR path): embedding against the
    reference's eval forward (tests/golden/bn_eval.npz), every buffer untouched, and a clip's embedding independent of its batch."""
    g = np.load(os.path.join(GOLD, "bn_eval.npz"))
    seed, aseed = [int(v) for v in g[f"{variant}.affine_seed"]]
    net, _ = load_net(variant, dev, seed, aseed)
    bufs = {k[len(variant) + 5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{variant}.buf.")}
    net.load_state_dict(bufs, strict=False)
    net.eval()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.from_numpy(g[f"{variant}.x"]).to(dev)
    with torch.no_grad():
        y = net(x)
        y0 = net(x[:1])
    e = rel(y, g[f"{variant}.y"])
    print(f"{variant} eval: embedding rel {e:.2e} vs the reference")
    assert e <= 3e-2, e
    assert rel(y0, y[:1]) < 1e-6                                   # batch composition does not enter
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k


@pytest.mark.parametrize("depth", ["stem", "stem+layer1"])
@pytest.mark.parametrize("training", [True, False])
def test_resnet_stem_gradients_flat_bound(dev, depth, training):
    """ResNet stem gradients with a FLAT bound.  Through all 17 convolutions the gradients of a randomly
    initialised ResNet-18 are chaotic at any batch size and in eval mode alike (next test: the oracle's own fp32-vs-bf16 distance is
    25-45 %, and two bf16 evaluations that differ only in fp32 summation order sit 10-20 % apart), so the flat bound is asserted on the
    network cut where it IS conditioned: the ResNet-C stem (three 3x3 convolutions + BatchNorm2d + ReLU, models/resnet.py:177-188) +
    MaxPool(3, 2, 1) [+ layer1's two BasicBlocks] + global average pool, 32 clips x 96 frames (49 152 rows per stem BatchNorm), in
    train mode (batch statistics, SyncBN code path) and in eval mode (running statistics).  Same kernels as the full network: direct
    C_in = 1 convolution, im2col + GEMM, col2im, tall BatchNorm forward / backward, overlapping max-pool, add + ReLU, average pool.
    HIP-vs-mirror <= 2e-2 on every parameter; the fp32 distance is printed."""
    from oracle import resnet as oresnet, rounding as R
    import torch.nn as nn
    variant = "resnet18"
    net, sd = load_net(variant, dev, 3, 11)
    keep = 1 if depth == "stem+layer1" else 0
    for i, name in enumerate(["layer1", "layer2", "layer3", "layer4"]):
        if i >= keep:
            setattr(net, name, nn.Sequential())
    layers = [2 if i < keep else 0 for i in range(4)]
    net.train(training)
    g = torch.Generator().manual_seed(33)
    if not training:                                       # non-trivial running statistics
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if k.endswith("running_mean"):
                    v.copy_(0.1 * torch.randn(v.shape, generator=g).to(dev)); sd[k] = v.detach().cpu().clone()
                elif k.endswith("running_var"):
                    v.copy_((0.5 + torch.rand(v.shape, generator=g)).to(dev)); sd[k] = v.detach().cpu().clone()
    x = torch.randn(32, 1, 64, 96, generator=g) * 1.3 + 0.2
    y = net(x.to(dev))
    w = torch.randn(y.shape, generator=g)
    (y * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    got = {n: p.grad.detach().float().cpu() for n, p in net.named_parameters()}

    def oracle_grads():
        p = {k: v.clone().requires_grad_("running" not in k and k in got) for k, v in sd.items()}
        yo = oresnet.forward(x, p, variant, training=training, layers=layers)
        (yo * w).sum().backward()
        return {k: v.grad for k, v in p.items() if v.grad is not None}
    fgrads = oracle_grads()
    with R.mirror_hip_bf16():
        mgrads = oracle_grads()
    rows = {n: (rel(got[n], mgrads[n]), rel(got[n], fgrads[n]), rel(mgrads[n], fgrads[n])) for n in got}
    for n in rows:
        print(f"   {n:26s} HIP-vs-mirror {rows[n][0]:.4f}   HIP-vs-fp32 {rows[n][1]:.4f}   (mirror-vs-fp32 sensitivity {rows[n][2]:.4f})")
    assert len(rows) == (9 if keep == 0 else 21)
    # measured: stem 0.5 % (train) / 0.2 % (eval), stem + layer1 0.6 % (eval) -- flat 2e-2; stem + layer1 in train mode 3.2 % at a
    # mirror-vs-fp32 sensitivity of 12 % (seven BatchNorms on batch statistics deep): 5e-2
    bound = 5e-2 if (training and keep) else 2e-2
    assert max(v[0] for v in rows.values()) <= bound, rows
    assert max(v[1] for v in rows.values()) <= 1.5e-1, rows


def test_resnet_train_gradients_are_chaotic_at_any_batch(dev):
    """Why the whole-network ResNet gradient checks are sensitivity-relative (tests/gradcheck.py) and not flat: at 32 clips (192 rows in
    the smallest BatchNorm, 49 152 in the stem) the oracle's own gradients still move by 30-50 % when it rounds where the HIP path
    stores bf16 -- the same as at the fixture's 4 clips.  Printed for the log; asserted: the fixture really is that sensitive (else a
    flat bound would be due), and HIP-vs-mirror stays inside the sensitivity-relative rule."""
    from oracle import resnet as oresnet, rounding as R
    variant = "resnet18"
    net, sd = load_net(variant, dev, 3, 11)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(32, 1, 64, 96, generator=g) * 1.3 + 0.2
    y = net(x.to(dev))
    w = torch.randn(y.shape, generator=g)
    (y * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    got = {n: p.grad.detach().float().cpu() for n, p in net.named_parameters()}

    def oracle_grads():
        p = {k: v.clone().requires_grad_("running" not in k) for k, v in sd.items()}
        yo = oresnet.forward(x, p, variant)
        (yo * w).sum().backward()
        return {k: v.grad for k, v in p.items() if v.grad is not None}
    fgrads = oracle_grads()
    with R.mirror_hip_bf16():
        mgrads = oracle_grads()
    sens = [rel(mgrads[n], fgrads[n]) for n in got]
    errs = [rel(got[n], mgrads[n]) for n in got]
    print(f"resnet18 train, B=32: bf16 sensitivity (mirror-vs-fp32) median {np.median(sens):.4f} max {max(sens):.4f}; HIP-vs-mirror median {np.median(errs):.4f} max {max(errs):.4f}")
    assert np.median(sens) > 0.1
    check_step_gradients("resnet18 B=32", got, mgrads, fgrads, 55)


def test_cfg1_resnet18_1s_b32_bt_step_vs_oracle(dev):
    """BASELINE config 1 (ResNet-18, 1 s synthetic clips -> 64 x 96 log-mel crops, batch 32, Barlow Twins; the reference's CPU-runnable
    case) as main.py:86-119 runs it, on the HIP path behind the reference's classes: embeddings by cosine, loss rel <= 3e-2 against the
    fp32 oracle on the same weights and views, every gradient against the bf16-mirror oracle, then one LARS step (the reference's
    optimiser for non-ViT encoders, utils/hyperparameters.py:107) leaves finite weights and a finite second loss."""
    from oracle import heads as oh, resnet as oresnet, rounding as R
    from ssl_audio_amd import hyperparameters as hp, loss as sloss, model, utils
    B, T = 32, 96
    cfg = hp.make_args(model_type="resnet18", batch_size=B, crop_frames=T)
    torch.manual_seed(0)
    net = utils.MultiCropWrapper(model.ModelWrapper(cfg), model.BarlowTwinsHead(cfg, 512)).to(dev)
    assert net.backbone.feature_dim == 512
    crit = sloss.BarlowTwinsLoss(cfg, ncrops=2).to(dev)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    base = torch.randn(B, 1, 64, T, generator=g)
    views = [base + 0.3 * torch.randn(B, 1, 64, T, generator=g), base + 0.3 * torch.randn(B, 1, 64, T, generator=g)]
    z1 = net([views[0].to(dev)], ncrops=1)                           # teacher pass: the first global crop (main.py:86-91)
    z2 = net([views[1].to(dev)], ncrops=1)                           # student pass: the other global crop, L = 0 (main.py:106-109)
    z = torch.cat([z1, z2])
    loss = crit(z2, z1, ngcrops_each=1)                              # forward_loss(teacher, student) (utils/loss.py:43)
    loss.backward()
    torch.cuda.synchronize()
    got = {k: p.grad.detach().float().cpu() for k, p in net.named_parameters() if p.grad is not None}

    def oracle(mirror):
        enc = {k[len("backbone.encoder."):]: v.clone().requires_grad_("running" not in k and v.dtype.is_floating_point)
               for k, v in sd0.items() if k.startswith("backbone.encoder.")}
        head = {k[len("head."):]: v.clone().requires_grad_("running" not in k and v.dtype.is_floating_point) for k, v in sd0.items() if k.startswith("head.")}
        ctx = R.mirror_hip_bf16() if mirror else R.mirror_hip_bf16(False)
        with ctx:
            a, _ = oh.head_forward(oresnet.forward(views[0], enc, "resnet18"), head, 1)      # two passes: BatchNorm statistics per pass
            b, _ = oh.head_forward(oresnet.forward(views[1], enc, "resnet18"), head, 1)
            zz = torch.cat([a, b])
            ls, _ = oh.bt_forward_loss(a, b, cfg.alpha, cfg.lmbda)
            ls.backward()
        grads = {"backbone.encoder." + k: v.grad for k, v in enc.items() if v.grad is not None}
        grads.update({"head." + k: v.grad for k, v in head.items() if v.grad is not None})
        return zz.detach(), float(ls.detach()), grads

    z_ref, ref_loss, fgrads = oracle(False)
    _, mloss, mgrads = oracle(True)
    cos = F.cosine_similarity(z.detach().double().cpu(), z_ref.double(), dim=1)
    print("cfg1 ResNet-18 1s B=32: loss", float(loss), "oracle", ref_loss, "mirror", mloss, "min cos", float(cos.min()))
    assert float(cos.min()) >= 0.995, float(cos.min())
    assert abs(float(loss) - ref_loss) / abs(ref_loss) <= 3e-2, (float(loss), ref_loss)
    assert set(got) == set(fgrads), set(got) ^ set(fgrads)
    check_step_gradients("cfg1 step", got, mgrads, fgrads, 60)
    # ---- one optimiser step with the reference's LARS, second forward stays finite and the loss moves
    opt = utils.LARS(utils.get_param_groups(net), lr=0.2, weight_decay=1e-6, weight_decay_filter=True, lars_adaptation_filter=True)
    opt.step()
    opt.zero_grad()
    with torch.no_grad():
        z1, z2 = net([views[0].to(dev)], ncrops=1), net([views[1].to(dev)], ncrops=1)
        loss2 = float(crit(z2, z1, ngcrops_each=1))
    assert np.isfinite(loss2) and loss2 != float(loss)
    assert all(torch.isfinite(p).all() for p in net.parameters())
