"""ResNet-18 encoders of the reference (models/resnet.py: `resnet18`, `resnet18_ReGP_NRF`; BASELINE config 1) on the MI355X kernels.

`ResNet` owns the parameters under the reference's names (`conv1.{0,3,6}.weight`, `conv1.{1,4,7}.*`, `layer<i>.<b>.conv{1,2}.weight`,
`layer<i>.<b>.bn{1,2}.*`, `layer<i>.<b>.downsample.{0,1}.*`; `fc` is the Identity model.py:74-81 installs), so checkpoints carry over.
Compute is one autograd Function with a hand-written schedule, everything channel-last (a feature map IS the [B*H*W, C] matrix):

  x [B,1,F,T] --3x3 conv (C_in = 1, direct kernel)--> BN+ReLU --[im2col -> bf16 MFMA GEMM -> BN+ReLU] x2--> MaxPool(3, 2, 1)
     --> 8 BasicBlocks: [im2col -> GEMM -> BN+ReLU -> im2col -> GEMM -> BN] + identity (or subsample -> 1x1 GEMM -> BN) -> ReLU
     --> global average pool (resnet18) or max + mean over time of the (mel x channel) stack (ReGP, resnet18_ReGP_NRF)

Only BasicBlock networks are built: the Bottleneck variants (`resnet50*`) raise NotImplementedError in `ModelWrapper`.
BatchNorm2d statistics are exchanged across data-parallel ranks like the projector's (SyncBN, utils/utils.py:411).
"""
import torch
import torch.nn as nn

from . import dist as sdist
from . import ops
from .convstem import BN_EPS, BN_MOMENTUM, _bn_forward, _kpad, _pack_conv_weight, bn_bwd_sums
from .engine import BF16_WEIGHTS, _wgrad, grad_target

BF16 = torch.bfloat16


def _pair(s):
    return tuple(s) if isinstance(s, (tuple, list)) else (s, s)


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=_pair(stride), padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=_pair(stride), bias=False)


class BasicBlock(nn.Module):
    """Parameter holder, models/resnet.py:33-80 (forward lives in ResNetFn)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = _pair(stride)


class ResNet(nn.Module):
    """models/resnet.py:141-274 for BasicBlock, the ResNet-C stem (C=True) and D=False -- what the reference's factories build."""

    def __init__(self, layers, strides, ReGP=False):
        super().__init__()
        self.ReGP = ReGP
        self.inplanes = 64
        self.stem_stride = _pair(strides[0])
        self.conv1 = nn.Sequential(
            nn.Conv2d(1, 32, kernel_size=3, stride=self.stem_stride, padding=1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, 32, kernel_size=3, stride=1, padding=1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, 64, kernel_size=3, stride=1, padding=1, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], strides[1])
        self.layer2 = self._make_layer(128, layers[1], strides[2])
        self.layer3 = self._make_layer(256, layers[2], strides[3])
        self.layer4 = self._make_layer(512, layers[3], strides[4])
        if not ReGP:
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Identity()                                   # model.py:74-81 replaces the classifier by Identity
        for m in self.modules():                                   # models/resnet.py:199-204
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if _pair(stride) != (1, 1) or self.inplanes != planes:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes, stride), nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(self.inplanes, planes))
        return nn.Sequential(*layers)

    def blocks(self):
        return [b for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer]

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("ssl_audio_amd.resnet: CUDA (ROCm) tensors only -- there is no CPU path")
        return ResNetFn.apply(x, self, *self.parameters())


def resnet18():
    """models/resnet.py:277-292."""
    return ResNet([2, 2, 2, 2], [2, 1, 2, 2, 2])


def resnet18_ReGP_NRF():
    """models/resnet.py:349-360: ReGP + narrow receptive field (no stem stride, time-only stride in the last stage)."""
    return ResNet([2, 2, 2, 2], [1, 1, 2, 2, [1, 2]], ReGP=True)


# ---------------------------------------------------------------------------------------------------------------------------------
class _ConvBN:
    """One convolution + BatchNorm2d of the forward, with what its backward needs."""
    __slots__ = ("conv", "bn", "k", "stride", "relu", "S", "H", "W", "Ci", "Ho", "Wo", "Co", "a_in", "x_img", "h", "mean", "rstd", "train")


def _conv_bn_fwd(conv, bn, a16, S, H, W, Ci, relu, x_img=None, training=True):
    """a16 bf16 [S*H*W, Ci] (or x_img fp32 [S,1,H,W] for the single-channel first layer) -> record with h = conv output (fp32) and the
    batch statistics.  3x3 (pad 1) through im2col + GEMM, 1x1 through row subsampling + GEMM."""
    r = _ConvBN()
    r.conv, r.bn, r.relu = conv, bn, relu
    r.k = conv.kernel_size[0]
    r.stride = _pair(conv.stride)
    r.S, r.H, r.W, r.Ci = S, H, W, Ci
    sh, sw = r.stride
    r.Co = conv.weight.shape[0]
    dev = conv.weight.device
    if r.k == 3:
        r.Ho, r.Wo = ops.conv_out_size(H, sh), ops.conv_out_size(W, sw)
    else:
        r.Ho, r.Wo = (H - 1) // sh + 1, (W - 1) // sw + 1
    M = S * r.Ho * r.Wo
    r.h = torch.empty(M, r.Co, device=dev)
    r.x_img, r.a_in = x_img, a16
    if x_img is not None:
        ops.conv3x3_c1_fwd(x_img, conv.weight.detach().reshape(r.Co, 9).contiguous(), None, r.stride, r.h)
    elif r.k == 3:
        kp = _kpad(Ci)
        P = torch.empty(M, kp, dtype=BF16, device=dev)
        ops.im2col3x3(a16, S, H, W, Ci, r.stride, P)
        ops.gemm(P, _pack_conv_weight(conv.weight, kp), out_f32=r.h)
        del P
    else:
        xs = a16
        if r.stride != (1, 1):
            xs = torch.empty(M, Ci, dtype=BF16, device=dev)
            ops.subsample_fwd(a16, S, H, W, Ci, r.stride, xs)
        r.a_in = xs                                              # the 1x1 convolution's GEMM operand (already subsampled)
        ops.gemm(xs, BF16_WEIGHTS.get(conv.weight), out_f32=r.h)
    r.train = training
    r.mean, r.rstd = _bn_forward(r.h, M, bn, training)
    return r


def _conv_bn_bwd(r, dz, grads, dx_out=None, need_dx=True):
    """dz fp32 [M, Co]: gradient at the BatchNorm output (after the ReLU when r.relu).  Accumulates the parameter gradients into `grads`
    (id -> autograd value) and returns the gradient at the convolution's input: fp32 [S*H*W, Ci] written by col2im for a 3x3, or, for a
    1x1, ADDED into dx_out at the sampled pixels."""
    conv, bn = r.conv, r.bn
    dev = dz.device
    M, Co, Ci = r.h.shape[0], r.Co, r.Ci
    W_ = sdist.get_world_size()
    gamma, beta = bn.weight, bn.bias
    s = torch.empty(2, Co, device=dev)
    ops.bn_bwd_stats_tall(dz, r.h, r.mean, r.rstd, gamma, beta, r.relu, s[0], s[1])
    dgb, grads[id(gamma)] = grad_target(gamma)
    dbb, grads[id(beta)] = grad_target(beta)
    ops.axpy(dbb, s[0])
    ops.axpy(dgb, s[1])
    s = bn_bwd_sums(s, r.train)
    cpad = (Co + 63) // 64 * 64 if r.x_img is None else Co            # zero columns up to the dgrad GEMM's K granule
    dh_full = torch.zeros(M, cpad, dtype=BF16, device=dev) if cpad != Co else torch.empty(M, Co, dtype=BF16, device=dev)
    dh = dh_full[:, :Co]
    ops.bn_bwd_apply(dz, r.h, r.mean, r.rstd, gamma, beta, r.relu, s[0], s[1], 1.0 / (M * W_), dx_bf16=dh)
    dwbuf, grads[id(conv.weight)] = grad_target(conv.weight)
    if r.x_img is not None:
        ops.conv3x3_c1_wgrad(r.x_img, dh, r.stride, dwbuf.view(Co, 9))
        return None
    if r.k == 1:
        _wgrad(dh, r.a_in, dwbuf.view(Co, Ci))
        if not need_dx:
            return None
        dxs = torch.empty(M, Ci, dtype=BF16, device=dev)
        w16 = BF16_WEIGHTS.get(conv.weight)
        if cpad != Co:
            wp = torch.zeros(cpad, Ci, dtype=BF16, device=dev)
            wp[:Co] = w16
            w16 = wp
        ops.gemm(dh_full, w16, b_kmajor=False, out_bf16=dxs)
        if r.stride == (1, 1):
            raise NotImplementedError("1x1 convolution without stride is not part of the BasicBlock networks")
        ops.subsample_bwd_add(dxs, r.S, r.H, r.W, Ci, r.stride, dx_out)
        return dx_out
    kp = _kpad(Ci)
    P = torch.empty(M, kp, dtype=BF16, device=dev)
    ops.im2col3x3(r.a_in, r.S, r.H, r.W, Ci, r.stride, P)            # recomputed: cheaper than keeping 9x the activations
    dwp = torch.zeros(Co, kp, device=dev)
    _wgrad(dh, P, dwp)
    del P
    ops.axpy(dwbuf.view(-1), dwp[:, :9 * Ci].reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2).contiguous().view(-1))
    if not need_dx:
        return None
    dP = torch.empty(M, kp, dtype=BF16, device=dev)
    ops.gemm(dh_full, _pack_conv_weight(conv.weight, kp, rows=cpad), b_kmajor=False, out_bf16=dP)
    da = torch.empty(r.S * r.H * r.W, Ci, device=dev)
    ops.col2im3x3(dP, r.S, r.H, r.W, Ci, r.stride, da)
    return da


def _bn_relu_out(r):
    """BN (+ ReLU) of a record -> bf16 activation for the next convolution."""
    a = torch.empty(r.h.shape[0], r.Co, dtype=BF16, device=r.h.device)
    ops.bn_apply(r.h, r.mean, r.rstd, r.bn.weight.detach(), r.bn.bias.detach(), r.relu, y_bf16=a)
    return a


class ResNetFn(torch.autograd.Function):
    """x [S,1,F,T] fp32 -> embedding [S, 512] (average pool) or [S, C * F'] (ReGP).  Gradients: every parameter (the input gets none)."""

    @staticmethod
    def forward(ctx, x, net, *params):
        S, _, F_, T_ = x.shape
        dev = x.device
        x = x.contiguous()
        tape = []
        # ---- ResNet-C stem (models/resnet.py:177-188)
        tr = net.training
        r = _conv_bn_fwd(net.conv1[0], net.conv1[1], None, S, F_, T_, 1, True, x_img=x, training=tr)
        a = _bn_relu_out(r)
        stem = [r]
        for l in (1, 2):
            r = _conv_bn_fwd(net.conv1[3 * l], net.conv1[3 * l + 1], a, S, stem[-1].Ho, stem[-1].Wo, stem[-1].Co, True, training=tr)
            a = _bn_relu_out(r)
            stem.append(r)
        H, W, C = stem[-1].Ho, stem[-1].Wo, stem[-1].Co
        # ---- MaxPool2d(3, 2, 1)
        Hp, Wp = ops.pool_out_size(H), ops.pool_out_size(W)
        p16 = torch.empty(S * Hp * Wp, C, dtype=BF16, device=dev)
        p32 = torch.empty(S * Hp * Wp, C, device=dev)
        pidx = torch.empty(S * Hp * Wp, C, dtype=torch.uint8, device=dev)
        ops.maxpool3s2_fwd(a, S, H, W, C, p16, p32, pidx)
        pool = (H, W, C, pidx)
        del a
        # ---- BasicBlocks (models/resnet.py:62-80)
        y16, y32, H, W = p16, p32, Hp, Wp
        for blk in net.blocks():
            r1 = _conv_bn_fwd(blk.conv1, blk.bn1, y16, S, H, W, C, True, training=tr)
            a1 = _bn_relu_out(r1)
            r2 = _conv_bn_fwd(blk.conv2, blk.bn2, a1, S, r1.Ho, r1.Wo, r1.Co, False, training=tr)
            z = torch.empty(r2.h.shape[0], r2.Co, device=dev)
            ops.bn_apply(r2.h, r2.mean, r2.rstd, blk.bn2.weight.detach(), blk.bn2.bias.detach(), False, y_f32=z)
            rd = None
            idn = y32
            if blk.downsample is not None:
                rd = _conv_bn_fwd(blk.downsample[0], blk.downsample[1], y16, S, H, W, C, False, training=tr)
                idn = torch.empty(rd.h.shape[0], rd.Co, device=dev)
                ops.bn_apply(rd.h, rd.mean, rd.rstd, blk.downsample[1].weight.detach(), blk.downsample[1].bias.detach(), False, y_f32=idn)
            out32 = torch.empty_like(z)
            out16 = torch.empty(z.shape, dtype=BF16, device=dev)
            ops.add_relu_fwd(z, idn, out32, out16)
            tape.append((r1, r2, rd, out32))
            y16, y32, H, W, C = out16, out32, r2.Ho, r2.Wo, r2.Co
        # ---- head
        if net.ReGP:                                             # models/resnet.py:260-265: (B, T, mel * ch), max + mean over time
            fr = torch.empty(S, W, H * C, device=dev)
            ops.nhwc_to_frames(y16, S, H, W, C, frames_f32=fr.view(S * W, H * C))
            out = torch.empty(S, H * C, device=dev)
            arg = torch.empty(S, H * C, dtype=torch.int32, device=dev)
            ops.meanmax_time_fwd(fr, out, arg)
            ctx.head = (arg,)
        else:
            out = torch.empty(S, C, device=dev)
            ops.avgpool_fwd(y32, S, H * W, C, out)
            ctx.head = ()
        ctx.net, ctx.stem, ctx.pool, ctx.tape, ctx.last = net, stem, pool, tape, (S, H, W, C)
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, dout):
        net, stem, tape = ctx.net, ctx.stem, ctx.tape
        S, H, W, C = ctx.last
        dev = dout.device
        dout = dout.contiguous()
        grads = {}
        # ---- head
        dy = torch.empty(S * H * W, C, device=dev)
        if net.ReGP:
            dfr = torch.empty(S, W, H * C, device=dev)
            ops.meanmax_time_bwd(dout, ctx.head[0], dfr)
            ops.frames_to_nhwc(dfr.view(S * W, H * C), None, S, H, W, C, dy)
        else:
            ops.avgpool_bwd(dout, S, H * W, C, dy)
        # ---- blocks, last to first
        for r1, r2, rd, out32 in reversed(tape):
            ds = torch.empty_like(out32)
            ops.relu_bwd(dy, None, out32, ds)                    # through the block's final ReLU
            da1 = _conv_bn_bwd(r2, ds, grads)                    # bn2 + conv2  -> gradient at relu(bn1(.))
            dx = _conv_bn_bwd(r1, da1, grads)                    # bn1 (+ReLU) + conv1 -> gradient at the block input (written by col2im)
            if rd is not None:
                _conv_bn_bwd(rd, ds, grads, dx_out=dx)           # identity path: BN + strided 1x1 convolution, added at the sampled pixels
            else:
                ops.axpy(dx.view(-1), ds.view(-1))
            dy = dx
        # ---- MaxPool backward, stem
        Hs, Ws, Cs, pidx = ctx.pool
        da = torch.empty(S * Hs * Ws, Cs, device=dev)
        ops.maxpool3s2_bwd(dy, pidx, S, Hs, Ws, Cs, da)
        for r in reversed(stem):
            da = _conv_bn_bwd(r, da, grads)
        return (None, None) + tuple(grads.get(id(p)) for p in ctx.params)
