"""AudioNTT2022 (BYOL-A v2 encoder, model.py:130-191) on the MI355X kernels -- the reference's default `--model_type audiontt`.

The modules own the parameters with the reference's names (`features.{0,1,4,5}.*`, `fc.{0,3}.*`); compute is `AudioNTTFn`:

  x [B,1,64,T] -conv3x3 (C_in = 1, bias)-> BN2d + ReLU -> MaxPool2 -> [im2col -> bf16 MFMA GEMM (+ bias)] -> BN2d + ReLU -> MaxPool2
    -> frames [B*T/4, 64*16] -> Linear + ReLU + Dropout(0.3) -> Linear + ReLU -> stack [conv frames | MLP] -> max_t + mean_t -> [B, d]

With `squeeze_excitation=True` an SE gate (model.py:196-213; se_block.hip) follows each MaxPool, as in the reference.
Channel-last feature maps (conv.hip), BatchNorm statistics exchanged across data-parallel ranks like every other BatchNorm of the
path.  Dropout draws its keep mask with torch's RNG (index bookkeeping, like the MAE masking noise) and applies it inside the
ReLU kernel; `keep=` makes it explicit for parity tests.
"""
import torch
import torch.nn as nn

from . import dist as sdist
from . import ops
from .convstem import _bn_forward, _kpad, _pack_conv_weight, bn_bwd_sums
from .engine import BF16_WEIGHTS, _wgrad, grad_target

BF16 = torch.bfloat16


class SE_Block(nn.Module):
    """Parameter holder of the squeeze-and-excitation gate (model.py:196-213): `excitation.0.weight` [c // r, c], `excitation.2.weight`
    [c, c // r], no biases.  Compute is ops.se_fwd / ops.se_bwd inside AudioNTTFn."""

    def __init__(self, c, r=16):
        super().__init__()
        self.squeeze = nn.AdaptiveAvgPool2d(1)
        self.excitation = nn.Sequential(nn.Linear(c, c // r, bias=False), nn.ReLU(inplace=True), nn.Linear(c // r, c, bias=False), nn.Sigmoid())


class AudioNTT2022Encoder(nn.Module):
    def __init__(self, n_mels=64, d=3072, base_d=64, mlp_hidden_d=2048, conv_layers=2, stack=True, squeeze_excitation=False):
        super().__init__()
        if conv_layers != 2 or not stack or base_d % 8 != 0:
            raise NotImplementedError("AudioNTT2022Encoder: the reference configuration (2 conv blocks, stack=True) only")
        if squeeze_excitation and (256 % base_d != 0 or base_d < 16):
            raise NotImplementedError("SE_Block on the MI355X path needs a channel count that divides 256")
        self.squeeze_excitation = squeeze_excitation
        convs = [nn.Conv2d(1, base_d, 3, stride=1, padding=1), nn.BatchNorm2d(base_d), nn.ReLU(), nn.MaxPool2d(2, stride=2)]
        if squeeze_excitation:
            convs.append(SE_Block(c=base_d))
        for _ in range(1, conv_layers):
            convs.extend([nn.Conv2d(base_d, base_d, 3, stride=1, padding=1), nn.BatchNorm2d(base_d), nn.ReLU(), nn.MaxPool2d(2, stride=2)])
            if squeeze_excitation:
                convs.append(SE_Block(c=base_d))
        self.features = nn.Sequential(*convs)
        self.conv_d = base_d * (n_mels // (2 ** conv_layers))
        self.fc = nn.Sequential(nn.Linear(self.conv_d, mlp_hidden_d), nn.ReLU(), nn.Dropout(p=0.3), nn.Linear(mlp_hidden_d, d - self.conv_d), nn.ReLU())
        self.stack = stack

    def engine_params(self):
        """Conv / BatchNorm / MLP parameters in AudioNTTFn's order, then the two SE gates' (W1, W2) when present.  With the gates the
        second conv block sits one index later in `features` (model.py:141-151), as in the reference's state dict."""
        f, m = self.features, self.fc
        k = 5 if self.squeeze_excitation else 4
        base = [f[0].weight, f[0].bias, f[1].weight, f[1].bias, f[k].weight, f[k].bias, f[k + 1].weight, f[k + 1].bias,
                m[0].weight, m[0].bias, m[3].weight, m[3].bias]
        if self.squeeze_excitation:
            base += [f[4].excitation[0].weight, f[4].excitation[2].weight, f[9].excitation[0].weight, f[9].excitation[2].weight]
        return base


class AudioNTT2022(AudioNTT2022Encoder):
    def __init__(self, n_mels=64, d=3072, mlp_hidden_d=2048, squeeze_excitation=False):
        super().__init__(n_mels=n_mels, d=d, mlp_hidden_d=mlp_hidden_d, squeeze_excitation=squeeze_excitation)
        self.embed_dim = d

    def forward(self, x, keep=None):
        """x [B,1,n_mels,T] -> [B, d].  keep: optional Dropout keep mask [B*T', mlp_hidden] (uint8) for parity tests."""
        if self.training and keep is None:
            Tq = (x.shape[-1] // 2) // 2
            keep = (torch.rand(x.shape[0] * Tq, self.fc[0].weight.shape[0], device=x.device) >= self.fc[2].p).to(torch.uint8)
        return AudioNTTFn.apply(x, self, keep, *self.engine_params())


class AudioNTTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, keep, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, w4, b4, *se):
        B, _, H, W = x.shape
        dev = x.device
        x = x.contiguous()
        C = w1.shape[0]
        train = mod.training
        bn1, bn2 = mod.features[1], mod.features[6 if se else 5]

        def se_gate(p, L, wa, wb):
            """SE_Block.forward (model.py:207-211) on the pooled map p [B * L, C]: returns the gated map and what the backward needs."""
            R = wa.shape[0]
            s_, h_, e_ = torch.empty(B, C, device=dev), torch.empty(B, R, device=dev), torch.empty(B, C, device=dev)
            y = torch.empty_like(p)
            ops.se_fwd(p, B, L, C, wa.detach().contiguous(), wb.detach().contiguous(), s_, h_, e_, y)
            return y, (p, s_, h_, e_)

        def bn_relu(h, bn, gamma, beta):
            if train:
                mean, rstd = _bn_forward(h, h.shape[0], bn)
            else:                                              # eval: running statistics (the HEAR / linear-probe path)
                mean, rstd = bn.running_mean.detach(), torch.rsqrt(bn.running_var.detach() + bn.eps)
            a = torch.empty(h.shape, dtype=BF16, device=dev)
            ops.bn_apply(h, mean, rstd, gamma.detach(), beta.detach(), True, y_bf16=a)
            return a, mean, rstd

        # block 1: direct conv (C_in = 1) -> BN + ReLU -> pool
        h1 = torch.empty(B * H * W, C, device=dev)
        ops.conv3x3_c1_fwd(x, w1.detach().reshape(C, 9).contiguous(), b1.detach(), (1, 1), h1)
        a1, m1, r1 = bn_relu(h1, bn1, g1, be1)
        H2, W2 = H // 2, W // 2
        p1 = torch.empty(B * H2 * W2, C, dtype=BF16, device=dev); i1 = torch.empty(B * H2 * W2, C, dtype=torch.uint8, device=dev)
        ops.maxpool2_fwd(a1, B, H, W, C, p1, i1)
        del a1
        se_saved = [None, None]
        if se:
            p1, se_saved[0] = se_gate(p1, H2 * W2, se[0], se[1])
        # block 2: im2col GEMM -> BN + ReLU -> pool
        kp = _kpad(C)
        P = torch.empty(B * H2 * W2, kp, dtype=BF16, device=dev)
        ops.im2col3x3(p1, B, H2, W2, C, (1, 1), P)
        h2 = torch.empty(B * H2 * W2, C, device=dev)
        ops.gemm(P, _pack_conv_weight(w2, kp), bias=b2.detach(), out_f32=h2)
        del P
        a2, m2, r2 = bn_relu(h2, bn2, g2, be2)
        H4, W4 = H2 // 2, W2 // 2
        p2 = torch.empty(B * H4 * W4, C, dtype=BF16, device=dev); i2 = torch.empty(B * H4 * W4, C, dtype=torch.uint8, device=dev)
        ops.maxpool2_fwd(a2, B, H2, W2, C, p2, i2)
        del a2
        if se:
            p2, se_saved[1] = se_gate(p2, H4 * W4, se[2], se[3])
        # frames + MLP, written straight into the stacked [B*T', conv_d + (d - conv_d)] output
        conv_d, hid, dfc = H4 * C, w3.shape[0], w4.shape[0]
        M = B * W4
        stack = torch.empty(M, conv_d + dfc, device=dev)
        X = torch.empty(M, conv_d, dtype=BF16, device=dev)
        ops.nhwc_to_frames(p2, B, H4, W4, C, frames_bf16=X, frames_f32=stack[:, :conv_d])
        h3 = torch.empty(M, hid, device=dev)
        ops.gemm(X, BF16_WEIGHTS.get(w3), bias=b3.detach(), out_f32=h3)
        a3 = torch.empty(M, hid, dtype=BF16, device=dev)
        scale = 1.0 / (1.0 - mod.fc[2].p) if keep is not None else 1.0
        ops.relu_mask_fwd(h3, keep, scale, y_bf16=a3)
        h4 = torch.empty(M, dfc, device=dev)
        ops.gemm(a3, BF16_WEIGHTS.get(w4), bias=b4.detach(), out_f32=h4)
        ops.relu_mask_fwd(h4, None, 1.0, y_f32=stack[:, conv_d:])
        out = torch.empty(B, conv_d + dfc, device=dev)
        arg = torch.empty(B, conv_d + dfc, dtype=torch.int32, device=dev)
        ops.meanmax_time_fwd(stack.view(B, W4, conv_d + dfc), out, arg)
        ctx.saved = (x, h1, m1, r1, p1, i1, h2, m2, r2, i2, X, h3, a3, h4, arg, keep, scale)
        ctx.dims = (B, H, W, C, conv_d, hid, dfc)
        ctx.params = (w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, w4, b4)
        ctx.se, ctx.se_saved = se, se_saved
        ctx.train = train
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h1, m1, r1, p1, i1, h2, m2, r2, i2, X, h3, a3, h4, arg, keep, scale = ctx.saved
        B, H, W, C, conv_d, hid, dfc = ctx.dims
        w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, w4, b4 = ctx.params
        dev = x.device
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        M = B * W4
        Wn = sdist.get_world_size()
        dstack = torch.empty(B, W4, conv_d + dfc, device=dev)
        ops.meanmax_time_bwd(dout.contiguous().float(), arg, dstack)
        dstack = dstack.view(M, conv_d + dfc)
        # fc2 (+ ReLU)
        g4 = torch.empty(M, dfc, dtype=BF16, device=dev)
        ops.relu_mask_bwd(dstack[:, conv_d:], h4, None, 1.0, g4)
        dw4b, dw4 = grad_target(w4); db4b, db4 = grad_target(b4)
        _wgrad(g4, a3, dw4b)
        ops.colsum_bf16(g4, db4b, accumulate=True)
        da3 = torch.empty(M, hid, device=dev)
        ops.gemm(g4, BF16_WEIGHTS.get(w4), b_kmajor=False, out_f32=da3)
        # dropout + ReLU + fc1
        g3 = torch.empty(M, hid, dtype=BF16, device=dev)
        ops.relu_mask_bwd(da3, h3, keep, scale, g3)
        dw3b, dw3 = grad_target(w3); db3b, db3 = grad_target(b3)
        _wgrad(g3, X, dw3b)
        ops.colsum_bf16(g3, db3b, accumulate=True)
        dX = torch.empty(M, conv_d, device=dev)
        ops.gemm(g3, BF16_WEIGHTS.get(w3), b_kmajor=False, out_f32=dX)
        # frames (MLP input + the stacked copy) -> pooled map -> pool / BN / conv 2
        dp2 = torch.empty(B * H4 * W4, C, device=dev)
        ops.frames_to_nhwc(dX, dstack[:, :conv_d], B, H4, W4, C, dp2)
        se, dse = ctx.se, [None] * 4

        def se_gate_bwd(dy, saved, wa, wb, L, slot):
            """dy fp32 [B * L, C] at the gate's output -> gradient at its input; W1 / W2 gradients accumulated into their targets."""
            p_in, s_, h_, e_ = saved
            dx_ = torch.empty_like(dy)
            dab, dse[slot] = grad_target(wa); dbb, dse[slot + 1] = grad_target(wb)
            ops.se_bwd(dy, p_in, B, L, C, wa.detach().contiguous(), wb.detach().contiguous(), s_, h_, e_, dx_, dab, dbb)
            return dx_

        if se:
            dp2 = se_gate_bwd(dp2, ctx.se_saved[1], se[2], se[3], H4 * W4, 2)
        da2 = torch.empty(B * H2 * W2, C, device=dev)
        ops.maxpool2_bwd(dp2, i2, B, H2, W2, C, da2)

        def bn_bwd(da, h, mean, rstd, gamma, beta, rows):
            s = torch.empty(2, C, device=dev)
            ops.bn_bwd_stats_tall(da, h, mean, rstd, gamma, beta, True, s[0], s[1])
            dgb, dg = grad_target(gamma); dbb, dbt = grad_target(beta)
            ops.axpy(dbb, s[0]); ops.axpy(dgb, s[1])
            s = bn_bwd_sums(s, ctx.train)
            dh = torch.empty(rows, C, dtype=BF16, device=dev)
            ops.bn_bwd_apply(da, h, mean, rstd, gamma, beta, True, s[0], s[1], 1.0 / (rows * Wn), dx_bf16=dh)
            return dh, dg, dbt

        dh2, dg2, dbe2 = bn_bwd(da2, h2, m2, r2, g2, be2, B * H2 * W2)
        kp = _kpad(C)
        P = torch.empty(B * H2 * W2, kp, dtype=BF16, device=dev)
        ops.im2col3x3(p1, B, H2, W2, C, (1, 1), P)
        dwp = torch.zeros(C, kp, device=dev)
        _wgrad(dh2, P, dwp)
        del P
        dw2b, dw2 = grad_target(w2); db2b, db2 = grad_target(b2)
        ops.axpy(dw2b.view(-1), dwp[:, :9 * C].reshape(C, 3, 3, C).permute(0, 3, 1, 2).contiguous().view(-1))
        ops.colsum_bf16(dh2, db2b, accumulate=True)
        dP = torch.empty(B * H2 * W2, kp, dtype=BF16, device=dev)
        ops.gemm(dh2, _pack_conv_weight(w2, kp), b_kmajor=False, out_bf16=dP)
        dp1 = torch.empty(B * H2 * W2, C, device=dev)
        ops.col2im3x3(dP, B, H2, W2, C, (1, 1), dp1)
        del dP
        if se:
            dp1 = se_gate_bwd(dp1, ctx.se_saved[0], se[0], se[1], H2 * W2, 0)
        da1 = torch.empty(B * H * W, C, device=dev)
        ops.maxpool2_bwd(dp1, i1, B, H, W, C, da1)
        dh1, dg1, dbe1 = bn_bwd(da1, h1, m1, r1, g1, be1, B * H * W)
        dw1b, dw1 = grad_target(w1); db1b, db1 = grad_target(b1)
        ops.conv3x3_c1_wgrad(x, dh1, (1, 1), dw1b.view(C, 9), db1b)
        return (None, None, None, dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2, dw3, db3, dw4, db4) + (tuple(dse) if se else ())
