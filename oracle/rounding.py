"""bf16 rounding hooks for the oracle (TEST INFRASTRUCTURE).

The HIP path keeps the residual stream, LayerNorm / BatchNorm / loss arithmetic and every accumulator in fp32, and
rounds to bf16 exactly where a tensor is stored as a GEMM / attention operand (ssl_audio_amd/engine.py,
functional.py).  With `mirror_hip_bf16()` active the oracle rounds at those same points -- forward values AND the
gradients the HIP backward stores as bf16 -- so that a gradient comparison against the HIP path is tight
(<= 2e-2) even on fixtures whose fp32 gradients are ill-conditioned.  With the hooks off (default) the oracle is the
plain fp32 restatement that tests/test_oracle_golden.py pins against the reference's own outputs; the mirror mode is
the SAME code with these roundings inserted, nothing else changes.

  q(x)    value stored as bf16 by the forward, its gradient stored as bf16 by the backward   (h1, qkv, ao, h2)
  qf(x)   value stored as bf16, gradient kept fp32                                          (projector input / activation)
  qb(x)   value fp32, gradient consumed as bf16                                             (outputs whose dY feeds dgrad/wgrad)
  qw(w)   bf16 weight copy, fp32 gradient accumulation
"""
import contextlib

import torch

_ACTIVE = False


def _r(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return _r(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (_r(g) if ctx.bwd else g), None, None


class _GeluPair(torch.autograd.Function):
    """fc1 epilogue + fc2-dgrad epilogue of the HIP path: a = bf16(gelu(pre)), g' = bf16(gelu'(pre)) saved by the forward;
    backward dpre = bf16(dy * g')."""

    @staticmethod
    def forward(ctx, pre):
        x = pre.double()
        cdf = 0.5 * (1.0 + torch.erf(x * 0.7071067811865476))
        pdf = torch.exp(-0.5 * x * x) * 0.3989422804014327
        ctx.save_for_backward(_r((cdf + x * pdf).to(pre.dtype)))
        return _r((x * cdf).to(pre.dtype))

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return _r(g * d)


def active():
    return _ACTIVE


@contextlib.contextmanager
def mirror_hip_bf16(on=True):
    global _ACTIVE
    old, _ACTIVE = _ACTIVE, on
    try:
        yield
    finally:
        _ACTIVE = old


def q(x):
    return _Round.apply(x, True, True) if _ACTIVE else x


def qf(x):
    return _Round.apply(x, True, False) if _ACTIVE else x


def qb(x):
    return _Round.apply(x, False, True) if _ACTIVE else x


def qw(w):
    return _Round.apply(w, True, False) if _ACTIVE else w


def gelu(pre):
    return _GeluPair.apply(pre) if _ACTIVE else torch.nn.functional.gelu(pre)
