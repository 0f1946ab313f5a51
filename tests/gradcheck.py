"""Shared gradient-comparison helpers of the -m gpu tests (TEST INFRASTRUCTURE).

Three references exist for a gradient of the HIP path:
  fp32      the oracle as pinned against the reference (plain fp32),
  mirror    the same oracle rounding to bf16 exactly where the HIP path stores bf16 (oracle/rounding.py),
  golden    gradients captured from the reference itself (fp32).
Where the problem is well conditioned (encoder with a linear loss) HIP-vs-mirror is <= 2e-2 and that bound is asserted
directly.  Whole training steps are NOT well conditioned at random init: the projector's BatchNorm removes the batch mean of
a representation that is almost all batch mean, and the Barlow-Twins gradient G_ii = 2(c_ii - 1) is a difference of nearly
equal numbers, so bf16 rounding alone moves step gradients by 5-30 % (measured: mirror-vs-fp32, the fixture's *sensitivity*).
`check_step_gradients` therefore bounds HIP-vs-mirror by a small multiple of that measured sensitivity, per parameter, and
demands agreement in direction: a sign error gives relative error 2 and cosine -1, a wrong scale factor of 2 relative error 1
-- both far outside 3 x sensitivity.
"""
import numpy as np
import torch


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


def check_step_gradients(tag, got, mirror, fp32, min_params, factor=3.0, floor=2e-2, min_cos=0.9, noise_sens=0.5):
    """got / mirror / fp32: dicts name -> gradient tensor.  For every parameter whose fp32 gradient is not itself rounding noise
    (sensitivity < noise_sens): rel(got, mirror) <= max(factor * sensitivity, floor); for weight matrices also cosine >= min_cos."""
    rows = []
    for k, gm in mirror.items():
        if k not in got or k not in fp32 or float(fp32[k].norm()) < 1e-12:
            continue
        sens = rel(gm, fp32[k])
        if sens >= noise_sens:                       # e.g. norm.bias ahead of a bias-free projector + BatchNorm: true gradient ~ 0
            continue
        rows.append((k, rel(got[k], gm), sens, cosine(got[k], gm), gm.dim() >= 2))
    # cosine floor: two evaluations that each carry independent relative noise s have cosine ~ 1 / (1 + s^2); the flat min_cos applies
    # where the fixture is well conditioned (s < 0.23), the noise-aware floor where bf16 itself moves the gradient by more than that
    bad = [(k, round(e, 4), round(s, 4), round(c, 4)) for k, e, s, c, mat in rows
           if e > max(factor * s, floor) or (mat and c < min(min_cos, 1.0 / (1.0 + 2.0 * s * s)))]
    errs = [e for _, e, _, _, _ in rows]
    senss = [s for _, _, s, _, _ in rows]
    mats = [c for _, _, _, c, mat in rows if mat]
    print(f"{tag}: {len(rows)} gradients; HIP-vs-mirror median {np.median(errs):.4f} max {max(errs):.4f}; "
          f"fixture bf16 sensitivity (mirror-vs-fp32) median {np.median(senss):.4f} max {max(senss):.4f}; weight-matrix cosine min {min(mats):.4f}")
    assert len(rows) >= min_params, (tag, len(rows))
    assert not bad, (tag, "name, HIP-vs-mirror, sensitivity, cosine", bad[:6])
    return rows
