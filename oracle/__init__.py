"""CPU oracle for the Audio Barlow Twins pre-training hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (numpy + plain PyTorch CPU ops, fp32 unless a
function says fp64) of what jonahanton/SSL_audio computes on the path SURVEY.md §8 scopes.
Every function cites the reference file:line it follows.

Who may import it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
-- as the checker / the timed CPU baseline, never as the product.  `ssl_audio_amd/` must not
import it (tests/test_boundary.py greps for that) and fails loudly when the HIP library is missing.

Pinning status:
  * loss / head / predictor / augmentations / ViT + MAE encoder-decoder / full step: PINNED against
    golden vectors produced by running the reference itself in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz; checked in tests/test_oracle_golden.py).
  * log-mel frontend: PARITY UNPINNED.  The arithmetic lives in torchaudio.transforms.MelSpectrogram
    (un-vendored, version unpinned by the reference, absent from this image; call site
    datasets.py:39-48,115).  oracle/frontend.py restates torchaudio's documented defaults and is
    cross-checked against torch.stft only.
"""
