from ssl_audio_amd.augmentations import (MixGaussianNoise, MixupBYOLA, NormalizeBatch, RandomLinearFader,  # noqa: F401
                                         RandomResizeCrop, RunningNorm, log_mixup_exp)
