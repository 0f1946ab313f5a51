from ssl_audio_amd.augmentations import (MixupBYOLA, NormalizeBatch, RandomLinearFader, RandomResizeCrop,  # noqa: F401
                                         log_mixup_exp)
