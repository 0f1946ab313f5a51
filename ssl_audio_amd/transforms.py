"""AudioPairTransform with the reference's signature (utils/transforms.py:7-58): per-sample, returns the list
[global view 1, global view 2, local crops...].  The module instance is applied twice per clip exactly like the
reference (shared mixup bank, shared RNG streams).  For throughput use augmentations.BatchedPairAugment."""
import torch.nn as nn

from . import augmentations


class AudioPairTransform(nn.Module):
    def __init__(self, args, train_transform=True, multi_transform=True,
                 mixup_ratio=0.2, gauss_noise_ratio=0.2,
                 global_crop_scale=(0.6, 1.5), local_crop_scale=(0.05, 0.6)):
        super().__init__()
        self.multi_transform = multi_transform
        self.local_crops_number = args.local_crops_number
        if train_transform is True:
            global_transforms = []
            if args.mixup:
                global_transforms.append(augmentations.MixupBYOLA(ratio=mixup_ratio))
            if args.Gnoise:
                global_transforms.append(augmentations.MixGaussianNoise(ratio=gauss_noise_ratio))
            if args.RRC:
                out_size = (args.n_mels, args.crop_frames)
                global_transforms.append(augmentations.RandomResizeCrop(
                    out_size, virtual_crop_scale=tuple(args.virtual_crop_scale),
                    freq_scale=global_crop_scale, time_scale=global_crop_scale))
            if args.RLF:
                global_transforms.append(augmentations.RandomLinearFader())
            self.global_transform = nn.Sequential(*global_transforms)
        else:
            self.global_transform = nn.Identity()
        self.local_transform = nn.Sequential(augmentations.RandomResizeCrop(
            tuple(args.local_crops_size), virtual_crop_scale=(1, 1), freq_scale=local_crop_scale, time_scale=local_crop_scale))

    def forward(self, x):
        if self.multi_transform:
            crops = [self.global_transform(x), self.global_transform(x)]
            for _ in range(self.local_crops_number):
                crops.append(self.local_transform(x))
            return crops
        return self.global_transform(x)
