"""Configuration contract of the hot path: the flag NAMES and DEFAULTS main_bt_byol.py's modules read from `cfg`
(reference: utils/hyperparameters.py:27-110).  Table-driven; `vit_large` is added for BASELINE config 5
(not in the reference)."""
import argparse

MODELS = ['resnet50', 'resnet50_ReGP_NRF', 'resnet18', 'resnet18_ReGP_NRF', 'audiontt',
          'vit_base', 'vit_small', 'vit_tiny', 'vit_large', 'vitc_base', 'vitc_small', 'vitc_tiny']
DATASETS = ['fsd50k', 'audioset', 'librispeech', 'fsd50k+librispeech', 'audioset+librispeech', 'cifar10']
OPTIMIZERS = ['Adam', 'AdamW', 'SGD', 'LARS']

# (name, type, default)            -- plain valued flags
_VALUED = [
    ("model_type", str, "audiontt"), ("dataset", str, "fsd50k"), ("epochs", int, 100), ("epoch_save_f", int, 5),
    ("epoch_eval_f", int, 5), ("batch_size", int, 128), ("lmbda", float, 0.005), ("alpha", float, 1),
    ("projector_out_dim", int, 256), ("projector_n_hidden_layers", int, 1), ("projector_hidden_dim", int, 8192),
    ("local_crops_number", int, 0), ("unit_sec", float, 0.95), ("crop_frames", int, 96), ("sample_rate", int, 16000),
    ("n_fft", int, 1024), ("win_length", int, 1024), ("hop_length", int, 160), ("n_mels", int, 64), ("f_min", int, 60),
    ("f_max", int, 7800), ("num_workers", int, 20), ("mixup_ratio", float, 0.2), ("name", str, ""),
    ("mask_ratio", float, 0), ("mask_beta", float, 0.3), ("save_base_dir", str, ""), ("resume_path", str, None),
    ("optimizer", str, None), ("lr", float, None), ("lr_weights", float, None), ("lr_biases", float, None), ("wd", float, None),
]
# (name, type, default)            -- list valued flags (nargs='+')
_LISTS = [("local_crops_size", int, [16, 16]), ("virtual_crop_scale", float, [1, 1.5]), ("patch_size", int, [16, 16])]
# name -> default                  -- store_true switches
_SWITCHES = {
    "lr_schedule": False, "no_eval": False, "HSIC": False, "Gnoise": False, "pre_norm": False, "post_norm": False,
    "distributed": False, "use_fp16": False, "use_fp16_eval": False, "squeeze_excitation": False, "mask": False,
    "random_mask_ratio": False, "mask_ratio_schedule": False, "use_learned_pos_embd": False, "use_cls": True,
    "use_mean_pool": False, "masked_recon": False, "stop_gradient": False, "predictor": False,
}
# positive flag (default True) -> its negative twin
_PAIRED = {"mixup": "no_mixup", "RRC": "no_RRC", "RLF": "no_RLF", "load_lms": "load_wav"}
_CHOICES = {"model_type": MODELS, "dataset": DATASETS}


def get_std_parameters():
    parser = argparse.ArgumentParser(add_help=False)
    for name, typ, default in _VALUED:
        parser.add_argument("--" + name, type=typ, default=default, choices=_CHOICES.get(name))
    for name, typ, default in _LISTS:
        parser.add_argument("--" + name, type=typ, nargs="+", default=default)
    for name, default in _SWITCHES.items():
        parser.add_argument("--" + name, action="store_true", default=default)
    for pos, neg in _PAIRED.items():
        parser.add_argument("--" + pos, action="store_true", default=True)
        parser.add_argument("--" + neg, action="store_false", dest=pos)
    return parser


def get_hyperparameters():
    return [get_std_parameters()]


def setup_hyperparameters(args):
    """Model-dependent optimiser defaults (utils/hyperparameters.py:101-110): ViT -> AdamW, lr 1e-4*B/128, wd 0.06."""
    vit = 'vit' in args.model_type
    defaults = (dict(optimizer='AdamW', lr=1e-4 * args.batch_size / 128, wd=0.06) if vit else
                dict(optimizer='LARS', lr_weights=0.4 * args.batch_size / 128, lr_biases=0.0048 * args.batch_size / 128, wd=1e-5))
    for k, v in defaults.items():
        if getattr(args, k) is None:
            setattr(args, k, v)


def make_args(**overrides):
    """Namespace with the reference's defaults, for programmatic use (bench, tests)."""
    args = get_std_parameters().parse_args([])
    for k, v in overrides.items():
        setattr(args, k, v)
    setup_hyperparameters(args)
    return args
