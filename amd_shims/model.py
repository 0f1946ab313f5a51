from ssl_audio_amd.model import BarlowTwinsHead, BarlowTwinsPredictor, ModelWrapper, ViT  # noqa: F401  (model.py:11-127)
