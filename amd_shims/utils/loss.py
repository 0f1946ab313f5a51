from ssl_audio_amd.loss import BarlowTwinsLoss  # noqa: F401  (utils/loss.py:8-48)
