from ssl_audio_amd.transforms import AudioPairTransform  # noqa: F401  (utils/transforms.py:7-58)
