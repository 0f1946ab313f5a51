from ssl_audio_amd.hyperparameters import *  # noqa: F401,F403  (utils/hyperparameters.py)
from ssl_audio_amd.hyperparameters import get_hyperparameters, get_std_parameters, setup_hyperparameters  # noqa: F401
