"""Hot-path names from the MI355X package; everything else falls through to the reference's utils/utils.py."""
import importlib.util
import os
import sys

from ssl_audio_amd.utils import (EMA, LARS, MultiCropWrapper, adjust_learning_rate, cosine_scheduler, encode_vit, get_param_groups,  # noqa: F401
                                 get_rank, get_world_size, init_distributed_mode, is_dist_avail_and_initialized, is_main_process,
                                 model_setup_ddp, off_diagonal, save_on_master, sine_scheduler_increase, update_moving_average)


def _load_reference_utils():
    here = os.path.dirname(os.path.abspath(__file__))
    for d in sys.path:
        cand = os.path.join(d, "utils", "utils.py")
        if os.path.isfile(cand) and os.path.dirname(os.path.abspath(cand)) != here:
            spec = importlib.util.spec_from_file_location("_reference_utils_utils", cand)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
    return None


_ref = _load_reference_utils()
if _ref is not None:
    for _name in dir(_ref):
        if not _name.startswith("_") and _name not in globals():
            globals()[_name] = getattr(_ref, _name)
