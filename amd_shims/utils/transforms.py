"""AudioPairTransform from the MI355X package (utils/transforms.py:7-58); other names (the CIFAR transform) fall through to the
reference's utils/transforms.py when it is importable on this host."""
import importlib.util
import os
import sys

from ssl_audio_amd.transforms import AudioPairTransform  # noqa: F401


def _load_reference():
    here = os.path.dirname(os.path.abspath(__file__))
    for d in sys.path:
        cand = os.path.join(d, "utils", "transforms.py")
        if os.path.isfile(cand) and os.path.dirname(os.path.abspath(cand)) != here:
            try:
                spec = importlib.util.spec_from_file_location("_reference_utils_transforms", cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                return mod
            except Exception:          # e.g. torchvision absent: the audio path does not need it
                return None
    return None


_ref = _load_reference()
if _ref is not None:
    for _name in dir(_ref):
        if not _name.startswith("_") and _name not in globals():
            globals()[_name] = getattr(_ref, _name)
