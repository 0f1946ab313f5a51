"""HEAR-benchmark API of the reference (hear/sample/vit.py, hear/utils.py) on the MI355X path: `from ssl_audio_amd.hear import vit`."""
from . import utils, vit  # noqa: F401
