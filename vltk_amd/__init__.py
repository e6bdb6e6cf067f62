"""vltk_amd -- MI355X-native Faster R-CNN visual-feature extraction for vltk.

Only what the hot path needs: the config object, seeded synthetic weights, the
`FRCNN` host mirror over the HIP C ABI (`libvltk_hip.so`), the
`VisnExtraction`-compatible adapter and the image-sharding helper.
Importing the package does not need a GPU; constructing `FRCNN` does.
"""
import os as _os

# The forward overlaps half-batches on a second HIP stream.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4); once RCCL's streams exist the second stream shares the main stream's queue and the overlap is lost
# (measured: 374 instead of 402 images/s).  Takes effect only if HIP has not been initialised yet; an explicit value wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .config import Config, fpn_config, fpn_config_dict, vg_c4_config, vg_c4_config_dict  # noqa: F401
from .weights import make_state_dict, synthetic_images  # noqa: F401


def __getattr__(name):
    if name == "FRCNN":
        from .frcnn import FRCNN
        return FRCNN
    if name == "LxmertEncoder":          # N3: the encoder that consumes the extractor's output
        from .lxmert import LxmertEncoder
        return LxmertEncoder
    raise AttributeError(name)
