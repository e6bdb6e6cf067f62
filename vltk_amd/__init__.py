"""vltk_amd -- MI355X-native Faster R-CNN visual-feature extraction for vltk.

Only what the hot path needs: the config object, seeded synthetic weights, the
`FRCNN` host mirror over the HIP C ABI (`libvltk_hip.so`), the
`VisnExtraction`-compatible adapter and the image-sharding helper.
Importing the package does not need a GPU; constructing `FRCNN` does.
"""
from .config import Config, vg_c4_config, vg_c4_config_dict  # noqa: F401
from .weights import make_state_dict, synthetic_images  # noqa: F401


def __getattr__(name):
    if name == "FRCNN":
        from .frcnn import FRCNN
        return FRCNN
    if name == "LxmertEncoder":          # N3: the encoder that consumes the extractor's output
        from .lxmert import LxmertEncoder
        return LxmertEncoder
    raise AttributeError(name)
