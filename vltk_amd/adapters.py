"""`VisnExtraction`-compatible FRCNN extractor plugin (the reference's plugin boundary).

Mirrors `vltk.adapters.frcnn.FRCNN` (reference vltk/adapters/frcnn.py:10-64) and the registry
lookup `Adapters().get("frcnn")` (vltk/adapters/__init__.py:53): same static methods
`setup() -> (model, model_config)`, `schema(max_detections, visual_dim)`, `forward(model, entry)`
with the same entry keys (vltk/vars.py: "image", "size", "wh_scale", ...) and the same return
layout (dict of length-1 lists, extended row-wise by the caller, abc/extraction.py:211-213).
`forward_batch` is the batched extension (the reference runs batch 1).

`setup()` upstream fetches "unc-nlp/frcnn-vg-finetuned" by name (network); here it takes a LOCAL
checkpoint directory (env VLTK_AMD_FRCNN_PATH or argument) or, with `synthetic=True`, the seeded
synthetic weights used by the tests and the benchmark.
"""
import os

import torch

from .config import Config, vg_c4_config

# string keys of vltk/vars.py:38-60 that the extraction loop uses
IMG, SIZE, SCALE, RAWSIZE, FILEPATH, IMGID, SPLIT = "image", "size", "wh_scale", "rawsize", "filepath", "imgid", "split"
FEATURES, BOX = "features", "box"


def rescale_box(boxes, wh_scale):
    """reference vltk/utils/adapters.py:205-216 (in place)."""
    h_scale, w_scale = wh_scale[1], wh_scale[0]
    boxes[:, 0] *= w_scale
    boxes[:, 1] *= h_scale
    boxes[:, 2] *= w_scale
    boxes[:, 3] *= h_scale
    return boxes


def _features():
    """Arrow feature constructors of vltk/features.py:81-95 (`datasets` objects when importable)."""
    try:
        import datasets as ds
        ids = lambda: ds.Sequence(length=-1, feature=ds.Value("float32"))                      # noqa: E731
        box = lambda: ds.Sequence(length=-1, feature=ds.Sequence(length=-1, feature=ds.Value("float32")))   # noqa: E731
        f3d = lambda n, d: ds.Array2D((n, d), dtype="float32")                                 # noqa: E731
    except Exception:                                                                          # pragma: no cover
        ids = lambda: {"type": "sequence", "dtype": "float32"}                                 # noqa: E731
        box = lambda: {"type": "sequence2d", "dtype": "float32"}                               # noqa: E731
        f3d = lambda n, d: {"type": "array2d", "shape": (n, d), "dtype": "float32"}            # noqa: E731
    return ids, box, f3d


class FRCNN:
    """Registry key = class name lower-cased = "frcnn"."""

    _is_feature = True
    _batch_size = 128           # Arrow write batch of the caller (abc/extraction.py:26)

    # vltk/adapters/frcnn.py:13-23 (VisionConfig kwargs; kept as a plain description)
    default_processor = {
        "transforms": ["FromFile", "Resize", "ToTensor", "Normalize"],
        "size": 800, "max_size": 1333, "mode": "bilinear", "pad_value": 0.0,
        "mean": [102.9801 / 255, 115.9465 / 255, 122.7717 / 255], "std": [1.0, 1.0, 1.0],
    }

    @staticmethod
    def setup(path=None, synthetic=False, precision=None, seed=1234):
        from .frcnn import FRCNN as FasterRCNN
        path = path or os.environ.get("VLTK_AMD_FRCNN_PATH")
        if synthetic or path is None:
            if not synthetic:
                raise EnvironmentError(
                    "Can't load weights for 'unc-nlp/frcnn-vg-finetuned': fetch-by-name needs the network. "
                    "Point VLTK_AMD_FRCNN_PATH at a local directory with pytorch_model.bin + config.yaml, "
                    "or call setup(synthetic=True).")
            from .weights import make_state_dict
            model_config = vg_c4_config()
            model = FasterRCNN(model_config, precision=precision).load_state_dict(make_state_dict(model_config, seed))
            return model.eval(), model_config
        model_config = Config.from_pretrained(path)
        return FasterRCNN.from_pretrained(path, config=model_config, precision=precision), model_config

    @staticmethod
    def schema(max_detections=36, visual_dim=2048):
        ids, box, f3d = _features()
        return {"attr_ids": ids(), "object_ids": ids(), FEATURES: f3d(max_detections, visual_dim), BOX: box()}

    @staticmethod
    def forward(model, entry):
        size, scale_wh, image = entry[SIZE], entry[SCALE], entry[IMG]
        model_out = model(images=image.unsqueeze(0), image_shapes=torch.as_tensor(size).unsqueeze(0),
                          padding="max_detections", pad_value=0.0, location="cpu")
        return FRCNN._rows(model_out, [scale_wh], 0)

    @staticmethod
    def forward_batch(model, entries):
        """Batched counterpart: images must share one (padded) size.  Returns the same dict with B rows."""
        images = torch.stack([e[IMG] for e in entries])
        sizes = torch.stack([torch.as_tensor(e[SIZE]) for e in entries])
        model_out = model(images=images, image_shapes=sizes, padding="max_detections", pad_value=0.0, location="cpu")
        out = None
        for i, e in enumerate(entries):
            row = FRCNN._rows(model_out, [e[SCALE]], i)
            if out is None:
                out = row
            else:
                for k, v in row.items():
                    out[k].extend(v)
        return out

    @staticmethod
    def _rows(model_out, scales, i):
        boxes = model_out["boxes"][i].detach().cpu().clone()
        scale = torch.as_tensor(scales[0], dtype=torch.float32)
        normalized_boxes = torch.round(rescale_box(boxes, 1 / scale))      # adapters/frcnn.py:57
        return {
            "object_ids": [model_out["obj_ids"][i].tolist()],
            "attr_ids": [model_out["attr_ids"][i].tolist()],
            BOX: [normalized_boxes.tolist()],
            FEATURES: [model_out["roi_features"][i].detach().cpu()],
        }


class Adapters:
    """Minimal counterpart of vltk/adapters/__init__.py: name -> extractor class."""

    _registry = {"frcnn": FRCNN}

    def avail(self):
        return sorted(self._registry)

    def get(self, name):
        return self._registry[name.lower()]

    def add(self, cls):
        self._registry[cls.__name__.lower()] = cls

    def is_extraction(self, name):
        return getattr(self.get(name), "_is_feature", False)
