"""`VisnExtraction`-compatible FRCNN extractor plugin (the reference's plugin boundary).

Mirrors `vltk.adapters.frcnn.FRCNN` (reference vltk/adapters/frcnn.py:10-64), the class method
`VisnExtraction.extract` it inherits (vltk/abc/extraction.py:95-246), `Adapter.load`
(vltk/abc/adapter.py:424-462) and the registry lookup `Adapters().get("frcnn")`
(vltk/adapters/__init__.py:53): same static methods `setup() -> (model, model_config)`,
`schema(max_detections, visual_dim)`, `forward(model, entry)` with the same entry keys
(vltk/vars.py: "image", "size", "wh_scale", ...) and the same return layout (dict of length-1
lists, extended row-wise by the caller, abc/extraction.py:211-213); `extract(datadir, ...)`
with the reference's signature, split discovery and save path
(`<datadir>/[<dataset>/]frcnn/<split>.arrow`, abc/adapter.py:284-320), returning
`{split: FRCNN(arrow_table=..., split=..., meta_dict=...)}` objects that answer
`len()`, `[i]`, `.get(imgid)`, `.img_to_row_map`, `.imgids` like the reference's Arrow-backed
adapters (abc/adapter.py:169-222).  `forward_batch` is the batched extension (the reference runs
batch 1).

`setup()` upstream fetches "unc-nlp/frcnn-vg-finetuned" by name (network); here it takes a LOCAL
checkpoint directory (env VLTK_AMD_FRCNN_PATH or argument) or, with `synthetic=True`, the seeded
synthetic weights used by the tests and the benchmark.

Two processing modes in `extract`:
  * default (no `processor`, no `processor_config`): the pipelined GPU path (pipeline.py) -- files are decoded with PIL,
    cross PCIe as uint8, and are resized / normalised on the GPU under the legacy `Preprocess` contract (BGR 0-255,
    pixel mean, `scales_yx = raw/size`, legacy/processing.py:29-150), batched through the HIP forward;
  * `processor="reference"` / a `processor_config` dict / any callable: the reference's own per-image loop
    (extraction.py:142-220) with `cls.forward` at batch 1.  "reference" builds the live adapter's
    `default_processor` (RGB 0-1, mean/255 -- self-declared incorrect upstream, adapters/frcnn.py:12).
"""
import inspect
import os

import numpy as np
import torch

from .config import Config, vg_c4_config

# string keys of vltk/vars.py:38-60 that the extraction loop uses
IMG, SIZE, SCALE, RAWSIZE, FILEPATH, IMGID, SPLIT = "image", "size", "wh_scale", "rawsize", "filepath", "imgid", "split"
FEATURES, BOX = "features", "box"
SPLITALIASES = ("test", "dev", "eval", "val", "validation", "evaluation", "train")       # vltk/vars.py:63-71
IMGFILES = ("jpeg", "jpg", "png")                                                         # abc/adapter.py:25


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


def _collect_args(func, kwargs):
    """Name-matched subset of kwargs (the reference's inspection.collect_args_to_func, inspection.py:99-117)."""
    names = set(inspect.signature(func).parameters)
    return {k: v for k, v in kwargs.items() if k in names}


class ReferenceImageProcessor:
    """The live adapter's `default_processor` chain on the CPU, as the reference runs it
    (adapters/frcnn.py:13-23, processing/image.py:52-145): `FromFile -> Resize(size, max_size) -> ToTensor ->
    Normalize(mean, std)`.  `_size` / `_rawsize` are PIL (W, H) and `_scale = size / rawsize` in (w, h), exactly the
    attributes `get_size / get_scale / get_rawsize` read back (image.py:12-49).  The output-size rule is torchvision's
    `Resize(int, max_size=)`: an absent third-party dependency, restated from its published behaviour (parity unpinned)."""

    def __init__(self, size=800, max_size=1333, mean=None, std=None, mode="bilinear", pad_value=0.0, transforms=None, **_):
        self.size, self.max_size = int(size), (int(max_size) if max_size else None)
        self.mean = None if mean is None else torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
        self.std = None if std is None else torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
        self.mode = mode
        self._size = self._rawsize = self._scale = None

    def output_wh(self, w, h):
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = self.size, int(self.size * long / short)
        if self.max_size is not None and new_long > self.max_size:
            new_short, new_long = int(self.max_size * new_short / new_long), self.max_size
        return (new_short, new_long) if w <= h else (new_long, new_short)

    def __call__(self, filepath):
        from PIL import Image
        img = Image.open(filepath).convert("RGB")
        self._rawsize = torch.tensor(img.size)
        img = img.resize(self.output_wh(*img.size), {"bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC,
                                                     "nearest": Image.NEAREST}.get(self.mode, Image.BILINEAR))
        self._size = torch.tensor(img.size)
        self._scale = torch.tensor([self._size[0] / self._rawsize[0], self._size[1] / self._rawsize[1]])
        t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)    # ToTensor
        if self.mean is not None:
            t = (t - self.mean) / self.std
        return t


def _proc_attr(processor, name):
    """get_size / get_scale / get_rawsize of processing/image.py:12-49: the last transform that carries the attribute
    (or the processor object itself, for one-object processors)."""
    val = None
    for t in getattr(processor, "transforms", None) or ():
        if hasattr(t, name):
            val = getattr(t, name)
    if val is None:
        val = getattr(processor, name, None)
    return val


def decode_image_bgr(path):
    """File -> raw HWC uint8 in the model's channel order (INPUT.FORMAT "BGR", legacy/processing.py:113-120)."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(a[:, :, ::-1])


class FRCNN:
    """Registry key = class name lower-cased = "frcnn".  The class is the plugin (static `setup/schema/forward`, class
    methods `extract/load`); an INSTANCE is one extracted split, Arrow-backed like the reference's adapters."""

    _is_feature = True
    _is_annotation = False
    _batch_size = 128           # Arrow write batch of the reference's caller (abc/extraction.py:26)
    _meta_names = ["img_to_row_map", "dataset", "processor_args"]          # extraction.py:20-24

    # vltk/adapters/frcnn.py:13-23 (VisionConfig kwargs; kept as a plain description)
    default_processor = {
        "transforms": ["FromFile", "Resize", "ToTensor", "Normalize"],
        "size": 800, "max_size": 1333, "mode": "bilinear", "pad_value": 0.0,
        "mean": [102.9801 / 255, 115.9465 / 255, 122.7717 / 255], "std": [1.0, 1.0, 1.0],
    }

    # ---- an extracted split (abc/adapter.py:47-80, 169-222) ----
    def __init__(self, arrow_table, meta_dict=None, split=None, info=None, path=None, **kwargs):
        self.table, self.split, self.info, self.path = arrow_table, split, info, path
        meta_dict = dict(meta_dict or {})
        self._img_to_row_map = meta_dict.get("img_to_row_map", {})
        for k, v in meta_dict.items():
            k = k if isinstance(k, str) else k.decode()
            if k != "img_to_row_map":
                setattr(self, "meta_" + k, v)
        self._meta_dict = meta_dict

    def __len__(self):
        return self.table.num_rows

    def __getitem__(self, i):
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self.table.slice(i, 1).to_pylist()[0]

    @property
    def meta_dict(self):
        return self._meta_dict

    @property
    def img_to_row_map(self):
        return self._img_to_row_map

    @property
    def imgids(self):
        return tuple(self._img_to_row_map.keys())

    @property
    def n_imgs(self):
        return len(self.imgids)

    @property
    def name(self):
        return type(self).__name__.lower()

    @property
    def dataset(self):
        return self._meta_dict.get("dataset")

    @property
    def processor_args(self):
        return self._meta_dict.get("processor_args")

    @property
    def config(self):
        return self._meta_dict.get("model_config")

    def has(self, img_id):
        return img_id in self._img_to_row_map

    def get_idx(self, img_id):
        return self._img_to_row_map[img_id]

    def get(self, img_id):
        return self[self._img_to_row_map[img_id]]

    # ---- the plugin ----
    @staticmethod
    def setup(path=None, synthetic=False, precision=None, seed=1234, model_config=None):
        from .frcnn import FRCNN as FasterRCNN
        path = path or os.environ.get("VLTK_AMD_FRCNN_PATH")
        if synthetic or path is None:
            if not synthetic:
                raise EnvironmentError(
                    "Can't load weights for 'unc-nlp/frcnn-vg-finetuned': fetch-by-name needs the network. "
                    "Point VLTK_AMD_FRCNN_PATH at a local directory with pytorch_model.bin + config.yaml, "
                    "or call setup(synthetic=True).")
            from .weights import make_state_dict
            model_config = model_config if model_config is not None else vg_c4_config()
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

    # ---- path helpers (abc/adapter.py:284-349) ----
    @staticmethod
    def _get_valid_search_pathes(searchdir, name=None, splits=None):
        if splits is None:
            splits = SPLITALIASES
        elif isinstance(splits, str):
            splits = [splits]
        assert os.path.isdir(searchdir), f"The specificed datadir, {searchdir}, does not exist"
        if name is not None:
            searchdir = os.path.join(searchdir, name)
            assert os.path.isdir(searchdir), f"{searchdir} is not a dir"
        final_paths, valid_splits = [], []
        for splt in splits:
            path = os.path.join(searchdir, splt)
            if not os.path.isdir(path):
                continue
            final_paths.append(path)
            valid_splits.append(splt)
        assert final_paths, (searchdir, name, splits)
        return final_paths, valid_splits

    @staticmethod
    def _make_save_path(searchdir, dataset_name, extractor_name):
        if dataset_name is not None:
            savepath = os.path.join(searchdir, dataset_name, extractor_name)
        else:
            savepath = os.path.join(searchdir, extractor_name)
        print(f"will write to {savepath}")
        os.makedirs(savepath, exist_ok=True)
        return savepath

    @staticmethod
    def _iter_files(searchdirs):
        """Image files under the split directories (`**/*.{jpeg,jpg,png}`, abc/adapter.py:319-349); sorted, where the
        reference iterates a `set` in arbitrary order (the row order is free: readers go through img_to_row_map)."""
        from pathlib import Path
        files = set()
        for d in searchdirs:
            for suffix in IMGFILES:
                files.update(str(p) for p in Path(d).glob(f"**/*.{suffix}"))
        return sorted(files)

    @classmethod
    def extract(cls, datadir, processor_config=None, splits=None, subset_ids=None, dataset=None, img_format="jpg",
                processor=None, **kwargs):
        """`VisnExtraction.extract` (abc/extraction.py:95-246).  Extra keyword arguments are matched by name, as
        upstream (extraction.py:198): `dataset_name` (what the one in-tree caller passes, dataset/builder.py:36-38),
        `schema`'s (max_detections, visual_dim), `setup`'s (path, synthetic, precision, seed, model_config), plus
        `model=(model, model_config)` to reuse a built model and `batch_size` (images per forward, default 32).
        With torch.distributed initialised the images of every split shard across the ranks (parallel.py); the dict is
        returned on rank 0 (empty elsewhere).  `model.roi_outputs.{max,min}_detections` are set to the schema's width for
        the call and restored afterwards.  The reference-processor mode (`processor="reference"` / a `processor_config`)
        keeps an upstream quirk on purpose: `entry["size"]` is PIL's (W, H) and goes to the model as `image_shapes`, whose
        rows mean (h, w) (adapters/frcnn.py:50-52 with processing/image.py:139-141; upstream calls that processor
        "incorrect" itself, adapters/frcnn.py:12) -- boxes are therefore clipped to the transposed extent there.  The
        default GPU path follows the legacy (correct) contract."""
        from .extraction import ExtractionWriter, load_extraction
        dataset_name = dataset if dataset is not None else kwargs.pop("dataset_name", None)
        kwargs.pop("dataset_name", None)
        extractor_name = cls.__name__.lower()
        searchdirs, valid_splits = cls._get_valid_search_pathes(datadir, dataset_name, splits)
        savedir = cls._make_save_path(datadir, dataset_name, extractor_name)
        prebuilt = kwargs.pop("model", None)
        batch_size = int(kwargs.pop("batch_size", 32))
        if prebuilt is not None:
            model, model_config = prebuilt
        else:
            model, model_config = cls.setup(**_collect_args(cls.setup, kwargs))
        setattr(cls, "model", model)                                        # extraction.py:127
        sch = _collect_args(cls.schema, kwargs)
        D = int(sch.get("max_detections", model.roi_outputs.max_detections))
        F = int(sch.get("visual_dim", getattr(model, "visual_dim", 2048)))
        # the caller's model keeps its limits: they are set for this call and restored on every way out
        saved_limits = (model.roi_outputs.max_detections, model.roi_outputs.min_detections)
        try:
            if D != int(model.roi_outputs.max_detections):                  # rows are padded to the schema's width
                model.roi_outputs.max_detections = D
                model.roi_outputs.min_detections = min(int(model.roi_outputs.min_detections), D)
            splitdict = cls._extract_splits(model, model_config, searchdirs, valid_splits, savedir, dataset_name, subset_ids,
                                            processor, processor_config, batch_size, D, F, kwargs)
        finally:
            model.roi_outputs.max_detections, model.roi_outputs.min_detections = saved_limits
        return splitdict

    @classmethod
    def _extract_splits(cls, model, model_config, searchdirs, valid_splits, savedir, dataset_name, subset_ids, processor,
                        processor_config, batch_size, D, F, kwargs):
        from .extraction import ExtractionWriter, load_extraction
        # files -> per split (id, path), in the reference's terms: split = parent directory, id = stem up to the first dot
        print(f"extracting from {searchdirs}")
        per_split = {s: [] for s in valid_splits}
        seen = {s: set() for s in valid_splits}
        for path in cls._iter_files(searchdirs):
            parts = path.split("/")
            split, img_id = parts[-2], parts[-1].split(".")[0]
            if split not in per_split:
                continue
            if subset_ids is not None and img_id not in subset_ids:
                continue
            if img_id in seen[split]:
                print(f"skipping {img_id}. Already written to table")       # extraction.py:183-185 (here it does skip)
                continue
            seen[split].add(img_id)
            per_split[split].append((img_id, path))
        cfgd = model_config.to_dict() if hasattr(model_config, "to_dict") else model_config
        gpu_path = processor is None and processor_config is None
        if not gpu_path:
            if processor == "reference" or processor is None:
                pargs = dict(processor_config) if isinstance(processor_config, dict) else (
                    processor_config.to_dict() if processor_config is not None else dict(cls.default_processor))
                processor = ReferenceImageProcessor(**pargs)
            else:
                pargs = dict(processor_config) if isinstance(processor_config, dict) else {}
                if isinstance(processor, type):
                    processor = processor(**pargs)
        else:
            pargs = {"size": [model_config.INPUT.MIN_SIZE_TEST, model_config.INPUT.MAX_SIZE_TEST],
                     "mean": list(model_config.MODEL.PIXEL_MEAN), "std": list(model_config.MODEL.PIXEL_STD),
                     "format": "BGR", "pad_value": float(model_config.PAD_VALUE), "device": "gpu"}
        import torch.distributed as dist
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        if gpu_path and world > 1:      # checked for EVERY split before the first one is written
            short = {sp: len(it) for sp, it in per_split.items() if it and len(it) < world}
            if short:
                raise ValueError(f"splits with fewer images than ranks ({world}): {short}; nothing was written")
        if not gpu_path and world > 1:
            raise NotImplementedError("the per-image reference loop is single-process; use the default GPU path")
        splitdict = {}
        for split, items in per_split.items():
            if not items:
                continue
            savefile = os.path.join(savedir, f"{split}.arrow")
            if gpu_path:
                cls._extract_split_gpu(model, model_config, items, savefile, dataset_name, pargs, cfgd, batch_size, F,
                                       rank, world)
            else:
                fkw = _collect_args(cls.forward, kwargs)
                fkw.pop("model", None), fkw.pop("entry", None)
                with ExtractionWriter(savefile, D, F, dataset=dataset_name, processor_args=pargs, model_config=cfgd) as w:
                    for img_id, path in items:
                        entry = {FILEPATH: path, IMGID: img_id, SPLIT: split}
                        entry[IMG] = processor(path)
                        entry[SIZE] = _proc_attr(processor, "_size")
                        entry[SCALE] = _proc_attr(processor, "_scale")
                        entry[RAWSIZE] = _proc_attr(processor, "_rawsize")
                        out = cls.forward(model=model, entry=entry, **fkw)
                        assert isinstance(out, dict), "model outputs should be in dict format"
                        w.write_batch([img_id], np.asarray(out["object_ids"], np.float32), np.asarray(out["attr_ids"], np.float32),
                                      np.asarray(out[BOX], np.float32), np.stack([np.asarray(f) for f in out[FEATURES]]))
            if rank == 0:
                table, meta = load_extraction(savefile)
                print(f"Success! You wrote {table.num_rows} entry(s) and {os.path.getsize(savefile) >> 20} mb")
                print(f"Located: {savefile}")
                splitdict[split] = cls(arrow_table=table, split=split, info=None, meta_dict=meta, path=savefile)
        return splitdict

    @classmethod
    def _extract_split_gpu(cls, model, model_config, items, savefile, dataset_name, pargs, cfgd, batch_size, F, rank, world):
        from .parallel import shard_indices
        from .pipeline import ExtractionPipeline
        from .preprocess import Preprocess
        pipe = ExtractionPipeline(model, Preprocess(model_config, device=model.device), savefile, batch_size=batch_size,
                                  visual_dim=F, dataset=dataset_name, processor_args=pargs, model_config=cfgd)
        pipe.set_global_ids([i for i, _ in items])
        lo, hi = shard_indices(len(items), rank, world)
        mine = items[lo:hi]
        return pipe.run(_LazyImages(mine), n_items=len(items))

    @classmethod
    def load(cls, path, split=None, dataset_name=None, config=None, dataset=None):
        """`Adapter.load` (abc/adapter.py:424-462): an `.arrow` path, or `<path>/[<dataset>/]frcnn/<split>.arrow`
        (one split, or `{split: ...}` over every split present)."""
        from .extraction import load_extraction
        dataset_name = dataset_name if dataset_name is not None else dataset
        if ".arrow" in path:
            table, meta = load_extraction(path)
            return cls(arrow_table=table, split=split, meta_dict=meta, path=path)
        if dataset_name is not None:
            path = os.path.join(path, dataset_name)
        path = os.path.join(path, cls.__name__.lower())
        if split is not None:
            f = os.path.join(path, f"{split}.arrow")
            assert os.path.isfile(f), f"{f} does not exist"
            table, meta = load_extraction(f)
            return cls(arrow_table=table, split=split, meta_dict=meta, path=f)
        out = {}
        for s in SPLITALIASES:
            f = os.path.join(path, f"{s}.arrow")
            if os.path.isfile(f):
                table, meta = load_extraction(f)
                out[s] = cls(arrow_table=table, split=s, meta_dict=meta, path=f)
        return out


class _LazyImages:
    """(imgid, raw HWC uint8 BGR) pairs decoded on demand by the pipeline's loader thread."""

    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        for img_id, path in self.items:
            yield img_id, decode_image_bgr(path)


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
