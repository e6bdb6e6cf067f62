"""On-disk format of the extractor output: the Arrow IPC *stream* file
`<datadir>/<dataset>/frcnn/<split>.arrow` that unmodified vltk loaders read
(`Adapter._load_one_arrow`, reference vltk/abc/adapter.py:382-409).

Layout pinned by the reference's fixture tests/visualgenome/frcnn/train.arrow (written by
`VisnExtraction.extract` + `_save_dataset`, vltk/abc/extraction.py:157-246, abc/adapter.py:360-379,
utils/base.py:71-88): columns `attr_ids list<float>`, `box list<list<float>>`,
`features list<list<float>>`, `imgid string`, `object_ids list<float>`; schema metadata
`huggingface` (JSON description of the `datasets` features: Sequence / Array2D), `img_to_row_map`
(JSON imgid -> row), `model_config`, `dataset`, `processor_args` (dicts JSON-encoded, everything
else `str()`).

Unlike the reference (which keeps the whole table in an in-memory BufferOutputStream and loses it on
a crash), batches are streamed to `<file>.tmp` as they arrive and the final file -- same bytes
layout, metadata included -- is produced at close().
"""
import json
import os

import numpy as np
import pyarrow as pa

META_NAMES = ("img_to_row_map", "model_config", "dataset", "processor_args")


def _hf_features(max_detections, visual_dim):
    seq = {"feature": {"dtype": "float32", "id": None, "_type": "Value"}, "length": -1, "id": None, "_type": "Sequence"}
    return {"info": {"features": {
        "imgid": {"dtype": "string", "id": None, "_type": "Value"},
        "attr_ids": dict(seq), "object_ids": dict(seq),
        "features": {"shape": [max_detections, visual_dim], "dtype": "float32", "id": None, "_type": "Array2D"},
        "box": {"shape": [max_detections, 4], "dtype": "float32", "id": None, "_type": "Array2D"},
    }}}


def _base_schema():
    f32 = pa.float32()
    return pa.schema([
        pa.field("attr_ids", pa.list_(f32)), pa.field("box", pa.list_(pa.list_(f32))),
        pa.field("features", pa.list_(pa.list_(f32))), pa.field("imgid", pa.string()),
        pa.field("object_ids", pa.list_(f32)),
    ])


def _encode_meta(meta):
    out = {}
    for k, v in meta.items():          # utils/base.py:77-83
        if isinstance(v, dict):
            out[k] = json.dumps(v).encode("utf-8")
        elif isinstance(v, set):
            out[k] = "\n".join(v).encode("utf-8")
        else:
            out[k] = str(v).encode("utf-8")
    return out


def _list2d(a):
    """[B, D, F] float32 -> Arrow list<list<float>> without Python-level loops."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    B, D, F = a.shape
    inner = pa.ListArray.from_arrays(pa.array(np.arange(0, (B * D + 1) * F, F, dtype=np.int32)), pa.array(a.reshape(-1)))
    return pa.ListArray.from_arrays(pa.array(np.arange(0, (B + 1) * D, D, dtype=np.int32)), inner)


def _list1d(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    B, D = a.shape
    return pa.ListArray.from_arrays(pa.array(np.arange(0, (B + 1) * D, D, dtype=np.int32)), pa.array(a.reshape(-1)))


class ExtractionWriter:
    def __init__(self, savefile, max_detections=36, visual_dim=2048, dataset=None, processor_args=None,
                 model_config=None):
        self.savefile = savefile
        self.D, self.F = int(max_detections), int(visual_dim)
        self.meta = {"model_config": model_config, "dataset": dataset, "processor_args": processor_args or {}}
        self.img_to_row = {}
        self.schema = _base_schema()
        os.makedirs(os.path.dirname(os.path.abspath(savefile)), exist_ok=True)
        self._tmp = savefile + ".tmp"
        self._sink = pa.OSFile(self._tmp, "wb")
        self._writer = pa.ipc.new_stream(self._sink, self.schema)
        self.rows = 0

    def write_batch(self, imgids, object_ids, attr_ids, boxes, features):
        """imgids: B strings; object_ids/attr_ids [B, D]; boxes [B, D, 4]; features [B, D, F] (any float/int array-likes)."""
        feats = np.asarray(features, dtype=np.float32)
        if feats.shape[1:] != (self.D, self.F):
            raise ValueError(f"features must be [B, {self.D}, {self.F}], got {feats.shape}")
        B = feats.shape[0]
        ids = [str(i) for i in imgids]
        if len(ids) != B:
            raise ValueError(f"{len(ids)} imgids for {B} rows")
        # duplicates: the reference prints "skipping ..." and then writes the row anyway, leaving an orphan row behind
        # the re-pointed img_to_row_map entry (extraction.py:183-185).  Here the row really is skipped; nothing is
        # mutated before the whole batch has been looked at.
        keep, fresh = [], set()
        for i, iid in enumerate(ids):
            if iid in self.img_to_row or iid in fresh:
                print(f"skipping {iid}. Already written to table")
                continue
            fresh.add(iid)
            keep.append(i)
        if not keep:
            return
        attr_ids = np.asarray(attr_ids).reshape(B, self.D)
        object_ids = np.asarray(object_ids).reshape(B, self.D)
        boxes = np.asarray(boxes).reshape(B, self.D, 4)
        if len(keep) != B:
            feats, attr_ids, object_ids, boxes = feats[keep], attr_ids[keep], object_ids[keep], boxes[keep]
            ids = [ids[i] for i in keep]
            B = len(keep)
        batch = pa.record_batch([
            _list1d(attr_ids), _list2d(boxes), _list2d(feats), pa.array(ids, pa.string()), _list1d(object_ids),
        ], schema=self.schema)
        self._writer.write_batch(batch)
        for i, iid in enumerate(ids):
            self.img_to_row[iid] = self.rows + i
        self.rows += B

    def close(self):
        self._writer.close()
        self._sink.close()
        meta = dict(self.meta)
        meta["img_to_row_map"] = self.img_to_row
        md = {b"huggingface": json.dumps(_hf_features(self.D, self.F)).encode("utf-8")}
        for k, v in _encode_meta({k: meta[k] for k in META_NAMES}).items():
            md[k.encode()] = v
        final_schema = self.schema.with_metadata(md)
        with pa.memory_map(self._tmp) as src, pa.OSFile(self.savefile, "wb") as dst:
            reader = pa.ipc.open_stream(src)
            with pa.ipc.new_stream(dst, final_schema) as w:
                for b in reader:
                    w.write_batch(pa.RecordBatch.from_arrays(b.columns, schema=final_schema))
        os.remove(self._tmp)
        return self.savefile

    def __enter__(self):
        return self

    def abort(self):
        """Drop what was written: close the sink, remove `<savefile>.tmp` (an error inside the `with` block must not leave either behind)."""
        for obj in (self._writer, self._sink):
            try:
                obj.close()
            except Exception:
                pass
        if os.path.exists(self._tmp):
            os.remove(self._tmp)

    def __exit__(self, *exc):
        if exc[0] is None:
            self.close()
        else:
            self.abort()


def load_extraction(path):
    """Counterpart of Adapter._load_one_arrow (adapter.py:382-409): (pa.Table, meta_dict)."""
    with pa.memory_map(path) as m:
        table = pa.ipc.open_stream(m).read_all()
    meta = {}
    for k, v in (table.schema.metadata or {}).items():
        if k == b"huggingface":
            continue
        try:
            meta[k.decode()] = json.loads(v)
        except Exception:
            meta[k.decode()] = v
    return table, meta


def extract(model, entries, savedir, split="train", dataset=None, batch_size=32, processor_args=None):
    """Batched counterpart of VisnExtraction.extract's hot loop (extraction.py:142-220) over already-processed
    entries (dicts with "image" [3,H,W], "size", "wh_scale", "imgid"): forward in batches, rescale + round the
    boxes as the reference adapter does, stream rows to `<savedir>/<split>.arrow`."""
    from .adapters import FRCNN as Adapter, FEATURES, BOX, IMGID
    D = int(model.roi_outputs.max_detections)
    path = os.path.join(savedir, f"{split}.arrow")
    cfgd = model.config.to_dict() if hasattr(model.config, "to_dict") else None
    with ExtractionWriter(path, D, 2048, dataset=dataset, processor_args=processor_args, model_config=cfgd) as w:
        buf = []

        def flush():
            if not buf:
                return
            out = Adapter.forward_batch(model, buf)
            w.write_batch([e[IMGID] for e in buf], np.asarray(out["object_ids"], dtype=np.float32),
                          np.asarray(out["attr_ids"], dtype=np.float32), np.asarray(out[BOX], dtype=np.float32),
                          np.stack([np.asarray(f) for f in out[FEATURES]]))
            buf.clear()
        for e in entries:
            buf.append(e)
            if len(buf) == batch_size:
                flush()
        flush()
    return path
