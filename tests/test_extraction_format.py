"""CPU: the extractor's on-disk format against the reference's own fixture
tests/visualgenome/frcnn/train.arrow.  The fixture itself cannot travel to the GPU box, so the facts it pins
(column names / Arrow types / metadata keys / feature description) were read from it in the build container and
are restated here; when the reference is mounted the test re-reads the fixture and checks them directly."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

from vltk_amd.extraction import ExtractionWriter, load_extraction

FIXTURE = "/root/reference/tests/visualgenome/frcnn/train.arrow"
PINNED_COLUMNS = {"attr_ids": "list<item: float>", "box": "list<item: list<item: float>>",
                  "features": "list<item: list<item: float>>", "imgid": "string", "object_ids": "list<item: float>"}
PINNED_META = {"huggingface", "img_to_row_map", "model_config", "dataset", "processor_args"}


def _write(tmp_path, n=10, D=36, F=2048):
    g = np.random.Generator(np.random.PCG64(0))
    path = str(tmp_path / "visualgenome" / "frcnn" / "train.arrow")
    rows = dict(ids=[str(1000 + i) for i in range(n)], obj=g.integers(0, 1600, (n, D)), attr=g.integers(0, 400, (n, D)),
                box=g.uniform(0, 800, (n, D, 4)).astype(np.float32), feat=g.random((n, D, F), dtype=np.float32))
    with ExtractionWriter(path, D, F, dataset="/data/visualgenome", processor_args={"size": [800, 1333], "mode": "bilinear"},
                          model_config=None) as w:
        w.write_batch(rows["ids"][:4], rows["obj"][:4], rows["attr"][:4], rows["box"][:4], rows["feat"][:4])
        w.write_batch(rows["ids"][4:], rows["obj"][4:], rows["attr"][4:], rows["box"][4:], rows["feat"][4:])
    return path, rows


def test_roundtrip_and_layout(tmp_path):
    path, rows = _write(tmp_path)
    assert not os.path.exists(path + ".tmp")
    table, meta = load_extraction(path)
    assert {f.name: str(f.type) for f in table.schema} == PINNED_COLUMNS
    assert [f.name for f in table.schema] == sorted(PINNED_COLUMNS)                # datasets sorts feature names
    assert {k.decode() for k in table.schema.metadata} == PINNED_META
    assert table.num_rows == 10
    assert meta["img_to_row_map"] == {str(1000 + i): i for i in range(10)}
    assert meta["model_config"] == b"None" and meta["dataset"] == b"/data/visualgenome"
    assert meta["processor_args"]["size"] == [800, 1333]
    hf = json.loads(table.schema.metadata[b"huggingface"])["info"]["features"]
    assert hf["features"] == {"shape": [36, 2048], "dtype": "float32", "id": None, "_type": "Array2D"}
    assert hf["box"]["shape"] == [36, 4] and hf["attr_ids"]["_type"] == "Sequence"
    r3 = table.slice(3, 1).to_pylist()[0]
    assert r3["imgid"] == "1003"
    np.testing.assert_array_equal(np.asarray(r3["features"], dtype=np.float32), rows["feat"][3])
    np.testing.assert_array_equal(np.asarray(r3["box"], dtype=np.float32), rows["box"][3])
    np.testing.assert_array_equal(np.asarray(r3["object_ids"]), rows["obj"][3].astype(np.float32))
    np.testing.assert_array_equal(np.asarray(r3["attr_ids"]), rows["attr"][3].astype(np.float32))


def test_duplicate_and_shape_errors(tmp_path):
    w = ExtractionWriter(str(tmp_path / "x.arrow"), 4, 8)
    z = np.zeros
    w.write_batch(["a"], z((1, 4)), z((1, 4)), z((1, 4, 4)), z((1, 4, 8)))
    # a repeated imgid is skipped (the reference prints the same message, extraction.py:183-185), also inside one batch
    w.write_batch(["a"], z((1, 4)), z((1, 4)), z((1, 4, 4)), z((1, 4, 8)))
    w.write_batch(["b", "a", "b", "c"], np.arange(16.0).reshape(4, 4), z((4, 4)), z((4, 4, 4)), z((4, 4, 8)))
    assert w.rows == 3 and w.img_to_row == {"a": 0, "b": 1, "c": 2}
    with pytest.raises(ValueError, match="features must be"):
        w.write_batch(["b"], z((1, 4)), z((1, 4)), z((1, 4, 4)), z((1, 5, 8)))
    path = w.close()
    table, meta = load_extraction(path)
    assert table.num_rows == 3 and meta["img_to_row_map"] == {"a": 0, "b": 1, "c": 2}
    assert table.column("object_ids").to_pylist()[2] == [12.0, 13.0, 14.0, 15.0]      # row "c" = row 3 of that batch


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="reference fixture only exists in the build container")
def test_matches_reference_fixture(tmp_path):
    with pa.memory_map(FIXTURE) as m:
        ref = pa.ipc.open_stream(m).read_all()          # plain Arrow read: nothing from the file is executed
    assert {f.name: str(f.type) for f in ref.schema} == PINNED_COLUMNS
    assert {k.decode() for k in ref.schema.metadata} == PINNED_META
    path, _ = _write(tmp_path)
    ours, _ = load_extraction(path)
    assert [f.name for f in ours.schema] == [f.name for f in ref.schema]
    assert [f.type for f in ours.schema] == [f.type for f in ref.schema]
    a = json.loads(ref.schema.metadata[b"huggingface"])["info"]["features"]
    b = json.loads(ours.schema.metadata[b"huggingface"])["info"]["features"]
    assert a == b
    assert type(json.loads(ref.schema.metadata[b"img_to_row_map"])) is type(json.loads(ours.schema.metadata[b"img_to_row_map"]))
    assert ref.schema.metadata[b"model_config"] == ours.schema.metadata[b"model_config"] == b"None"


def test_error_inside_writer_block_leaves_nothing_behind(tmp_path):
    """An exception inside the `with` block closes the sink and removes `<split>.arrow.tmp`."""
    path = str(tmp_path / "train.arrow")
    with pytest.raises(RuntimeError):
        with ExtractionWriter(path, 4, 8) as w:
            w.write_batch(["a"], np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32), np.zeros((1, 4, 4), np.float32),
                          np.zeros((1, 4, 8), np.float32))
            raise RuntimeError("forward failed")
    assert os.listdir(tmp_path) == []


def test_extract_restores_the_callers_detection_limits(tmp_path, monkeypatch):
    """`extract()` sets roi_outputs.{max,min}_detections to the schema's width for the call only, also when it fails."""
    from types import SimpleNamespace
    from vltk_amd.adapters import FRCNN as Adapter
    (tmp_path / "train").mkdir()
    model = SimpleNamespace(roi_outputs=SimpleNamespace(max_detections=36, min_detections=36), visual_dim=2048)
    seen = {}

    def boom(cls, model, *a, **k):
        seen["limits"] = (model.roi_outputs.max_detections, model.roi_outputs.min_detections)
        raise RuntimeError("split failed")
    monkeypatch.setattr(Adapter, "_extract_splits", classmethod(boom))
    with pytest.raises(RuntimeError):
        Adapter.extract(str(tmp_path), model=(model, {}), max_detections=10)
    assert seen["limits"] == (10, 10)
    assert (model.roi_outputs.max_detections, model.roi_outputs.min_detections) == (36, 36)
