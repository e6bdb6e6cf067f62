"""-m gpu: the plugin boundary with the REAL model.

* `Adapters().get("frcnn").extract(datadir, dataset=...)` (reference abc/extraction.py:95-246, called that way by
  tests/test_adapter_extract.py:29-32 and dataset/builder.py:36-38) over a temp dir of PIL-written JPEG / PNG files:
  the Arrow files equal the same images pushed through the steps one by one;
* `adapters.FRCNN.forward` / `forward_batch` (adapters/frcnn.py:44-64) against direct model calls;
* `state_dict + config.yaml -> FRCNN.from_pretrained(dir)` (frcnn.py:1757-1922; weights-only load);
* forward handles dropped out of order.
"""
import gc
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from vltk_amd import FRCNN, adapters, make_state_dict, synthetic_images   # noqa: E402
from vltk_amd.config import Config, vg_c4_config_dict                      # noqa: E402
from vltk_amd.extraction import load_extraction                            # noqa: E402
from vltk_amd.preprocess import Preprocess                                 # noqa: E402


def small_cfg(**kw):
    d = vg_c4_config_dict(depth=50, post_nms_topk=24, detections=8, **kw)
    d["input"]["min_size_test"], d["input"]["max_size_test"] = 96, 160
    return Config(d)


@pytest.fixture(scope="module")
def built():
    cfg = small_cfg()
    sd = make_state_dict(cfg, seed=99)
    m = FRCNN(cfg, precision="fp16").load_state_dict(sd).eval()
    m.roi_outputs.nms_thresh = [0.3, 1.0]           # 1.0 suppresses nothing: every image reaches max_detections
    return cfg, sd, m


def write_images(root, dataset, spec):
    """spec: {split: [(name, (h, w), ext)]} -> files of seeded noise smoothed enough to survive JPEG."""
    from PIL import Image
    g = np.random.Generator(np.random.PCG64(17))
    for split, files in spec.items():
        os.makedirs(os.path.join(root, dataset, split), exist_ok=True)
        for name, (h, w), ext in files:
            a = g.uniform(0, 255, (h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
            img = Image.fromarray(a).resize((w, h), Image.BICUBIC)
            img.save(os.path.join(root, dataset, split, f"{name}.{ext}"))


SPEC = {"train": [("100", (120, 150), "jpg"), ("101", (200, 140), "jpg"), ("102", (90, 160), "png"),
                  ("103", (130, 130), "jpeg"), ("104", (100, 180), "jpg")],
        "val": [("200", (110, 170), "jpg"), ("201", (160, 100), "jpg")]}


def test_extract_entry_point_over_image_files(built, tmp_path):
    cfg, sd, m = built
    root = str(tmp_path)
    write_images(root, "coco2014", SPEC)
    os.makedirs(os.path.join(root, "coco2014", "annotations"), exist_ok=True)        # not a split: ignored
    res = adapters.Adapters().get("frcnn").extract(root, dataset="coco2014", model=(m, cfg), batch_size=2, max_detections=8)
    assert sorted(res) == ["train", "val"]
    assert os.path.isfile(os.path.join(root, "coco2014", "frcnn", "train.arrow"))    # abc/adapter.py:309-317
    tr = res["train"]
    assert len(tr) == 5 and tr.n_imgs == 5 and sorted(tr.imgids) == ["100", "101", "102", "103", "104"]
    assert tr.dataset == b"coco2014" or tr.dataset == "coco2014"
    assert tr.processor_args["format"] == "BGR" and tr.config["resnets"]["depth"] == 50
    # step by step: decode -> GPU Preprocess -> forward with scales_yx -> round, the same batches of two (an image's
    # result depends on the zero-padded canvas it shares with its batch, in the reference as here: legacy/processing.py:98-110)
    pre = Preprocess(cfg)
    for split, files in SPEC.items():
        files = sorted(files)                                # extract() walks the files in sorted path order
        for lo in range(0, len(files), 2):
            grp = files[lo:lo + 2]
            raws = [torch.from_numpy(adapters.decode_image_bgr(os.path.join(root, "coco2014", split, f"{n}.{e}"))) for n, _, e in grp]
            _, images, sizes, scales_yx = pre(raws, [n for n, _, _ in grp])
            out = m(images, sizes, scales_yx=scales_yx, padding="max_detections", return_tensors="pt", location="cpu")
            for i, (name, _, _) in enumerate(grp):
                row = res[split].get(name)
                assert row["imgid"] == name
                np.testing.assert_array_equal(np.asarray(row["features"], np.float32), out["roi_features"][i].numpy())
                np.testing.assert_array_equal(np.asarray(row["box"], np.float32), torch.round(out["boxes"][i]).numpy())
                np.testing.assert_array_equal(np.asarray(row["object_ids"], np.float32), out["obj_ids"][i].float().numpy())
                np.testing.assert_array_equal(np.asarray(row["attr_ids"], np.float32), out["attr_ids"][i].float().numpy())
    # Adapter.load (abc/adapter.py:424-462) and the `dataset_name=` spelling of the in-tree caller (builder.py:36-38)
    again = adapters.FRCNN.load(root, dataset_name="coco2014")
    assert sorted(again) == ["train", "val"] and again["val"].table.equals(res["val"].table)
    one = adapters.FRCNN.load(root, split="val", dataset_name="coco2014")
    assert len(one) == 2 and one.has("201") and not one.has("100")
    res2 = adapters.FRCNN.extract(root, dataset_name="coco2014", splits="val", subset_ids={"201"}, model=(m, cfg), max_detections=8)
    assert sorted(res2) == ["val"] and res2["val"].imgids == ("201",)
    assert res2["val"].get("201")["imgid"] == "201" and len(res2["val"].get("201")["features"]) == 8


def test_extract_reference_loop_with_cpu_processor(built, tmp_path):
    """`processor="reference"`: the reference's own per-image loop (extraction.py:142-220) -- the live adapter's CPU
    processor chain, `cls.forward` at batch 1, `round(rescale_box(boxes, 1 / wh_scale))`."""
    cfg, sd, m = built
    root = str(tmp_path)
    write_images(root, "vg", {"train": SPEC["train"][:2]})
    pc = dict(adapters.FRCNN.default_processor, size=96, max_size=160)
    res = adapters.FRCNN.extract(root, dataset="vg", processor_config=pc, model=(m, cfg), max_detections=8)
    proc = adapters.ReferenceImageProcessor(**pc)
    for name, (h, w), ext in SPEC["train"][:2]:
        path = os.path.join(root, "vg", "train", f"{name}.{ext}")
        img = proc(path)
        assert proc._rawsize.tolist() == [w, h] and min(proc._size.tolist()) <= 96       # PIL (W, H)
        entry = {"image": img, "size": proc._size, "wh_scale": proc._scale}
        rows = adapters.FRCNN.forward(m, entry)
        row = res["train"].get(name)
        np.testing.assert_array_equal(np.asarray(row["features"], np.float32), rows["features"][0].numpy())
        assert row["box"] == rows["box"][0] and row["object_ids"] == [float(v) for v in rows["object_ids"][0]]


def test_adapter_forward_and_forward_batch_with_real_model(built):
    cfg, sd, m = built
    x = torch.from_numpy(synthetic_images(3, 96, 128, seed=3))
    entries = [{"image": x[i], "size": torch.tensor([96, 128]), "wh_scale": torch.tensor([0.5, 0.25]), "imgid": str(i)}
               for i in range(3)]
    direct = m(x, torch.tensor([[96, 128]] * 3), padding="max_detections", pad_value=0.0, location="cpu")
    batch = adapters.FRCNN.forward_batch(m, entries)
    assert set(batch) == {"object_ids", "attr_ids", "box", "features"} and len(batch["features"]) == 3
    for i, e in enumerate(entries):
        one = adapters.FRCNN.forward(m, e)                       # batch 1, as the reference runs it
        assert all(len(v) == 1 for v in one.values())
        exp_box = torch.round(adapters.rescale_box(direct["boxes"][i].clone(), 1 / e["wh_scale"])).tolist()
        for rows, j in ((one, 0), (batch, i)):
            assert rows["object_ids"][j] == direct["obj_ids"][i].tolist()
            assert rows["attr_ids"][j] == direct["attr_ids"][i].tolist()
            assert rows["box"][j] == exp_box
            assert torch.equal(rows["features"][j], direct["roi_features"][i])
        assert one["features"][0].shape == (8, 2048) and one["features"][0].device.type == "cpu"


def test_setup_synthetic_returns_model_and_config():
    model, model_config = adapters.FRCNN.setup(synthetic=True, model_config=small_cfg())
    assert isinstance(model, FRCNN) and model_config.RESNETS.DEPTH == 50 and not model.training
    out = model(torch.from_numpy(synthetic_images(1, 96, 128, seed=1)), torch.tensor([[96, 128]]))
    assert int(out["preds_per_image"][0]) >= 1


def test_from_pretrained_round_trip(built, tmp_path):
    """frcnn.py:1757-1922, local branch: a directory with pytorch_model.bin + config.yaml; also a direct file path with
    config=; the old gamma/beta key names; and the loader's errors."""
    cfg, sd, m = built
    d = tmp_path / "ckpt"
    d.mkdir()
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, str(d / "pytorch_model.bin"))
    cfg.dump_yaml(str(d / "config.yaml"))
    x = torch.from_numpy(synthetic_images(2, 96, 128, seed=8))
    sh = torch.tensor([[96, 128], [90, 120]])
    m(x, sh)
    ref = {k: v.clone() for k, v in m.forward_padded().items()}
    loaded = FRCNN.from_pretrained(str(d), precision="fp16")
    assert not loaded.training and loaded.config.to_dict() == cfg.to_dict()
    loaded.roi_outputs.nms_thresh = [0.3, 1.0]
    loaded(x, sh)
    for k, v in ref.items():
        assert torch.equal(v, loaded.forward_padded()[k]), k
    # direct file path + explicit config object; gamma/beta names (frcnn.py:1862-1872)
    old = {k.replace("norm.weight", "norm.gamma").replace("norm.bias", "norm.beta"): torch.from_numpy(np.asarray(v))
           for k, v in sd.items()}
    torch.save(old, str(tmp_path / "old.bin"))
    again = FRCNN.from_pretrained(str(tmp_path / "old.bin"), config=cfg, precision="fp16")
    again.roi_outputs.nms_thresh = [0.3, 1.0]
    again(x, sh)
    assert torch.equal(ref["roi_features"], again.forward_padded()["roi_features"])
    # the plugin's setup() over the same directory (adapters/frcnn.py:26-32 with a local path)
    model, model_config = adapters.FRCNN.setup(path=str(d))
    assert model_config.to_dict() == cfg.to_dict() and isinstance(model, FRCNN)
    with pytest.raises(EnvironmentError, match="no file named pytorch_model.bin"):
        (tmp_path / "empty").mkdir()
        cfg.dump_yaml(str(tmp_path / "empty" / "config.yaml"))
        FRCNN.from_pretrained(str(tmp_path / "empty"))
    with pytest.raises(OSError, match="Unable to load weights"):
        (tmp_path / "bad.bin").write_bytes(b"not a checkpoint")
        FRCNN.from_pretrained(str(tmp_path / "bad.bin"), config=cfg)
    with pytest.raises(EnvironmentError):
        FRCNN.from_pretrained("unc-nlp/frcnn-vg-finetuned", config=cfg)          # fetch-by-name needs the network


def test_dropped_handles_close_their_tickets_in_order(built):
    cfg, sd, m = built
    x = torch.from_numpy(synthetic_images(2, 96, 128, seed=8)).cuda()
    sh = torch.tensor([[96, 128], [90, 120]])
    m(x, sh)
    ref = m.forward_padded()["roi_features"].clone()
    a, b, c = (m.forward_async(x, sh) for _ in range(3))
    del b                               # dropped out of order: closes a's ticket, then its own
    gc.collect()
    assert torch.equal(a.wait_raw()["roi_features"], ref)          # a's result is still there
    assert torch.equal(c.wait_raw()["roi_features"], ref)
    many = [m.forward_async(x, sh) for _ in range(4)]              # all four slots are free again
    del many
    gc.collect()
    assert torch.equal(m(x, sh)["roi_features"][0], ref[0][: int(m.forward_padded()["preds_per_image"][0])])
