"""Pipelined extraction loop (vltk_amd/pipeline.py): host logic with a stand-in model on the CPU (one rank and two gloo
ranks), and -- on the GPU -- the pipeline's Arrow file against the same images pushed through the steps one by one."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_FAKE = r"""
import numpy as np, torch
from vltk_amd.parallel import OutputBlock, output_spec

class RO:
    max_detections = 3

class FakeModel:
    '''Deterministic stand-in: every output is a function of the image's first pixel (= its global index).'''
    device = torch.device("cpu")
    roi_outputs = RO()
    F = 8
    def __call__(self, images, sizes, scales_yx=None):
        B = images.shape[0]
        blk = OutputBlock(output_spec(B, 3, self.F))
        v = images[:, 0, 0, 0]
        blk["obj_ids"].copy_((v.view(B, 1) * 10 + torch.arange(3)).long())
        blk["attr_ids"].copy_((v.view(B, 1) * 100 + torch.arange(3)).long())
        blk["obj_probs"].fill_(0.5); blk["attr_probs"].fill_(0.25)
        blk["boxes"].copy_((v.view(B, 1, 1) + torch.tensor([0.5, 1.5, 2.5, 3.49])).expand(B, 3, 4) * scales_yx[:, :1].view(B, 1, 1))
        blk["preds_per_image"].fill_(3)
        blk["roi_features"].copy_(v.view(B, 1, 1).expand(B, 3, self.F) + torch.arange(self.F) / 16)
        self._blk = blk
    def forward_padded(self):
        return self._blk

def fake_preprocess(raws, ids):
    x = torch.stack([torch.as_tensor(r).float().permute(2, 0, 1) for r in raws])
    n = x.shape[0]
    return ids, x, torch.tensor([[4, 6]] * n), torch.full((n, 2), 2.0)

def items(lo, hi):
    return [(f"img{i}", np.full((4, 6, 3), i, dtype=np.uint8)) for i in range(lo, hi)]

def check(path, n):
    from vltk_amd.extraction import load_extraction
    table, meta = load_extraction(path)
    assert table.num_rows == n
    rows = {r["imgid"]: r for r in table.to_pylist()}
    assert sorted(rows) == sorted(f"img{i}" for i in range(n))
    for i in range(n):
        r = rows[f"img{i}"]
        assert r["object_ids"] == [10.0 * i, 10.0 * i + 1, 10.0 * i + 2]
        assert r["attr_ids"] == [100.0 * i, 100.0 * i + 1, 100.0 * i + 2]
        # boxes (i + .5, 1.5, 2.5, 3.49) * 2 -> rounded half to even
        assert r["box"][0] == [float(np.round(np.float32(2 * (i + d)))) for d in (0.5, 1.5, 2.5, 3.49)], r["box"][0]
        assert r["features"][2] == [i + k / 16 for k in range(8)]
        assert meta["img_to_row_map"][f"img{i}"] == [x["imgid"] for x in table.to_pylist()].index(f"img{i}")
"""


def test_pipeline_one_rank_ragged(tmp_path):
    ns = {}
    sys.path.insert(0, ROOT)
    exec(_FAKE, ns)
    from vltk_amd.pipeline import ExtractionPipeline
    path = str(tmp_path / "vg" / "frcnn" / "train.arrow")
    pipe = ExtractionPipeline(ns["FakeModel"](), ns["fake_preprocess"], path, batch_size=3, visual_dim=8, dataset="vg")
    out = pipe.run(ns["items"](0, 7))
    assert out == path and pipe.images_done == 7
    ns["check"](path, 7)


def test_pipeline_writer_error_surfaces(tmp_path):
    ns = {}
    sys.path.insert(0, ROOT)
    exec(_FAKE, ns)
    from vltk_amd.pipeline import ExtractionPipeline
    pipe = ExtractionPipeline(ns["FakeModel"](), ns["fake_preprocess"], str(tmp_path / "t.arrow"), batch_size=2, visual_dim=8)
    calls = []

    def failing(*a):                      # the second batch cannot be written: the error must reach the caller of run()
        calls.append(1)
        if len(calls) == 2:
            raise ValueError("no space left on device")
    pipe.writer.write_batch = failing
    with pytest.raises(ValueError, match="no space left"):
        pipe.run(ns["items"](0, 9))


def test_pipeline_duplicate_ids_are_skipped(tmp_path):
    """The reference prints "skipping ..." for a repeated imgid (extraction.py:183-185); here the row really is skipped
    and the run continues."""
    ns = {}
    sys.path.insert(0, ROOT)
    exec(_FAKE, ns)
    from vltk_amd.pipeline import ExtractionPipeline
    path = str(tmp_path / "t.arrow")
    pipe = ExtractionPipeline(ns["FakeModel"](), ns["fake_preprocess"], path, batch_size=2, visual_dim=8)
    dup = ns["items"](0, 3) + ns["items"](1, 2) + ns["items"](3, 9)
    assert pipe.run(dup) == path
    ns["check"](path, 9)


def test_pipeline_fewer_images_than_ranks_raises_before_any_collective(tmp_path):
    ns = {}
    sys.path.insert(0, ROOT)
    exec(_FAKE, ns)
    from vltk_amd.pipeline import ExtractionPipeline
    pipe = ExtractionPipeline(ns["FakeModel"](), ns["fake_preprocess"], str(tmp_path / "t.arrow"), batch_size=2, visual_dim=8)
    pipe.world = 4                        # as rank 0 of four
    with pytest.raises(ValueError, match="every rank needs at least one image"):
        pipe.run(ns["items"](0, 1), n_items=3)


_WORKER = _FAKE + r"""
import os, sys, torch.distributed as dist
from vltk_amd.parallel import shard_indices
from vltk_amd.pipeline import ExtractionPipeline
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
r, N = dist.get_rank(), 11
lo, hi = shard_indices(N, r, 2)
path = sys.argv[4]
pipe = ExtractionPipeline(FakeModel(), fake_preprocess, path, batch_size=2, visual_dim=8, dataset="vg")
pipe.set_global_ids([f"img{i}" for i in range(N)])
out = pipe.run(items(lo, hi), n_items=N)
dist.barrier()
if r == 0:
    assert out == path
    check(path, N)
else:
    assert out is None
dist.destroy_process_group()
print("ok", r)
"""


def test_pipeline_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text("import sys\nsys.path.insert(0, sys.argv[1])\n" + _WORKER)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), str(tmp_path / "train.arrow")],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


@pytest.mark.gpu
def test_pipeline_matches_step_by_step_on_gpu(tmp_path):
    """Raw uint8 images of three sizes -> pipeline (batch 4, ragged tail) vs the same steps called one batch at a time."""
    from vltk_amd import FRCNN, make_state_dict
    from vltk_amd.config import Config, vg_c4_config_dict
    from vltk_amd.extraction import ExtractionWriter, load_extraction
    from vltk_amd.pipeline import ExtractionPipeline
    from vltk_amd.preprocess import Preprocess
    d = vg_c4_config_dict(post_nms_topk=30, detections=12)
    d["input"]["min_size_test"], d["input"]["max_size_test"] = 160, 256
    cfg = Config(d)
    model = FRCNN(cfg, precision="fp16").load_state_dict(make_state_dict(cfg, seed=1234)).eval()
    pre = Preprocess(cfg)
    g = np.random.Generator(np.random.PCG64(5))
    shapes = [(120, 160), (96, 200), (150, 110)]
    items = [(f"id{i}", g.integers(0, 256, shapes[i % 3] + (3,), dtype=np.uint8)) for i in range(10)]
    p1 = str(tmp_path / "a" / "train.arrow")
    pipe = ExtractionPipeline(model, pre, p1, batch_size=4, dataset="synthetic", model_config=cfg.to_dict())
    assert pipe.run(items) == p1
    # step by step, no threads, no pinned ring
    p2 = str(tmp_path / "b" / "train.arrow")
    with ExtractionWriter(p2, 12, 2048, dataset="synthetic", model_config=cfg.to_dict()) as w:
        for k in range(0, 10, 4):
            part = items[k:k + 4]
            raws = [torch.from_numpy(r).cuda() for _, r in part]
            while len(raws) < 4:
                raws.append(raws[-1])
            _, images, sizes, scales = pre(raws, list(range(4)))
            model(images, sizes, scales_yx=scales)
            blk = {kk: v.cpu().numpy() for kk, v in model.forward_padded().items()}
            n = len(part)
            w.write_batch([i for i, _ in part], blk["obj_ids"][:n].astype(np.float32), blk["attr_ids"][:n].astype(np.float32),
                          np.round(blk["boxes"][:n]), blk["roi_features"][:n])
    t1, m1 = load_extraction(p1)
    t2, m2 = load_extraction(p2)
    assert t1.num_rows == 10 and t1.equals(t2) and m1["img_to_row_map"] == m2["img_to_row_map"]
    assert np.isfinite(np.asarray(t1.column("features")[0].as_py())).all()


_GPU_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from vltk_amd import FRCNN, make_state_dict
from vltk_amd.config import Config, vg_c4_config_dict
from vltk_amd.parallel import shard_indices
from vltk_amd.pipeline import ExtractionPipeline
from vltk_amd.preprocess import Preprocess
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
if world > 1:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
d = vg_c4_config_dict(post_nms_topk=30, detections=12)
d["input"]["min_size_test"], d["input"]["max_size_test"] = 160, 256
cfg = Config(d)
model = FRCNN(cfg, precision="fp16").load_state_dict(make_state_dict(cfg, seed=1234)).eval()
g = np.random.Generator(np.random.PCG64(5))
shapes = [(120, 160), (96, 200), (150, 110)]
N = 11
items = [(f"id{i}", g.integers(0, 256, shapes[i % 3] + (3,), dtype=np.uint8)) for i in range(N)]
lo, hi = shard_indices(N, rank, world)
pipe = ExtractionPipeline(model, Preprocess(cfg), out, batch_size=3, dataset="synthetic")
pipe.set_global_ids([i for i, _ in items])
pipe.run(items[lo:hi], n_items=N)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
print("ok", rank)
"""


@pytest.mark.gpu
def test_pipeline_two_ranks_on_one_gpu_match_one_rank(tmp_path):
    """The real model under the multi-rank loop: two processes share the GPU (gloo carries the flat output block between
    them; RCCL needs one GPU per rank), rank 0's Arrow file holds the same rows as a one-rank run of the same images.
    An image's outputs do not depend on what else is in its batch, so the different batch composition must not matter."""
    from vltk_amd.extraction import load_extraction
    script = tmp_path / "w.py"
    script.write_text(_GPU_WORKER)
    port = str(33500 + os.getpid() % 2000)
    one = subprocess.run([sys.executable, str(script), ROOT, "0", "1", port, str(tmp_path / "one.arrow")],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port, str(tmp_path / "two.arrow")],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    t1, m1 = load_extraction(str(tmp_path / "one.arrow"))
    t2, m2 = load_extraction(str(tmp_path / "two.arrow"))
    assert t1.num_rows == t2.num_rows == 11
    r1 = {r["imgid"]: r for r in t1.to_pylist()}
    r2 = {r["imgid"]: r for r in t2.to_pylist()}
    assert sorted(r1) == sorted(r2)
    for k in r1:
        assert r1[k] == r2[k], k
