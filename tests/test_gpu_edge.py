"""-m gpu: the reference's own edge branches, one run each, against the oracle.

(a) an image that keeps ZERO proposals: the size filter of find_top_rpn_proposals (frcnn.py:371) drops every box of one
    image, the head then runs the reference's empty-tensor path for it (frcnn.py:799-809, `_NewEmptyTensorOp` :464-473) --
    here: a count of 0, zero rows, no fault, and the other images of the batch bit-for-bit what they are without it;
(b) a non-finite image: `_clip_box` asserts `isfinite(box).all()` on the host (frcnn.py:148) -> AssertionError from
    `model(...)` and from `forward_async().wait()`, the handle usable afterwards;
(c) BASELINE configs[0] literally (tests/frcnn_test.py:15-31): 4 x 800x1333, 36 detections, nms_thresh [0.5, 1.0, 0.1],
    `padding="max_detections", return_tensors="np"`, one image checked against the oracle.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.frcnn_oracle import FRCNNOracle            # noqa: E402
from vltk_amd import FRCNN, make_state_dict, synthetic_images, vg_c4_config   # noqa: E402

import gpu_util as G                                   # noqa: E402
from test_gpu_e2e import stage_chain_check             # noqa: E402


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_image_with_zero_proposals(precision):
    """MIN_SIZE = 12 and one image whose `image_shapes` row is (10, 10): every proposal of that image is clipped to at most
    10 x 10 (frcnn.py:147-153) and fails `w > 12 & h > 12` (:156-160, :371); the others keep theirs."""
    cfg = vg_c4_config(depth=50, post_nms_topk=24, detections=8, overrides=[("proposal_generator", "min_size", 12)])
    sd = make_state_dict(cfg, seed=5)
    x = torch.from_numpy(synthetic_images(3, 128, 160, seed=3))
    shapes = [[128, 160], [10, 10], [112, 150]]
    m = FRCNN(cfg, precision=precision).load_state_dict(sd).eval()
    out = m(x, torch.tensor(shapes))
    counts = m.get_stage("proposal_counts").cpu().tolist()
    assert counts[1] == 0 and counts[0] > 0 and counts[2] > 0, counts
    assert out["preds_per_image"].tolist()[1] == 0
    for k in ("obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "roi_features"):
        assert out[k][1].shape[0] == 0, k
    pad = m.forward_padded()
    for k in ("obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "roi_features"):
        assert torch.count_nonzero(pad[k][1]) == 0, k                      # zero rows, as every row past a count
    # every stage against the oracle fed with the GPU's own upstream tensors, the empty image included
    oracle = FRCNNOracle(cfg, sd, emulate=None if precision == "fp32" else "fp16")
    stage_chain_check(m, out, oracle, shapes, tol=1e-4 if precision == "fp32" else 1e-3)
    ref = oracle.forward(x, shapes)
    assert ref["preds_per_image"].tolist()[1] == 0
    # the other images do not notice: same batch with image 1 at full size
    full = {k: v.clone() for k, v in pad.items()}
    m(x, torch.tensor([[128, 160], [128, 160], [112, 150]]))
    other = m.forward_padded()
    for k in full:
        assert torch.equal(full[k][0], other[k][0]) and torch.equal(full[k][2], other[k][2]), k
    assert int(other["preds_per_image"][1]) > 0


def test_all_images_with_zero_proposals():
    """The whole batch empty (MIN_SIZE larger than the images): zero counts everywhere, no fault, handle reusable."""
    cfg = vg_c4_config(depth=50, post_nms_topk=16, detections=4, overrides=[("proposal_generator", "min_size", 1000)])
    sd = make_state_dict(cfg, seed=5)
    x = torch.from_numpy(synthetic_images(2, 96, 128, seed=3))
    m = FRCNN(cfg, precision="fp16").load_state_dict(sd).eval()
    out = m(x, torch.tensor([[96, 128], [96, 128]]))
    assert out["preds_per_image"].tolist() == [0, 0]
    assert m.get_stage("proposal_counts").cpu().tolist() == [0, 0]
    out = m(x, torch.tensor([[96, 128], [96, 128]]), padding="max_detections", return_tensors="np")
    assert out["roi_features"].shape == (2, 4, 2048) and not out["roi_features"].any()


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_nonfinite_image_raises_assertion_and_handle_survives(precision, bad):
    """frcnn.py:148: `assert torch.isfinite(tensor).all(), "Box tensor contains infinite or NaN!"` -- through vk_forward
    (model(...)) and through vk_forward_begin / _end (forward_async().wait()); the next forward on the same handle is
    bit-identical to the one before."""
    cfg = vg_c4_config(depth=50, post_nms_topk=16, detections=6)
    sd = make_state_dict(cfg, seed=9)
    x = torch.from_numpy(synthetic_images(2, 96, 128, seed=1))
    shapes = torch.tensor([[96, 128], [96, 128]])
    m = FRCNN(cfg, precision=precision).load_state_dict(sd).eval()
    m(x, shapes)
    good = {k: v.clone() for k, v in m.forward_padded().items()}
    xb = x.clone()
    xb[1, :, 40:60, 50:70] = bad
    with pytest.raises(AssertionError, match="infinite or NaN"):
        m(xb, shapes)
    h = m.forward_async(xb, shapes)
    with pytest.raises(AssertionError, match="infinite or NaN"):
        h.wait()
    # two in flight, the bad one first: its assertion does not poison the good one behind it
    h1, h2 = m.forward_async(xb, shapes), m.forward_async(x, shapes)
    with pytest.raises(AssertionError):
        h1.wait()
    h2.wait()
    for k in good:
        assert torch.equal(good[k], m.forward_padded()[k]), k
    m(x, shapes)
    for k in good:
        assert torch.equal(good[k], m.forward_padded()[k]), k


def test_configs0_call_shape_full_size():
    """BASELINE configs[0] = tests/frcnn_test.py:15-31 with synthetic inputs: ResNet-101-C4, 4 images of 800x1333,
    min = max = 36 detections, nms_thresh [0.5, 1.0, 0.1], score_thresh 0.2, padded numpy outputs; image 0 against the
    oracle (strict fp32: identical detections, tensors <= 1e-3)."""
    cfg = vg_c4_config(post_nms_topk=300, detections=36)
    sd = make_state_dict(cfg, seed=1234)
    x = torch.from_numpy(synthetic_images(4, 800, 1333, seed=0xF2C))
    sizes = torch.tensor([[800, 1333]] * 4)
    scales = torch.tensor([[1.25, 1.25], [1.0, 1.0], [0.8, 0.8], [1.5, 0.75]])
    outs = {}
    for precision in ("fp32", "fp16"):
        m = FRCNN(cfg, precision=precision).load_state_dict(sd).eval()
        m.roi_outputs.nms_thresh = [0.5, 1.0, 0.1]          # tests/frcnn_test.py:16-19
        m.roi_outputs.score_thresh = 0.2
        m.roi_outputs.min_detections = 36
        m.roi_outputs.max_detections = 36
        out = m(x, sizes, scales_yx=scales, padding="max_detections", max_detections=cfg.max_detections, return_tensors="np")
        assert list(out) == ["obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "sizes", "preds_per_image",
                             "roi_features", "normalized_boxes"]
        assert out["roi_features"].shape == (4, 36, 2048) and out["roi_features"].dtype == np.float32
        assert out["boxes"].shape == (4, 36, 4) and out["normalized_boxes"].shape == (4, 36, 4)
        assert out["obj_ids"].shape == (4, 36) and out["obj_ids"].dtype == np.int64
        assert out["obj_probs"].shape == (4, 36) and out["attr_probs"].shape == (4, 36)
        assert out["preds_per_image"].shape == (4,) and out["sizes"].tolist() == [[800, 1333]] * 4
        assert all(1 <= int(c) <= 36 for c in out["preds_per_image"])
        outs[precision] = out
        del m
    torch.set_num_threads(16)
    oracle = FRCNNOracle(cfg, sd)
    oracle.nms_thresh, oracle.score_thresh, oracle.min_detections, oracle.max_detections = [0.5, 1.0, 0.1], 0.2, 36, 36
    ref = oracle.forward(x[:1], [[800, 1333]], scales_yx=scales[:1])
    out = outs["fp32"]
    c = int(ref["preds_per_image"][0])
    assert int(out["preds_per_image"][0]) == c
    np.testing.assert_array_equal(out["obj_ids"][0, :c], ref["obj_ids"][0].numpy())
    np.testing.assert_array_equal(out["attr_ids"][0, :c], ref["attr_ids"][0].numpy())
    for k in ("roi_features", "boxes", "obj_probs", "attr_probs"):
        e = G.rel_err(out[k][0, :c], ref[k][0])
        print(f"\n[configs[0], fp32 strict vs oracle, image 0] {k} rel err {e:.3e}")
        assert e <= 1e-3, k
    assert not out["roi_features"][0, c:].any()
    # the benched mode on the same call: detections that coincide with the strict run's (by box within 1 px and class)
    o16 = outs["fp16"]
    c16 = int(o16["preds_per_image"][0])
    matched = 0
    for i in range(c16):
        d = np.abs(out["boxes"][0, :c] - o16["boxes"][0, i]).max(axis=1)
        j = int(d.argmin())
        matched += int(d[j] <= 1.0 and int(o16["obj_ids"][0, i]) == int(out["obj_ids"][0, j]))
    print(f"[configs[0], fp16 vs fp32 strict, image 0] {matched} of {c16} detections matched by box (1 px) and class")
    assert matched >= c16 // 2


def test_head_chunk_beyond_4gib_keeps_the_separate_mean():
    """ResNeXt 32x8d head (2048 mid channels) over 9600 RoIs in ONE chunk: the conv3 input of the last block is 7.7 GB, past
    the 32-bit byte offsets of the fused conv3 + spatial-mean kernels.  The plan must fall back to the separate mean kernel
    (it used to fail the forward with EINVAL: found by `bench.py --arch x152` at batch 32) and give the same detections and
    features as two chunks of 4800 RoIs, which take the fused form."""
    cfg = vg_c4_config(depth=50, num_groups=32, width_per_group=8, post_nms_topk=800, detections=36)      # 12 images x 800 = 9600 RoIs
    sd = make_state_dict(cfg, seed=77)
    x = torch.from_numpy(synthetic_images(12, 320, 416, seed=5))
    sizes = torch.tensor([[320, 416]] * 12)
    m = FRCNN(cfg, precision="fp16").load_state_dict(sd).eval()
    outs = []
    for chunk in (4800, 9600):
        m.set_option("head_chunk", chunk)
        m(x, sizes)
        outs.append({k: v.cpu() for k, v in m.forward_padded().items()})
    a, b = outs
    for k in ("obj_ids", "attr_ids", "preds_per_image"):
        assert torch.equal(a[k], b[k]), k
    # fused: exact sums of the f16 outputs rounded once; separate kernel: fp32 running mean of the same f16 values
    ef, eb = G.rel_err(b["roi_features"], a["roi_features"]), G.rel_err(b["boxes"], a["boxes"])
    print(f"\n[one 9600-RoI chunk vs two of 4800] roi_features {ef:.2e} boxes {eb:.2e}")
    assert ef <= 1e-5 and eb <= 1e-5
