"""-m gpu: the FPN detector end to end (vltk_amd/frcnn_fpn.py: ResNet-FPN -> multi-level RPN -> RoIAlign -> 2-FC box head ->
the reference's box predictor / ROIOutputs) against oracle/fpn_oracle.py FPNDetectorOracle.

PARITY UNPINNED vs the reference end to end: it has no FPN model.  Its level-agnostic pieces are the same
code paths the C4 tests pin (bottleneck, RPN head, proposals over several levels, predictor, ROIOutputs); the neck and
RoIAlign follow detectron2 / torchvision as restated in the oracle.  Stage chaining as in test_gpu_e2e.py: every stage is
compared with the oracle fed the GPU's own upstream tensors, then the free-running outputs are compared."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.fpn_oracle import FPNDetectorOracle        # noqa: E402
from vltk_amd import FRCNN, fpn_config, make_state_dict, synthetic_images   # noqa: E402
from vltk_amd.frcnn_fpn import FRCNNFPN                # noqa: E402

import gpu_util as G                                   # noqa: E402


def nchw(t):
    return t.float().permute(0, 3, 1, 2).contiguous().cpu()


def build(precision, depth=50, seed=3, **kw):
    # anchors twice the usual size so that P3's proposals reach the second pooling level (see tests/test_fpn_oracle.py)
    cfg = fpn_config(depth=depth, post_nms_topk=200, pre_nms_topk=300, detections=10,
                     overrides=[("anchor_generator", "sizes", [[64], [128], [256], [512], [1024]])], **kw)
    sd = make_state_dict(cfg, seed=seed)
    m = FRCNN(cfg, precision=precision).load_state_dict(sd).eval()
    assert isinstance(m, FRCNNFPN)
    return cfg, sd, m


def inputs():
    x = synthetic_images(2, 320, 448, seed=5)
    shapes = [[320, 448], [300, 400]]
    x[1, :, 300:, :] = 0
    x[1, :, :, 400:] = 0
    return torch.from_numpy(x), shapes


def stage_chain(m, out, o, x, shapes, tol):
    N, R = len(shapes), m.config.RPN.POST_NMS_TOPK_TEST
    A = m.A
    # bottom-up + neck, free-running (few layers deep at depth 50)
    st_ref = o.backbone(x)
    c = {k: nchw(m.get_stage(k)) for k in ("res2", "res3", "res4", "res5")}
    for k in c:
        assert G.rel_err(c[k], st_ref[k]) <= (5e-3 if o.emulate else 1e-4), k
    pyr_ref = o.neck(c)                                      # the oracle's neck on the GPU's own C2..C5
    pyr = [nchw(m.get_stage(f"p{i}")) for i in range(2, 7)]
    for a, b in zip(pyr, pyr_ref):
        assert a.shape == b.shape and G.rel_err(a, b) <= tol
    # RPN head per level on the GPU's pyramid
    heads = []
    for i, p in enumerate(pyr):
        r = m.get_stage(f"rpn_out{i + 2}").cpu()
        obj = r[..., :A].permute(0, 3, 1, 2).contiguous()
        dlt = r[..., A:5 * A].permute(0, 3, 1, 2).contiguous()
        o_obj, o_dlt = o.rpn_head(p)
        assert G.rel_err(obj, o_obj) <= tol and G.rel_err(dlt, o_dlt) <= tol, i
        heads.append((obj, dlt))
    # proposals: indices bit-exact (identical logits), boxes to exp() rounding
    props = o.proposals(heads, shapes)
    pb, pl, pc = (m.get_stage(k).cpu() for k in ("proposal_boxes", "proposal_logits", "proposal_counts"))
    for i in range(N):
        cnt = int(pc[i])
        assert cnt == len(props[i][0])
        np.testing.assert_array_equal(pl[i, :cnt].numpy(), props[i][1].numpy())
        assert G.rel_err(pb[i, :cnt], props[i][0]) <= 2e-6
    boxes = [pb[i, :int(pc[i])] for i in range(N)]
    rows = np.concatenate([np.arange(int(pc[i])) + i * R for i in range(N)])
    pooled_ref, lv_ref = o.box_pool(pyr, boxes)
    np.testing.assert_array_equal(m.get_stage("levels").cpu().numpy()[rows], lv_ref.numpy())
    pooled = m.get_stage("pooled").float().permute(0, 3, 1, 2).cpu()[rows]
    assert G.rel_err(pooled, pooled_ref) <= (2e-3 if o.emulate else 1e-5)
    feat = m.get_stage("box_features").cpu()[rows]
    assert G.rel_err(feat, o.box_head(pooled)) <= tol
    # predictor + outputs on the GPU's features (the reference's FastRCNNOutputLayers / ROIOutputs semantics)
    s_ref, a_ref, d_ref = o.predictor(feat)
    C1, A1 = s_ref.shape[1], a_ref.shape[1]
    s = m.get_stage("obj_logits").cpu()[rows][:, :C1]
    assert G.rel_err(s, s_ref) <= tol
    same = s.argmax(-1) == s_ref.argmax(-1)
    a = m.get_stage("attr_logits").cpu()[rows][:, :A1]
    assert G.rel_err(a[same], a_ref[same]) <= tol
    cls = s[:, :-1].argmax(-1)
    chosen = m.get_stage("chosen_deltas").cpu()[rows]
    assert G.rel_err(chosen, d_ref.view(len(rows), -1, 4)[torch.arange(len(rows)), cls]) <= max(tol, 1e-5)
    full = torch.zeros(len(rows), d_ref.shape[1])
    full.view(len(rows), -1, 4)[torch.arange(len(rows)), cls] = chosen
    res = o.roi_outputs(s, a, full, boxes, feat, shapes)
    for i, (mb, c_, ms, aid, ap, ft, ids) in enumerate(res):
        assert int(out["preds_per_image"][i]) == len(c_)
        np.testing.assert_array_equal(out["obj_ids"][i].cpu().numpy(), c_.numpy())
        np.testing.assert_array_equal(out["attr_ids"][i].cpu().numpy(), aid.numpy())
        np.testing.assert_array_equal(out["roi_features"][i].cpu().numpy(), ft.numpy())
        assert G.rel_err(out["obj_probs"][i].cpu(), ms) <= 2e-6 and G.rel_err(out["boxes"][i].cpu(), mb) <= 2e-6


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("fp16", 1e-3)])
def test_fpn_detector_vs_oracle(precision, tol):
    cfg, sd, m = build(precision)
    x, shapes = inputs()
    out = m(x, torch.tensor(shapes))
    o = FPNDetectorOracle(cfg, sd, emulate=None if precision == "fp32" else "fp16")
    stage_chain(m, out, o, x, shapes, tol)
    lv = m.get_stage("levels").cpu()
    assert len(torch.unique(lv)) >= 2, lv.bincount()            # more than one pyramid level really is used
    if precision == "fp32":        # free-running: identical detections, outputs at 1e-3
        ref = o.forward(x, shapes)
        np.testing.assert_array_equal(out["preds_per_image"].numpy(), ref["preds_per_image"].numpy())
        for i in range(len(shapes)):
            np.testing.assert_array_equal(out["obj_ids"][i].cpu().numpy(), ref["obj_ids"][i].numpy())
            for k in ("roi_features", "boxes", "obj_probs", "attr_probs"):
                assert G.rel_err(out[k][i].cpu(), ref[k][i]) <= 1e-3, (k, i)
        assert out["roi_features"][0].shape[1] == 1024


def test_fpn_detector_call_surface_and_invariances():
    cfg, sd, m = build("fp16")
    x, shapes = inputs()
    sh = torch.tensor(shapes)
    m.roi_outputs.nms_thresh = [0.5, 1.0, 0.1]                  # tests/frcnn_test.py:16-19
    m.roi_outputs.min_detections = m.roi_outputs.max_detections = 10
    out = m(x, sh, scales_yx=torch.tensor([[1.25, 1.25], [2.0, 2.0]]), padding="max_detections", max_detections=10, return_tensors="np")
    assert list(out.keys()) == ["obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "sizes", "preds_per_image",
                                "roi_features", "normalized_boxes"]
    assert out["roi_features"].shape == (2, 10, 1024) and (out["preds_per_image"] == 10).all()
    ref = {k: v.clone() for k, v in m.forward_padded().items()}
    m(x.flip(0), sh.flip(0), scales_yx=torch.tensor([[2.0, 2.0], [1.25, 1.25]]))          # images are independent
    for k, v in ref.items():
        assert torch.equal(v, m.forward_padded()[k].flip(0)), k
    p = m.forward_async(x, sh)
    assert torch.equal(p.wait_raw()["obj_ids"], m.forward_padded()["obj_ids"])
    with pytest.raises(NotImplementedError):
        m.train()(x, sh)
    m.eval()
    bad = dict(sd)
    bad.pop("roi_heads.box_head.fc2.bias")
    with pytest.raises(OSError, match="missing key"):
        FRCNN(cfg).load_state_dict(bad)


def test_fpn_detector_from_pretrained_round_trip(tmp_path):
    cfg, sd, m = build("fp16")
    x, shapes = inputs()
    m(x, torch.tensor(shapes))
    ref = m.forward_padded()["roi_features"].clone()
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, str(tmp_path / "pytorch_model.bin"))
    cfg.dump_yaml(str(tmp_path / "config.yaml"))
    again = FRCNN.from_pretrained(str(tmp_path), precision="fp16")
    assert isinstance(again, FRCNNFPN)
    again(x, torch.tensor(shapes))
    assert torch.equal(ref, again.forward_padded()["roi_features"])
