"""Pin the oracle (oracle/frcnn_oracle.py) against the golden vectors that
tools/gen_golden.py produced from the reference's own module
(/root/reference/vltk/modeling/frcnn.py loaded under stubs).

fp32, same ATen provider: the restatement must agree to float rounding
(tolerance 1e-5 relative to the tensor's max magnitude; indices exact).
"""
import os

import numpy as np
import pytest
import torch

from oracle.frcnn_oracle import FRCNNOracle
from vltk_amd.config import Config, vg_c4_config, vg_c4_config_dict
from vltk_amd.weights import make_state_dict, synthetic_images

TOL = 1e-5


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
    assert err <= tol, err


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "kat_ops.npz"))


def sub_sd(kat, tag, prefix=""):
    p = tag + "/sd/"
    return {prefix + k[len(p):]: torch.from_numpy(kat[k]) for k in kat.files if k.startswith(p)}


def tiny_oracle(sd, **over):
    d = vg_c4_config_dict()
    for k, v in over.items():
        sec, key = k.split("__")
        d[sec][key] = v
    return FRCNNOracle(Config(d), sd)


@pytest.mark.parametrize("tag,caffe", [("stem_caffe", True), ("stem_pad1", False)])
def test_stem(kat, tag, caffe):
    o = tiny_oracle(sub_sd(kat, tag, "backbone.stem."), model__max_pool=caffe)
    y = o.stem(torch.from_numpy(kat[tag + "/x"]))
    close(y, kat[tag + "/y"])


@pytest.mark.parametrize("tag", ["blk_s2_in1x1", "blk_s2_in3x3", "blk_identity", "blk_dil2", "blk_groups"])
def test_bottleneck(kat, tag):
    cin, cout, mid, stride, groups, s1x1, dil = kat[tag + "/args"].tolist()
    o = tiny_oracle(sub_sd(kat, tag, "blk."), resnets__num_groups=groups, resnets__stride_in_1x1=bool(s1x1))
    y = o.bottleneck(torch.from_numpy(kat[tag + "/x"]), "blk", stride, dilation=dil)
    close(y, kat[tag + "/y"])


def test_anchors(kat):
    from vltk_amd.weights import cell_anchors
    cell = cell_anchors([32, 64, 128, 256, 512], [0.5, 1.0, 2.0])
    np.testing.assert_array_equal(cell, kat["anchors/cell"])
    # SURVEY.md §8a row 8 known answers
    np.testing.assert_allclose(cell[0], [-22.6274, -11.3137, 22.6274, 11.3137], atol=1e-4)
    np.testing.assert_allclose(cell[14], [-181.0193, -362.0387, 181.0193, 362.0387], atol=1e-4)
    o = tiny_oracle({"proposal_generator.anchor_generator.cell_anchors.0": torch.from_numpy(cell)})
    for hw in ((3, 4), (10, 14)):
        a = o.grid_anchors(*hw)
        np.testing.assert_array_equal(a.numpy(), kat[f"anchors/grid_{hw[0]}x{hw[1]}"])
    a = o.grid_anchors(3, 4).view(3, 4, 15, 4)
    np.testing.assert_allclose(a[0, 1, 0], [-6.6274, -11.3137, 38.6274, 11.3137], atol=1e-4)
    np.testing.assert_allclose(a[1, 0, 0], [-22.6274, 4.6863, 22.6274, 27.3137], atol=1e-4)


@pytest.mark.parametrize("tag,w", [("deltas_rpn", (1.0, 1.0, 1.0, 1.0)), ("deltas_roi", (10.0, 10.0, 5.0, 5.0))])
def test_apply_deltas(kat, tag, w):
    y = FRCNNOracle.apply_deltas(torch.from_numpy(kat[tag + "/deltas"]), torch.from_numpy(kat[tag + "/boxes"]), w)
    np.testing.assert_array_equal(y.numpy(), kat[tag + "/y"])
    # SURVEY.md §8a row 10 known answer (clamped dw)
    np.testing.assert_allclose(y[0, :4].numpy(), [-490.40002, -8.3897696, 509.60004, 17.989769], rtol=1e-6)


def test_rpn(kat):
    sd = sub_sd(kat, "rpn", "proposal_generator.")
    o = tiny_oracle(sd, proposal_generator__hidden_channels=32, rpn__pre_nms_topk_test=400,
                    rpn__post_nms_topk_test=40)
    feat = torch.from_numpy(kat["rpn/feat"])
    obj, dlt = o.rpn_head(feat)
    close(obj, kat["rpn/objectness"])
    close(dlt, kat["rpn/deltas"])
    # stage-level: same logits in -> identical proposals out (indices bit-exact)
    res = o.rpn_proposals(torch.from_numpy(kat["rpn/objectness"]), torch.from_numpy(kat["rpn/deltas"]),
                          kat["rpn/shapes"].tolist())
    for i, (b, l) in enumerate(res):
        np.testing.assert_array_equal(b.numpy(), kat[f"rpn/boxes_{i}"])
        np.testing.assert_array_equal(l.numpy(), kat[f"rpn/logits_{i}"])


def test_predictor(kat):
    o = tiny_oracle(sub_sd(kat, "pred", "roi_heads.box_predictor."))
    s, a, d = o.predictor(torch.from_numpy(kat["pred/x"]))
    close(s, kat["pred/scores"])
    close(a, kat["pred/attr"])
    close(d, kat["pred/deltas"])


@pytest.mark.parametrize("tag", ["roiout", "roiout_scaled"])
def test_roi_outputs(kat, tag):
    d = vg_c4_config_dict()
    d["min_detections"], d["max_detections"] = 6, 8
    o = FRCNNOracle(Config(d), {})
    o.nms_thresh = kat["roiout/nms_thresh"].tolist()
    props = [torch.from_numpy(kat[f"roiout/props_{i}"]) for i in range(2)]
    scales = torch.from_numpy(kat["roiout/scales"]) if tag == "roiout_scaled" else None
    res = o.roi_outputs(torch.from_numpy(kat["roiout/obj_logits"]), torch.from_numpy(kat["roiout/attr_logits"]),
                        torch.from_numpy(kat["roiout/box_deltas"]), props, torch.from_numpy(kat["roiout/feats_in"]),
                        kat["roiout/sizes"].tolist(), scales)
    for i, (mb, cls, ms, aid, ap, ft, ids) in enumerate(res):
        np.testing.assert_array_equal(cls.numpy(), kat[f"{tag}/classes_{i}"])
        np.testing.assert_array_equal(aid.numpy(), kat[f"{tag}/attrs_{i}"])
        np.testing.assert_array_equal(mb.numpy(), kat[f"{tag}/boxes_{i}"])
        np.testing.assert_array_equal(ms.numpy(), kat[f"{tag}/probs_{i}"])
        np.testing.assert_array_equal(ap.numpy(), kat[f"{tag}/attr_probs_{i}"])
        np.testing.assert_array_equal(ft.numpy(), kat[f"{tag}/feats_{i}"])
        assert 6 <= len(cls) <= 8 or o.nms_thresh[-1] == 0.9


def test_e2e_r101_small(golden_dir):
    """Whole FRCNN.forward (ResNet-101-C4, seeded weights regenerated from the seed) vs the reference."""
    g = np.load(os.path.join(golden_dir, "e2e_r101_small.npz"))
    n, h, w = g["nhw"].tolist()
    cfg = vg_c4_config(depth=int(g["depth"]), post_nms_topk=int(g["post_topk"]), detections=int(g["det"]))
    sd = make_state_dict(cfg, seed=int(g["weights_seed"]))
    x = torch.from_numpy(synthetic_images(n, h, w, seed=int(g["images_seed"])))
    shapes = g["shapes"].tolist()
    for i, (hh, ww) in enumerate(shapes):
        x[i, :, hh:, :] = 0
        x[i, :, :, ww:] = 0
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    out, st = FRCNNOracle(cfg, sd).forward(x, shapes, return_stages=True)
    close(st["res4"], g["res4"])
    close(st["rpn_objectness"], g["rpn_objectness"])
    close(st["rpn_deltas"], g["rpn_deltas"])
    for i in range(n):
        close(st["proposal_boxes"][i], g[f"proposal_boxes_{i}"])
        close(st["proposal_logits"][i], g[f"proposal_logits_{i}"])
    close(st["pooled"][:, :8], g["roipool_c0_7"])
    close(st["feature_pooled"], g["feature_pooled"])
    close(st["obj_logits"], g["obj_logits"])
    close(st["attr_logits"], g["attr_logits"])
    close(st["box_deltas"][:, :256], g["box_deltas_head"])
    np.testing.assert_array_equal(out["preds_per_image"].numpy(), g["preds_per_image"])
    for i in range(n):
        np.testing.assert_array_equal(out["obj_ids"][i].numpy(), g[f"obj_ids_{i}"])
        np.testing.assert_array_equal(out["attr_ids"][i].numpy(), g[f"attr_ids_{i}"])
        close(out["obj_probs"][i], g[f"obj_probs_{i}"])
        close(out["attr_probs"][i], g[f"attr_probs_{i}"])
        close(out["boxes"][i], g[f"boxes_{i}"])
        close(out["roi_features"][i], g[f"roi_features_{i}"])


VARIANTS = {   # tools/gen_golden.py VARIANTS: configuration switches of the reference the main fixture does not take
    "resnext50_8x8d": [("resnets", "num_groups", 8), ("resnets", "width_per_group", 8)],
    "r50_halve": [("roi_box_head", "res5halve", True)],
    "r50_halve_s3x3": [("roi_box_head", "res5halve", True), ("resnets", "stride_in_1x1", False)],
}


VARIANTS_X152 = {   # tools/gen_golden.py VARIANTS_X152 (BASELINE configs[3] at its real depth / group count), e2e_x152.npz
    "resnext152_32x8d": [("resnets", "depth", 152), ("resnets", "num_groups", 32), ("resnets", "width_per_group", 8)],
}


def variant_inputs(g, tag):
    n, h, w = g["nhw"].tolist()
    cfg = vg_c4_config(depth=50, post_nms_topk=16, detections=6, overrides={**VARIANTS, **VARIANTS_X152}[tag])
    sd = make_state_dict(cfg, seed=int(g["seed"]))
    x = torch.from_numpy(synthetic_images(n, h, w, seed=int(g["seed"])))
    shapes = g["shapes"].tolist()
    for i, (hh, ww) in enumerate(shapes):
        x[i, :, hh:, :] = 0
        x[i, :, :, ww:] = 0
    return cfg, sd, x, shapes


@pytest.mark.parametrize("tag", list(VARIANTS) + list(VARIANTS_X152))
def test_e2e_config_variants(golden_dir, tag):
    """ResNeXt groups / RES5HALVE / stride in the 3x3 (frcnn.py:217-219, 932, 942-952, 1345-1355), and ResNeXt-152
    32x8d at full depth (BASELINE configs[3]): the oracle against the reference's own end-to-end output for each."""
    g = np.load(os.path.join(golden_dir, "e2e_x152.npz" if tag in VARIANTS_X152 else "e2e_variants.npz"))
    cfg, sd, x, shapes = variant_inputs(g, tag)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    out, st = FRCNNOracle(cfg, sd).forward(x, shapes, return_stages=True)
    close(st["res4"][:, :32], g[f"{tag}/res4_c0_31"])
    close(st["res4"].double().sum(dim=(2, 3)).float(), g[f"{tag}/res4_sum"])
    close(st["feature_pooled"], g[f"{tag}/feature_pooled"])
    np.testing.assert_array_equal(out["preds_per_image"].numpy(), g[f"{tag}/preds_per_image"])
    for i in range(len(shapes)):
        np.testing.assert_array_equal(out["obj_ids"][i].numpy(), g[f"{tag}/obj_ids_{i}"])
        np.testing.assert_array_equal(out["attr_ids"][i].numpy(), g[f"{tag}/attr_ids_{i}"])
        close(out["obj_probs"][i], g[f"{tag}/obj_probs_{i}"])
        close(out["boxes"][i], g[f"{tag}/boxes_{i}"])
        close(out["roi_features"][i], g[f"{tag}/roi_features_{i}"])
