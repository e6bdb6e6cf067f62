"""Seeded synthetic weights in the reference's state_dict key layout.

The real checkpoint ("unc-nlp/frcnn-vg-finetuned") is fetched by name upstream
(reference: vltk/adapters/frcnn.py:30-32) and is unavailable offline
(SURVEY.md D3), so tests and benchmarks use weights generated here.  Key names
and shapes follow the reference module tree (vltk/modeling/frcnn.py:857-979
stem/bottleneck, :1545-1555 RPN head, :1365-1385 res5, :1705-1719 predictor,
:1429-1431 cell anchors) so that a real checkpoint dropped in by a user loads
through the same path.

The generator does NOT use torch's RNG: every tensor gets its own
numpy PCG64 stream keyed by (seed, crc32(name)), so the GPU box regenerates
bit-identical weights from the seed alone and nothing large is committed.

Scales are chosen so that activations stay O(1) through ~100 layers (last BN
of every bottleneck has a small gamma, as in zero-init-residual practice) and
the classifier / objectness logits have std >= 3 so arg-max and NMS margins
are not vacuous (SURVEY.md §7 "Synthetic weights").
"""
import math
import os
import zlib
from collections import OrderedDict

import numpy as np

_CALIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
# head -> target std of its output on the calibration batch (tools/calibrate_weights.py)
HEAD_TARGET_STD = OrderedDict([
    ("proposal_generator.rpn_head.objectness_logits", 4.0),
    ("proposal_generator.rpn_head.anchor_deltas", 0.25),
    ("roi_heads.box_predictor.cls_score", 4.0),
    ("roi_heads.box_predictor.bbox_pred", 0.5),
    ("roi_heads.box_predictor.attr_score", 4.0),
])

BLOCKS_PER_STAGE = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}  # frcnn.py:226


def cell_anchors(sizes, aspect_ratios):
    """reference: AnchorGenerator.generate_cell_anchors frcnn.py:1479-1497 (float64 math, cast to f32)."""
    out = []
    for size in sizes:
        area = size ** 2.0
        for ar in aspect_ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return np.asarray(out, dtype=np.float64).astype(np.float32)


def _bottleneck_spec(prefix, cin, cmid, cout, groups, has_shortcut):
    spec = []
    if has_shortcut:
        spec.append((prefix + ".shortcut", (cout, cin, 1, 1), "conv_bn"))
    spec.append((prefix + ".conv1", (cmid, cin, 1, 1), "conv_bn"))
    spec.append((prefix + ".conv2", (cmid, cmid // groups, 3, 3), "conv_bn"))
    spec.append((prefix + ".conv3", (cout, cmid, 1, 1), "conv_bn_last"))
    return spec


def layer_spec(cfg):
    """Ordered [(module_prefix, weight_shape, kind)] for every parametrised layer."""
    r = cfg.RESNETS
    groups, wpg = r.NUM_GROUPS, r.WIDTH_PER_GROUP
    stem_c, res2_c = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS
    nblocks = BLOCKS_PER_STAGE[r.DEPTH]
    spec = [("backbone.stem.conv1", (stem_c, len(cfg.MODEL.PIXEL_MEAN), 7, 7), "conv_bn")]
    cin, cout, cmid = stem_c, res2_c, groups * wpg
    for si, stage in enumerate(("res2", "res3", "res4")):
        for b in range(nblocks[si]):
            spec += _bottleneck_spec(f"backbone.{stage}.{b}", cin, cmid, cout, groups, cin != cout)
            cin = cout
        cout *= 2
        cmid *= 2
    res4_c = cin
    hid = cfg.PROPOSAL_GENERATOR.HIDDEN_CHANNELS
    hid = res4_c if hid == -1 else hid
    A = len(cfg.ANCHOR_GENERATOR.SIZES[0]) * len(cfg.ANCHOR_GENERATOR.ASPECT_RATIOS[0])
    spec += [
        ("proposal_generator.rpn_head.conv", (hid, res4_c, 3, 3), "conv_bias"),
        ("proposal_generator.rpn_head.objectness_logits", (A, hid, 1, 1), "conv_bias_obj"),
        ("proposal_generator.rpn_head.anchor_deltas", (4 * A, hid, 1, 1), "conv_bias_delta"),
    ]
    res5_c = res2_c * 8
    cmid5 = groups * wpg * 8
    cin = res4_c
    for b in range(3):
        spec += _bottleneck_spec(f"roi_heads.res5.{b}", cin, cmid5, res5_c, groups, cin != res5_c)
        cin = res5_c
    C = cfg.ROI_HEADS.NUM_CLASSES
    nb = 1 if cfg.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG else C
    spec += [
        ("roi_heads.box_predictor.cls_score", (C + 1, res5_c), "fc_cls"),
        ("roi_heads.box_predictor.bbox_pred", (nb * 4, res5_c), "fc_box"),
    ]
    if cfg.ROI_BOX_HEAD.ATTR:
        spec += [
            ("roi_heads.box_predictor.cls_embedding", (C + 1, res5_c // 8), "embedding"),
            ("roi_heads.box_predictor.fc_attr", (res5_c // 4, res5_c + res5_c // 8), "fc"),
            ("roi_heads.box_predictor.attr_score", (cfg.ROI_BOX_HEAD.NUM_ATTRS + 1, res5_c // 4), "fc_cls"),
        ]
    return spec


def fpn_layer_spec(cfg):
    """The FPN detector's parametrised layers in detectron2's key layout (backbone.bottom_up.*, backbone.fpn_lateral{2..5},
    backbone.fpn_output{2..5}, proposal_generator.rpn_head.*, roi_heads.box_head.fc{1,2}, roi_heads.box_predictor.*)."""
    r = cfg.RESNETS
    groups, wpg = r.NUM_GROUPS, r.WIDTH_PER_GROUP
    stem_c, res2_c = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS
    nblocks = BLOCKS_PER_STAGE[r.DEPTH]
    spec = [("backbone.bottom_up.stem.conv1", (stem_c, len(cfg.MODEL.PIXEL_MEAN), 7, 7), "conv_bn")]
    cin, cout, cmid = stem_c, res2_c, groups * wpg
    stage_c = []
    for si, stage in enumerate(("res2", "res3", "res4", "res5")):
        for b in range(nblocks[si]):
            spec += _bottleneck_spec(f"backbone.bottom_up.{stage}.{b}", cin, cmid, cout, groups, cin != cout)
            cin = cout
        stage_c.append(cout)
        cout *= 2
        cmid *= 2
    fc = cfg.FPN.OUT_CHANNELS
    for lvl, c in zip((2, 3, 4, 5), stage_c):
        spec.append((f"backbone.fpn_lateral{lvl}", (fc, c, 1, 1), "conv_bias_lin"))
        spec.append((f"backbone.fpn_output{lvl}", (fc, fc, 3, 3), "conv_bias_lin"))
    hid = cfg.PROPOSAL_GENERATOR.HIDDEN_CHANNELS
    hid = fc if hid == -1 else hid
    A = len(cfg.ANCHOR_GENERATOR.SIZES[0]) * len(cfg.ANCHOR_GENERATOR.ASPECT_RATIOS[0])
    spec += [
        ("proposal_generator.rpn_head.conv", (hid, fc, 3, 3), "conv_bias"),
        ("proposal_generator.rpn_head.objectness_logits", (A, hid, 1, 1), "conv_bias_obj"),
        ("proposal_generator.rpn_head.anchor_deltas", (4 * A, hid, 1, 1), "conv_bias_delta"),
    ]
    P, fcd = cfg.ROI_BOX_HEAD.POOLER_RESOLUTION, cfg.ROI_BOX_HEAD.FC_DIM
    cin = fc * P * P
    for i in range(cfg.ROI_BOX_HEAD.NUM_FC):
        spec.append((f"roi_heads.box_head.fc{i + 1}", (fcd, cin), "fc"))
        cin = fcd
    C = cfg.ROI_HEADS.NUM_CLASSES
    nb = 1 if cfg.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG else C
    spec += [("roi_heads.box_predictor.cls_score", (C + 1, fcd), "fc_cls"),
             ("roi_heads.box_predictor.bbox_pred", (nb * 4, fcd), "fc_box")]
    if cfg.ROI_BOX_HEAD.ATTR:            # FastRCNNOutputLayers sizes everything from input_size (frcnn.py:1711-1719)
        spec += [("roi_heads.box_predictor.cls_embedding", (C + 1, fcd // 8), "embedding"),
                 ("roi_heads.box_predictor.fc_attr", (fcd // 4, fcd + fcd // 8), "fc"),
                 ("roi_heads.box_predictor.attr_score", (cfg.ROI_BOX_HEAD.NUM_ATTRS + 1, fcd // 4), "fc_cls")]
    return spec


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def calib_path(cfg, seed):
    r = cfg.RESNETS
    fpn = "_fpn" if len(cfg.RPN.IN_FEATURES) > 1 else ""
    return os.path.join(_CALIB_DIR, f"head_calib_r{r.DEPTH}_g{r.NUM_GROUPS}x{r.WIDTH_PER_GROUP}{fpn}_seed{seed}.npz")


def make_state_dict(cfg, seed=1234, calibrated=True):
    """name -> np.ndarray (float32; num_batches_tracked int64), in the reference's key layout.

    With `calibrated` and a committed calibration file for (depth, groups, seed)
    (vltk_amd/data/, written once by tools/calibrate_weights.py from a CPU
    forward of the base weights), every prediction head is rescaled and
    re-biased so that its outputs are zero-mean with the std of HEAD_TARGET_STD
    on the calibration batch: random residual nets have a large common-mode
    feature component that would otherwise make one class win everywhere.
    The file holds one gain and one bias vector per head (~10 KB); generation
    itself stays a pure function of (cfg, seed, file).
    """
    sd = _base_state_dict(cfg, seed)
    path = calib_path(cfg, seed)
    if calibrated and os.path.exists(path):
        cal = np.load(path)
        for head in HEAD_TARGET_STD:
            if head + ".gain" in cal.files:
                sd[head + ".weight"] = (sd[head + ".weight"] * cal[head + ".gain"]).astype(np.float32)
                sd[head + ".bias"] = cal[head + ".bias"].astype(np.float32)
    return sd


def _base_state_dict(cfg, seed):
    sd = OrderedDict()
    fpn = len(cfg.RPN.IN_FEATURES) > 1
    for prefix, shape, kind in (fpn_layer_spec(cfg) if fpn else layer_spec(cfg)):
        g = _rng(seed, prefix)
        fan_in = int(np.prod(shape[1:]))
        if kind.startswith("conv_bn"):
            std = math.sqrt(2.0 / fan_in)
            if prefix.endswith("stem.conv1"):
                std /= 50.0      # inputs are mean-subtracted 0-255 pixels (std ~50): bring the stem output to O(1)
            sd[prefix + ".weight"] = (g.standard_normal(shape) * std).astype(np.float32)
            c = shape[0]
            lo, hi = (0.1, 0.3) if kind == "conv_bn_last" else (0.5, 1.5)
            sd[prefix + ".norm.weight"] = g.uniform(lo, hi, c).astype(np.float32)
            sd[prefix + ".norm.bias"] = (g.standard_normal(c) * 0.1).astype(np.float32)
            sd[prefix + ".norm.running_mean"] = (g.standard_normal(c) * 0.1).astype(np.float32)
            sd[prefix + ".norm.running_var"] = g.uniform(0.5, 1.5, c).astype(np.float32)
            sd[prefix + ".norm.num_batches_tracked"] = np.asarray(0, dtype=np.int64)
        elif kind.startswith("conv_bias"):
            gain = {"conv_bias": math.sqrt(2.0), "conv_bias_obj": 4.0, "conv_bias_delta": 0.35, "conv_bias_lin": 1.0}[kind]
            sd[prefix + ".weight"] = (g.standard_normal(shape) * gain / math.sqrt(fan_in)).astype(np.float32)
            sd[prefix + ".bias"] = (g.standard_normal(shape[0]) * 0.05).astype(np.float32)
        elif kind == "embedding":
            sd[prefix + ".weight"] = g.standard_normal(shape).astype(np.float32)
        else:
            gain = {"fc_cls": 4.0, "fc_box": 0.5, "fc": math.sqrt(2.0)}[kind]
            sd[prefix + ".weight"] = (g.standard_normal(shape) * gain / math.sqrt(fan_in)).astype(np.float32)
            sd[prefix + ".bias"] = (g.standard_normal(shape[0]) * 0.05).astype(np.float32)
        if not fpn and prefix == "backbone.res4.%d.conv3" % (BLOCKS_PER_STAGE[cfg.RESNETS.DEPTH][2] - 1):
            sd["proposal_generator.anchor_generator.cell_anchors.0"] = cell_anchors(
                cfg.ANCHOR_GENERATOR.SIZES[0], cfg.ANCHOR_GENERATOR.ASPECT_RATIOS[0])
        if fpn and prefix == "backbone.fpn_output5":
            ratios = cfg.ANCHOR_GENERATOR.ASPECT_RATIOS
            for li, sizes in enumerate(cfg.ANCHOR_GENERATOR.SIZES):
                sd[f"proposal_generator.anchor_generator.cell_anchors.{li}"] = cell_anchors(
                    sizes, ratios[li] if len(ratios) > 1 else ratios[0])
    return sd


def synthetic_images(n, h, w, seed=0xF2C, rank=0):
    """Counter-based synthetic batch: N(0, 50^2) clipped to [-123, 152], NCHW f32."""
    g = np.random.Generator(np.random.Philox(key=seed ^ rank))
    x = g.standard_normal((n, 3, h, w), dtype=np.float32) * 50.0
    return np.clip(x, -123.0, 152.0)
