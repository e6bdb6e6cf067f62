#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own module (build container only).

Runs only where /root/reference exists (never on the GPU box, never from tests).
It loads `/root/reference/vltk/modeling/frcnn.py` under stub modules, following
the recipe recorded in SURVEY.md §8c: the reference's package `__init__` cannot
be imported (datasets version skew, missing cv2/wget/torchvision), so `vltk`,
`vltk.compat`, `vltk.decorators` and the three torchvision entry points are
registered as stubs in `sys.modules`; everything else executed is the
reference's own arithmetic.  The torchvision ops are the restatements in
oracle/tv_ops.c (PARITY UNPINNED for those three ops).

Nothing is copied from the reference: the outputs are data (inputs, seeds,
expected tensors) written to tests/golden/*.npz.  Large weights are never
stored -- they are regenerated from (config, seed) by vltk_amd.weights.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import frcnn_oracle as orc          # noqa: E402  (tv-op restatements)
from vltk_amd.config import Config, vg_c4_config_dict   # noqa: E402
from vltk_amd.weights import make_state_dict, synthetic_images   # noqa: E402

REF = "/root/reference/vltk/modeling/frcnn.py"
OUT = os.path.join(ROOT, "tests", "golden")


class _RoIPool(torch.nn.Module):
    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale

    def forward(self, x, rois):
        size = self.output_size[0] if isinstance(self.output_size, (tuple, list)) else self.output_size
        return orc.roi_pool(x, rois, size, self.spatial_scale)


def _stable_sorted_nms(boxes, scores, thr):
    return orc.nms(boxes, scores, thr)


def load_reference():
    def _offline(*a, **k):
        raise EnvironmentError("offline: fetch-by-name loaders are not available")

    tv, ops, bx = (types.ModuleType(n) for n in ("torchvision", "torchvision.ops", "torchvision.ops.boxes"))
    ops.RoIPool, bx.nms, bx.batched_nms = _RoIPool, _stable_sorted_nms, orc.batched_nms
    ops.boxes, tv.ops = bx, ops
    vp = types.ModuleType("vltk")
    vp.__path__ = []
    vp.decorators = types.ModuleType("vltk.decorators")
    vc = types.ModuleType("vltk.compat")
    vc.WEIGHTS_NAME = "pytorch_model.bin"
    vc.Config = Config
    for n in ("cached_path", "hf_bucket_url", "is_remote_url", "load_checkpoint"):
        setattr(vc, n, _offline)
    sys.modules.update({"torchvision": tv, "torchvision.ops": ops, "torchvision.ops.boxes": bx,
                        "vltk": vp, "vltk.decorators": vp.decorators, "vltk.compat": vc})
    spec = importlib.util.spec_from_file_location("ref_frcnn", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def to_torch_sd(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def np_(t):
    return t.detach().cpu().numpy()


def tie_free(x, what):
    v = np.sort(np.asarray(x).reshape(-1))
    assert (np.diff(v) != 0).all(), f"{what}: ties present -- pick another seed"


# ---------------------------------------------------------------------------
def kat_ops(ref):
    """Per-op known-answer vectors from the reference's classes (tiny shapes, weights stored)."""
    g = torch.Generator().manual_seed(20260101)
    out = {}

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    def fill_bn(conv):
        c = conv.norm.num_features
        conv.norm.weight.data = torch.rand(c, generator=g) + 0.5
        conv.norm.bias.data = rnd(c, scale=0.1)
        conv.norm.running_mean.data = rnd(c, scale=0.1)
        conv.norm.running_var.data = torch.rand(c, generator=g) + 0.5

    def dump_sd(prefix, mod):
        for k, v in mod.state_dict().items():
            out[f"{prefix}/sd/{k}"] = np_(v)

    # -- stem (caffe max-pool, ceil_mode) on odd sizes
    for tag, caffe, hw in (("stem_caffe", True, (63, 95)), ("stem_pad1", False, (64, 96))):
        stem = ref.BasicStem(3, 16, "BN", caffe_maxpool=caffe).eval()
        stem.conv1.weight.data = rnd(16, 3, 7, 7, scale=0.1)
        fill_bn(stem.conv1)
        x = rnd(2, 3, *hw, scale=50.0)
        out[f"{tag}/x"] = np_(x)
        out[f"{tag}/y"] = np_(stem(x.clone()))
        dump_sd(tag, stem)

    # -- bottlenecks: (cin, cout, mid, stride, groups, stride_in_1x1, dilation)
    for tag, a in {
        "blk_s2_in1x1": (32, 64, 16, 2, 1, True, 1),
        "blk_s2_in3x3": (32, 64, 16, 2, 1, False, 1),
        "blk_identity": (64, 64, 16, 1, 1, True, 1),
        "blk_dil2": (32, 64, 32, 1, 1, True, 2),
        "blk_groups": (32, 64, 32, 1, 8, False, 1),
    }.items():
        cin, cout, mid, stride, groups, s1x1, dil = a
        blk = ref.BottleneckBlock(cin, cout, bottleneck_channels=mid, stride=stride, num_groups=groups,
                                  norm="BN", stride_in_1x1=s1x1, dilation=dil).eval()
        for name in ("shortcut", "conv1", "conv2", "conv3"):
            conv = getattr(blk, name)
            if conv is None:
                continue
            fan = conv.weight[0].numel()
            conv.weight.data = rnd(*conv.weight.shape, scale=(2.0 / fan) ** 0.5)
            fill_bn(conv)
        x = rnd(2, cin, 13, 18)
        out[f"{tag}/x"] = np_(x)
        out[f"{tag}/y"] = np_(blk(x.clone()))
        out[f"{tag}/args"] = np.asarray([cin, cout, mid, stride, groups, int(s1x1), dil])
        dump_sd(tag, blk)

    # -- anchors (frcnn.py:1406-1510): two grid sizes, default sizes/ratios
    cfg = Config(vg_c4_config_dict())
    ag = ref.AnchorGenerator(cfg, [ref.ShapeSpec(channels=8, stride=16)])
    out["anchors/cell"] = np_(ag.cell_anchors[0])
    for hw in ((3, 4), (10, 14)):
        a = ag([torch.zeros(2, 8, *hw)])
        out[f"anchors/grid_{hw[0]}x{hw[1]}"] = np_(a[0, 0])
        assert a.shape == (2, 1, hw[0] * hw[1] * 15, 4)

    # -- apply_deltas incl. the scale clamp and k>1 class-specific deltas
    for tag, w in (("deltas_rpn", (1.0, 1.0, 1.0, 1.0)), ("deltas_roi", (10.0, 10.0, 5.0, 5.0))):
        t = ref.Box2BoxTransform(weights=w)
        boxes = torch.rand(64, 4, generator=g) * 200
        boxes[:, 2:] += boxes[:, :2] + 1.0
        d = rnd(64, 12, scale=2.0)
        d[0, :4] = torch.tensor([0.1, -0.2, 10.0, 0.5]) * torch.tensor(w)
        boxes[0] = torch.tensor([0.0, 0.0, 16.0, 16.0])
        out[f"{tag}/boxes"], out[f"{tag}/deltas"] = np_(boxes), np_(d)
        out[f"{tag}/y"] = np_(t.apply_deltas(d, boxes))

    # -- RPN head + proposals (RPN.forward frcnn.py:1640-1673) on a small map
    cfgd = vg_c4_config_dict()
    cfgd["proposal_generator"]["hidden_channels"] = 32
    cfgd["rpn"]["pre_nms_topk_test"] = 400
    cfgd["rpn"]["post_nms_topk_test"] = 40
    cfg_s = Config(cfgd)
    rpn = ref.RPN(cfg_s, {"res4": ref.ShapeSpec(channels=48, stride=16)}).eval()
    for name, sc in (("conv", 0.1), ("objectness_logits", 0.5), ("anchor_deltas", 0.08)):
        m = getattr(rpn.rpn_head, name)
        m.weight.data = rnd(*m.weight.shape, scale=sc)
        m.bias.data = rnd(*m.bias.shape, scale=0.05)
    feat = torch.relu(rnd(2, 48, 9, 13))
    shapes = torch.tensor([[144, 208], [130, 190]])
    images = torch.zeros(2, 3, 144, 208)
    with torch.no_grad():
        obj, dlt = rpn.rpn_head([feat])
        tie_free(np_(obj[0]), "rpn kat logits")
        pb, lg = rpn(images, shapes, {"res4": feat})
    out["rpn/feat"], out["rpn/shapes"] = np_(feat), np_(shapes)
    out["rpn/objectness"], out["rpn/deltas"] = np_(obj[0]), np_(dlt[0])
    for i in range(2):
        out[f"rpn/boxes_{i}"], out[f"rpn/logits_{i}"] = np_(pb[i]), np_(lg[i])
    dump_sd("rpn", rpn)

    # -- predictor (FastRCNNOutputLayers frcnn.py:1676-1740), tiny
    pred = ref.FastRCNNOutputLayers(64, 10, False, use_attr=True, num_attrs=5).eval()
    for n_, p in pred.named_parameters():
        p.data = rnd(*p.shape, scale=0.3)
    f = torch.relu(rnd(12, 64))
    with torch.no_grad():
        s, a, d = pred(f)
    out["pred/x"], out["pred/scores"], out["pred/attr"], out["pred/deltas"] = np_(f), np_(s), np_(a), np_(d)
    dump_sd("pred", pred)

    # -- ROIOutputs.inference (frcnn.py:1262-1294) incl. the threshold-list retry of do_nms
    cfgd = vg_c4_config_dict()
    cfgd["roi_heads"]["num_classes"] = 10
    cfgd["roi_box_head"]["num_attrs"] = 5
    cfgd["min_detections"], cfgd["max_detections"] = 6, 8
    ro = ref.ROIOutputs(Config(cfgd))
    ro.nms_thresh = [0.05, 0.3, 0.9]
    R = [20, 17]
    props = []
    for r in R:
        b = torch.rand(r, 4, generator=g) * 80
        b[:, 2:] = b[:, :2] + 20 + torch.rand(r, 2, generator=g) * 60
        props.append(b)
    K = sum(R)
    obj_logits, attr_logits = rnd(K, 11, scale=3.0), rnd(K, 6, scale=3.0)
    box_deltas, feats = rnd(K, 40, scale=1.0), torch.relu(rnd(K, 16))
    sizes = [(120, 160), (100, 150)]
    scales = torch.tensor([[1.5, 2.0], [0.5, 0.75]])
    for tag, sc in (("roiout", None), ("roiout_scaled", scales)):
        res = ro(obj_logits, attr_logits, box_deltas, [p.clone() for p in props], feats, sizes, scales=sc)
        for name, lst in zip(("boxes", "classes", "probs", "attrs", "attr_probs", "feats"), res):
            for i, t in enumerate(lst):
                out[f"{tag}/{name}_{i}"] = np_(t)
    out["roiout/obj_logits"], out["roiout/attr_logits"] = np_(obj_logits), np_(attr_logits)
    out["roiout/box_deltas"], out["roiout/feats_in"] = np_(box_deltas), np_(feats)
    out["roiout/sizes"], out["roiout/scales"] = np.asarray(sizes), np_(scales)
    for i, p in enumerate(props):
        out[f"roiout/props_{i}"] = np_(p)
    out["roiout/nms_thresh"] = np.asarray(ro.nms_thresh)
    np.savez_compressed(os.path.join(OUT, "kat_ops.npz"), **out)
    print("kat_ops.npz:", len(out), "arrays")


# ---------------------------------------------------------------------------
def e2e(ref, name, n, h, w, shapes, post_topk, det, seed, depth=101):
    """End-to-end FRCNN.forward on small images with seeded R101 weights (weights NOT stored)."""
    cfgd = vg_c4_config_dict(depth=depth, post_nms_topk=post_topk, detections=det)
    cfg = Config(cfgd)
    sd = make_state_dict(cfg, seed=seed)
    net = ref.FRCNN(cfg).eval()
    net.load_state_dict(to_torch_sd(sd), strict=True)
    images = torch.from_numpy(synthetic_images(n, h, w, seed=seed))
    shp = torch.tensor(shapes)
    for i, (hh, ww) in enumerate(shapes):      # zero-pad beyond the content, like Preprocess does
        images[i, :, hh:, :] = 0
        images[i, :, :, ww:] = 0
    stages = {}
    hooks = [
        net.backbone.register_forward_hook(lambda m, i, o: stages.__setitem__("res4", o["res4"])),
        net.proposal_generator.rpn_head.register_forward_hook(
            lambda m, i, o: stages.update(obj=o[0][0], dlt=o[1][0])),
        net.proposal_generator.register_forward_hook(lambda m, i, o: stages.update(pboxes=o[0], plogits=o[1])),
        net.roi_heads.register_forward_hook(
            lambda m, i, o: stages.update(obj_logits=o[0], attr_logits=o[1], box_deltas=o[2], pooled=o[3])),
        net.roi_heads.pooler.register_forward_hook(lambda m, i, o: stages.__setitem__("roipool", o)),
    ]
    with torch.no_grad():
        o = net(images, shp)
    for hk in hooks:
        hk.remove()
    tie_free(np_(stages["obj"]), f"{name} rpn logits")
    out = {"images_seed": np.asarray(seed), "shapes": np.asarray(shapes), "nhw": np.asarray([n, h, w]),
           "post_topk": np.asarray(post_topk), "det": np.asarray(det), "depth": np.asarray(depth),
           "weights_seed": np.asarray(seed)}
    out["res4"] = np_(stages["res4"])
    out["rpn_objectness"], out["rpn_deltas"] = np_(stages["obj"]), np_(stages["dlt"])
    for i in range(n):
        out[f"proposal_boxes_{i}"], out[f"proposal_logits_{i}"] = np_(stages["pboxes"][i]), np_(stages["plogits"][i])
        for k in ("obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "roi_features"):
            out[f"{k}_{i}"] = np_(o[k][i])
    out["preds_per_image"] = np_(o["preds_per_image"])
    out["roipool_c0_7"] = np_(stages["roipool"][:, :8])            # channel subset: the full tensor is too large
    out["roipool_sum"] = np_(stages["roipool"].double().sum(dim=(2, 3)).float())
    out["feature_pooled"] = np_(stages["pooled"])
    out["obj_logits"], out["attr_logits"] = np_(stages["obj_logits"]), np_(stages["attr_logits"])
    bd = stages["box_deltas"]
    out["box_deltas_head"] = np_(bd[:, :256])                     # first 64 classes + a full-row checksum
    out["box_deltas_rowsum"] = np_(bd.double().sum(1).float())
    # score margins: how far the top-1 class is from the runner-up (index parity is margin-aware)
    p = torch.softmax(stages["obj_logits"], -1)[:, :-1]
    top2 = p.topk(2, dim=1).values
    out["cls_margin"] = np_(top2[:, 0] - top2[:, 1])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "preds_per_image", out["preds_per_image"], "res4 |mean| %.3f max %.2f" %
          (np.abs(out["res4"]).mean(), out["res4"].max()),
          "feat max %.2f" % out["feature_pooled"].max(),
          "n_props", [len(b) for b in stages["pboxes"]],
          "distinct obj ids", len(np.unique(np.concatenate([out[f"obj_ids_{i}"] for i in range(n)]))),
          "min cls margin %.2e" % out["cls_margin"].min())


VARIANTS = {
    # tag: config overrides (section, key, value) on the depth-50 build config
    "resnext50_8x8d": [("resnets", "num_groups", 8), ("resnets", "width_per_group", 8)],
    "r50_halve": [("roi_box_head", "res5halve", True)],
    "r50_halve_s3x3": [("roi_box_head", "res5halve", True), ("resnets", "stride_in_1x1", False)],
    # (CLS_AGNOSTIC_BBOX_REG=true is not a variant: the reference's do_nms indexes r*C + class into the R
    #  class-agnostic boxes and raises IndexError, frcnn.py:127-129)
}


# BASELINE configs[3]: ResNeXt-152 32x8d at its real depth and group count (small image, same seeded weights as bench.py
# --arch x152: seed 1234, head calibration vltk_amd/data/head_calib_r152_g32x8_seed1234.npz); kept in its own file so the
# other variants' fixture stays byte-identical
VARIANTS_X152 = {
    "resnext152_32x8d": [("resnets", "depth", 152), ("resnets", "num_groups", 32), ("resnets", "width_per_group", 8)],
}


def variant_config_dict(tag, post_topk=16, det=6, variants=None):
    d = vg_c4_config_dict(depth=50, post_nms_topk=post_topk, detections=det)
    for sec, key, val in (variants or VARIANTS)[tag]:
        d[sec][key] = val
    return d


def e2e_variants(ref, n=2, h=128, w=160, shapes=((128, 160), (112, 150)), seed=4321, variants=None, outfile="e2e_variants.npz"):
    """Compact end-to-end vectors of the reference for configuration switches the main fixture does not take:
    ResNeXt groups (frcnn.py:217-219, 942-952), RES5HALVE (:1345-1355), stride in the 3x3 (:932).  Weights are regenerated from the seed, never stored."""
    out = {"nhw": np.asarray([n, h, w]), "shapes": np.asarray(shapes), "seed": np.asarray(seed)}
    variants = variants or VARIANTS
    for tag in variants:
        cfg = Config(variant_config_dict(tag, variants=variants))
        sd = make_state_dict(cfg, seed=seed)
        net = ref.FRCNN(cfg).eval()
        net.load_state_dict(to_torch_sd(sd), strict=True)
        images = torch.from_numpy(synthetic_images(n, h, w, seed=seed))
        for i, (hh, ww) in enumerate(shapes):
            images[i, :, hh:, :] = 0
            images[i, :, :, ww:] = 0
        st = {}
        hooks = [
            net.backbone.register_forward_hook(lambda m, i, o: st.__setitem__("res4", o["res4"])),
            net.proposal_generator.rpn_head.register_forward_hook(lambda m, i, o: st.update(obj=o[0][0])),
            net.roi_heads.register_forward_hook(lambda m, i, o: st.update(obj_logits=o[0], pooled=o[3])),
        ]
        with torch.no_grad():
            o = net(images, torch.tensor(shapes))
        for hk in hooks:
            hk.remove()
        tie_free(np_(st["obj"]), f"{tag} rpn logits")
        out[f"{tag}/res4_c0_31"] = np_(st["res4"][:, :32])
        out[f"{tag}/res4_sum"] = np_(st["res4"].double().sum(dim=(2, 3)).float())
        out[f"{tag}/feature_pooled"] = np_(st["pooled"])
        out[f"{tag}/preds_per_image"] = np_(o["preds_per_image"])
        for k in ("obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "roi_features"):
            for i in range(n):
                out[f"{tag}/{k}_{i}"] = np_(o[k][i])
        top2 = st["obj_logits"].softmax(-1)[:, :-1].topk(2, dim=-1).values
        out[f"{tag}/cls_margin"] = np_(top2[:, 0] - top2[:, 1])
        print(tag, "preds", out[f"{tag}/preds_per_image"], "res4 max %.2f" % float(st["res4"].max()),
              "feat max %.2f" % float(st["pooled"].max()), "min cls margin %.2e" % out[f"{tag}/cls_margin"].min(),
              "obj ids", out[f"{tag}/obj_ids_0"][:6])
    np.savez_compressed(os.path.join(OUT, outfile), **out)
    print(outfile + ":", len(out), "arrays")


def fpn_ops(ref):
    """Vectors for the FPN-side fragments the reference holds: LastLevelMaxPool :825-836,
    LastLevelP6P7 :839-854, assign_boxes_to_levels :444-460 (called with objects that have `.area()`, which is what
    the function expects; the reference's own callers pass tensors and never reach it)."""
    g = torch.Generator().manual_seed(20260202)
    out = {}
    x = torch.randn(2, 16, 9, 13, generator=g)
    out["maxpool/x"], out["maxpool/y"] = np_(x), np_(ref.LastLevelMaxPool()(x)[0])
    blk = ref.LastLevelP6P7(32, 16).eval()
    for prm in blk.parameters():
        prm.data = torch.randn(prm.shape, generator=g) * 0.1
    c5 = torch.randn(2, 32, 11, 14, generator=g)
    with torch.no_grad():
        p6, p7 = blk(c5)
    out["p6p7/c5"], out["p6p7/p6"], out["p6p7/p7"] = np_(c5), np_(p6), np_(p7)
    for k, v in blk.state_dict().items():
        out[f"p6p7/sd/{k}"] = np_(v)

    class B:                                   # what assign_boxes_to_levels expects of a box list
        def __init__(self, t):
            self.t = t

        def area(self):
            return (self.t[:, 2] - self.t[:, 0]) * (self.t[:, 3] - self.t[:, 1])
    xy = torch.rand(400, 2, generator=g) * 600
    wh = torch.exp(torch.rand(400, 2, generator=g) * 7.5)            # 1 .. 1800 px sides
    boxes = torch.cat([xy, xy + wh], 1)
    exact = torch.tensor([[0, 0, 224, 224], [0, 0, 112, 112], [10, 10, 458, 458], [0, 0, 56, 56], [0, 0, 896, 896], [5, 5, 5, 5],
                          [0, 0, 1, 1], [0, 0, 4000, 4000]], dtype=torch.float32)
    boxes = torch.cat([boxes, exact])
    lv = ref.assign_boxes_to_levels([B(boxes[:200]), B(boxes[200:])], 2, 5, 224, 4)
    out["levels/boxes"], out["levels/assigned"] = np_(boxes), np_(lv)
    # multi-level proposals: the reference's RPN class cannot run several levels (below), but its pieces can: per-level
    # anchors from AnchorGenerator.grid_anchors, per-level decoding with Box2BoxTransform.apply_deltas in RPNOutputs' (n, y, x, a)
    # order (:758-780), then find_top_rpn_proposals itself over the list of levels (:264-390)
    cfgd = vg_c4_config_dict(depth=50, post_nms_topk=40, detections=8)
    cfgd["anchor_generator"]["sizes"] = [[32], [64], [128]]
    cfg = Config(cfgd)
    strides, A, N = [4, 8, 16], 3, 2
    shapes_in = [ref.ShapeSpec(channels=8, stride=st) for st in strides]
    ag = ref.AnchorGenerator(cfg, shapes_in)
    hw = [(24, 32), (12, 16), (6, 8)]
    tr = ref.Box2BoxTransform(weights=(1.0, 1.0, 1.0, 1.0))
    img_shapes = [(96, 128), (90, 117)]
    props, logits = [], []
    for li, (h, w) in enumerate(hw):
        obj = torch.randn(N, A, h, w, generator=g)
        tie_free(np_(obj), f"rpn level {li} logits")
        dlt = torch.randn(N, 4 * A, h, w, generator=g) * 0.4
        anchors = ag.grid_anchors(hw)[li]
        d = dlt.view(N, A, 4, h, w).permute(0, 3, 4, 1, 2).reshape(-1, 4)
        anc = anchors.unsqueeze(0).expand(N, -1, -1).reshape(-1, 4)
        props.append(tr.apply_deltas(d, anc).view(N, -1, 4))
        logits.append(obj.permute(0, 2, 3, 1).reshape(N, -1))
        out[f"mlrpn/obj_{li}"], out[f"mlrpn/dlt_{li}"] = np_(obj), np_(dlt)
        out[f"mlrpn/cell_{li}"] = np_(ag.cell_anchors[li])
    res = ref.find_top_rpn_proposals(props, logits, [None] * N, img_shapes, 0.7, 200, 40, 0, False)
    for i, (b, sc) in enumerate(res):
        out[f"mlrpn/boxes_{i}"], out[f"mlrpn/logits_{i}"] = np_(b), np_(sc)
    out["mlrpn/strides"], out["mlrpn/shapes"] = np.asarray(strides), np.asarray(img_shapes)
    out["mlrpn/pre_post_thr"] = np.asarray([200, 40, 0.7])
    print("multi-level proposals kept:", [len(b) for b, _ in res])
    np.savez_compressed(os.path.join(OUT, "fpn_ops.npz"), **out)
    print("fpn_ops.npz:", len(out), "arrays; level histogram", np.bincount(out["levels/assigned"]))
    # multi-level RPN through the reference's RPN class (expected to fail: it stacks per-level anchors of unequal length)
    try:
        cfgd = vg_c4_config_dict(depth=50, post_nms_topk=50, detections=8)
        cfgd["rpn"]["in_features"] = ["p2", "p3"]
        cfgd["anchor_generator"]["sizes"] = [[32], [64]]
        cfgd["proposal_generator"]["hidden_channels"] = -1
        cfg = Config(cfgd)
        shapes = {"p2": ref.ShapeSpec(channels=16, stride=4), "p3": ref.ShapeSpec(channels=16, stride=8)}
        rpn = ref.RPN(cfg, shapes).eval()
        feats = {"p2": torch.randn(1, 16, 12, 16, generator=g), "p3": torch.randn(1, 16, 6, 8, generator=g)}
        with torch.no_grad():
            rpn(torch.zeros(1, 3, 48, 64), torch.tensor([[48, 64]]), feats)
        print("multi-level RPN: the reference ran it")
    except Exception as e:                                           # recorded in DESIGN.md
        print("multi-level RPN through the reference's RPN class fails:", type(e).__name__, str(e)[:160])


if __name__ == "__main__" and "--fpn" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    fpn_ops(load_reference())
    sys.exit(0)

if __name__ == "__main__" and "--x152" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    e2e_variants(load_reference(), seed=1234, variants=VARIANTS_X152, outfile="e2e_x152.npz")
    sys.exit(0)

if __name__ == "__main__" and "--variants" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    e2e_variants(load_reference())
    sys.exit(0)

if __name__ == "__main__" and "--preprocess" not in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = load_reference()
    kat_ops(ref)
    e2e(ref, "e2e_r101_small", n=2, h=160, w=224, shapes=[[160, 224], [144, 200]], post_topk=30, det=12, seed=1234)


# ---------------------------------------------------------------------------
def preprocess_golden():
    """Golden vectors of the reference's legacy `Preprocess` (vltk/legacy/processing.py:29-150):
    float HWC (BGR 0-255) tensors -> shortest-edge bilinear resize -> (x - mean)/std -> zero-pad to the batch max."""
    pk = types.ModuleType("vltk.legacy")
    pk.__path__ = []
    tc = types.ModuleType("vltk.legacy.transformers_compat")
    tc.img_tensorize = lambda *a, **k: None
    sys.modules.update({"vltk.legacy": pk, "vltk.legacy.transformers_compat": tc})
    spec = importlib.util.spec_from_file_location("vltk.legacy.processing", "/root/reference/vltk/legacy/processing.py")
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "vltk.legacy"
    spec.loader.exec_module(mod)
    out = {}
    for tag, (mn, mx), shapes in (("small", (48, 80), [(37, 53), (60, 41), (50, 50)]),
                                  ("capped", (64, 96), [(30, 90), (100, 20)]),
                                  ("vg", (800, 1333), [(375, 500)])):
        d = vg_c4_config_dict()
        d["input"]["min_size_test"], d["input"]["max_size_test"] = mn, mx
        pre = mod.Preprocess(Config(d))
        # raw images are NOT stored: tests regenerate them from (seed 7700 + index, shape)
        raws = [torch.from_numpy(np.random.Generator(np.random.PCG64(7700 + i)).uniform(0, 255, (h, w, 3)).astype(np.float32))
                for i, (h, w) in enumerate(shapes)]
        out[f"{tag}/raw_shapes"] = np.asarray(shapes)
        ids, images, sizes, scales = pre([r.clone() for r in raws], list(range(len(raws))))
        out[f"{tag}/minmax"] = np.asarray([mn, mx])
        if tag == "vg":      # full-size case: keep a checksum + a crop, not 12 MB
            out[f"{tag}/images_crop"] = np_(images[:, :, 100:164, 200:264])
            out[f"{tag}/images_sum"] = np_(images.double().sum(dim=(2, 3)).float())
            out[f"{tag}/images_shape"] = np.asarray(images.shape)
        else:
            out[f"{tag}/images"] = np_(images)
        out[f"{tag}/sizes"], out[f"{tag}/scales_yx"] = np_(sizes), np_(scales)
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **out)
    print("preprocess.npz:", {k: v.shape for k, v in out.items() if k.endswith(("images", "sizes"))})


if __name__ == "__main__" and "--preprocess" in sys.argv:
    preprocess_golden()
