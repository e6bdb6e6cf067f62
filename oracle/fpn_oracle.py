"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the FPN-side pieces.

What the reference holds (frcnn.py): `LastLevelMaxPool` :825-836, `LastLevelP6P7` :839-854, `assign_boxes_to_levels`
:444-460 (dead code: it calls `.area()` on plain tensors), the multi-level loop of `ROIPooler.forward` :1200-1224 (over
RoIPool, and unreachable for the same reason).  Those are pinned by vectors from the reference's own classes
(tools/gen_golden.py --fpn -> tests/golden/fpn_ops.npz).  What it does NOT hold: an FPN neck and RoIAlign, which
north_star names; they are restated from detectron2's FPN (lateral 1x1 + nearest 2x top-down + 3x3 output convs, all
with bias, no norm) and torchvision's roi_align -> PARITY UNPINNED for those two.
Only tests/, smoke() and bench's cpu_baseline may import this module."""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from .frcnn_oracle import _fp, _lib, roi_pool


def last_level_maxpool(p5):                       # frcnn.py:835-836
    return F.max_pool2d(p5, kernel_size=1, stride=2, padding=0)


def last_level_p6p7(c5, w6, b6, w7, b7):           # frcnn.py:850-854
    p6 = F.conv2d(c5, w6, b6, 2, 1)
    return p6, F.conv2d(F.relu(p6), w7, b7, 2, 1)


def assign_boxes_to_levels(boxes, min_level, max_level, canonical_box_size=224, canonical_level=4):   # frcnn.py:444-460
    boxes = torch.as_tensor(boxes).float()
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    lv = torch.floor(canonical_level + torch.log2(torch.sqrt(area) / canonical_box_size + 1e-8))
    return torch.clamp(lv, min=min_level, max=max_level).to(torch.int64) - min_level


def roi_align(x, rois, output_size, spatial_scale, sampling_ratio=0, aligned=True):
    """torchvision.ops.roi_align semantics (oracle/tv_ops.c vko_roi_align).  x [N,C,H,W] f32, rois [K,5]."""
    x = np.ascontiguousarray(x.detach().cpu().numpy(), dtype=np.float32)
    r = np.ascontiguousarray(torch.as_tensor(rois).detach().cpu().numpy(), dtype=np.float32)
    N, C, H, W = x.shape
    K, P = r.shape[0], int(output_size)
    out = np.zeros((K, C, P, P), dtype=np.float32)
    if K:
        fn = _lib().vko_roi_align
        fn.restype = None
        fn(_fp(x), N, C, H, W, _fp(r), K, ctypes.c_float(spatial_scale), P, P, int(sampling_ratio), int(bool(aligned)), _fp(out))
    return torch.from_numpy(out)


def multilevel_pool(feats, scales, rois, output_size, kind="align", sampling_ratio=0, aligned=True, canonical_box_size=224,
                    canonical_level=4):
    """ROIPooler.forward :1181-1224 with its level loop (`output[inds] = pooler(x_level, rois[inds])`)."""
    rois = torch.as_tensor(rois).float()
    min_level, max_level = int(round(-np.log2(scales[0]))), int(round(-np.log2(scales[-1])))
    if len(feats) > 1:
        lv = assign_boxes_to_levels(rois[:, 1:], min_level, max_level, canonical_box_size, canonical_level)
    else:
        lv = torch.zeros(len(rois), dtype=torch.int64)
    out = torch.zeros((len(rois), feats[0].shape[1], output_size, output_size))
    for li, (x, s) in enumerate(zip(feats, scales)):
        inds = torch.nonzero(lv == li).squeeze(1)
        if len(inds) == 0:
            continue
        out[inds] = roi_align(x, rois[inds], output_size, s, sampling_ratio, aligned) if kind == "align" else roi_pool(x, rois[inds], output_size, s)
    return out, lv


def fpn_neck(feats, lateral, output, top_block="maxpool"):
    """detectron2 FPN.forward: feats = [C2..C5] (fine -> coarse); lateral[i] / output[i] = (weight, bias) of the 1x1 / 3x3
    convs of level i.  Returns [P2, ..., P5, P6]."""
    prev = F.conv2d(feats[-1], *lateral[-1])
    results = [F.conv2d(prev, *output[-1], padding=1)]
    for i in range(len(feats) - 2, -1, -1):
        top_down = F.interpolate(prev, scale_factor=2.0, mode="nearest")
        lat = F.conv2d(feats[i], *lateral[i])
        prev = lat + top_down[:, :, :lat.shape[2], :lat.shape[3]]
        results.insert(0, F.conv2d(prev, *output[i], padding=1))
    if top_block == "maxpool":
        results.append(last_level_maxpool(results[-1]))
    return results


def multilevel_proposals(objs, deltas, cell_anchors, strides, image_shapes, pre_topk, post_topk, nms_thresh, min_size=0.0,
                         weights=(1.0, 1.0, 1.0, 1.0), offset=0.0):
    """find_top_rpn_proposals frcnn.py:264-390 over several levels (per-level top-k, concat, clip, size filter, batched NMS
    with level ids, first post_topk), fed by per-level RPNOutputs decoding (:758-780).  objs[l] [N,A,Hl,Wl], deltas[l]
    [N,4A,Hl,Wl], cell_anchors[l] [A,4].  Returns per image (boxes [<=R,4], logits [<=R])."""
    from .frcnn_oracle import FRCNNOracle, argsort_desc, batched_nms
    N = objs[0].shape[0]
    tk_boxes, tk_scores, lvl_ids = [], [], []
    for li, (obj, dlt, base, stride) in enumerate(zip(objs, deltas, cell_anchors, strides)):
        _, A, Hf, Wf = obj.shape
        sx = torch.arange(offset * stride, Wf * stride, step=stride, dtype=torch.float32)
        sy = torch.arange(offset * stride, Hf * stride, step=stride, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
        anchors = (shifts.view(-1, 1, 4) + torch.as_tensor(base).float().view(1, -1, 4)).reshape(-1, 4)
        d = dlt.view(N, A, 4, Hf, Wf).permute(0, 3, 4, 1, 2).reshape(-1, 4)
        props = FRCNNOracle.apply_deltas(d, anchors.unsqueeze(0).expand(N, -1, -1).reshape(-1, 4), weights).view(N, -1, 4)
        logits = obj.permute(0, 2, 3, 1).reshape(N, -1)
        k = min(pre_topk, logits.shape[1])
        b_l, s_l = [], []
        for n in range(N):
            order = argsort_desc(logits[n])[:k]
            b_l.append(props[n][order])
            s_l.append(logits[n][order])
        tk_boxes.append(torch.stack(b_l))
        tk_scores.append(torch.stack(s_l))
        lvl_ids.append(torch.full((k,), li, dtype=torch.int64))
    boxes_all, scores_all, lvl_all = torch.cat(tk_boxes, 1), torch.cat(tk_scores, 1), torch.cat(lvl_ids)
    res = []
    for n in range(N):
        boxes, scores, lvl = boxes_all[n].clone(), scores_all[n], lvl_all
        FRCNNOracle.clip_box(boxes, image_shapes[n])
        keep = ((boxes[:, 2] - boxes[:, 0]) > min_size) & ((boxes[:, 3] - boxes[:, 1]) > min_size)
        if int(keep.sum()) != len(boxes):
            boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
        k = batched_nms(boxes, scores, lvl, nms_thresh)[:post_topk]
        res.append((boxes[k], scores[k]))
    return res


# ---------------------------------------------------------------------------------------------------------------------
# The FPN detector end to end (build extension; PARITY UNPINNED vs the reference, which has no FPN model).  Composition:
# detectron2's standard ResNet-FPN Faster R-CNN -- bottom-up ResNet res2..res5 (the reference's own BottleneckBlock
# :903-979 / BasicStem :857-888 arithmetic, restated in FRCNNOracle), FPN neck (above), one RPNHead :1513-1572 shared by the
# levels, find_top_rpn_proposals :264-390 over the levels (above), ROIPooler's level loop :1200-1224 with RoIAlign,
# a 2-FC box head, and the reference's own FastRCNNOutputLayers :1676-1740 / ROIOutputs :1227-1302 on its features.
# ---------------------------------------------------------------------------------------------------------------------
from collections import OrderedDict            # noqa: E402

from .frcnn_oracle import FRCNNOracle, _h      # noqa: E402


class FPNDetectorOracle(FRCNNOracle):
    STRIDES = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}

    def stem(self, x):                                        # BasicStem frcnn.py:872-879, detectron2 key prefix
        if self.emulate:
            x = _h(x)
        x = self._conv_bn(x, "backbone.bottom_up.stem.conv1", stride=2, padding=3, relu=True)
        if self.cfg.MODEL.MAX_POOL:
            return F.max_pool2d(x, kernel_size=3, stride=2, padding=0, ceil_mode=True)
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)

    def backbone(self, images, return_stages=False):          # ResNet.forward :1076-1090 with res2..res5 as outputs
        x = self.stem(images)
        stages = OrderedDict()
        for si, name in enumerate(("res2", "res3", "res4", "res5")):
            for b in range(self.nblocks[si]):
                first_stride = 1 if si == 0 else 2            # frcnn.py:237
                x = self.bottleneck(x, f"backbone.bottom_up.{name}.{b}", first_stride if b == 0 else 1)
            stages[name] = x
        return stages

    def _conv_b(self, x, prefix, padding=0):
        return self._conv_bias(x, prefix, padding=padding, relu=False)

    def neck(self, stages):
        """[C2..C5] -> [P2..P5, P6] (fpn_neck above, with the fast mode's f16 storage roundings when emulating)."""
        feats = [stages[k] for k in ("res2", "res3", "res4", "res5")]
        q = _h if self.emulate else (lambda t: t)
        prev = self._conv_b(feats[-1], "backbone.fpn_lateral5")
        results = [self._conv_b(prev, "backbone.fpn_output5", padding=1)]
        for i, lvl in zip((2, 1, 0), (4, 3, 2)):
            top_down = F.interpolate(prev, scale_factor=2.0, mode="nearest")
            lat = self._conv_b(feats[i], f"backbone.fpn_lateral{lvl}")
            prev = q(lat + top_down[:, :, :lat.shape[2], :lat.shape[3]])
            results.insert(0, self._conv_b(prev, f"backbone.fpn_output{lvl}", padding=1))
        results.append(last_level_maxpool(results[-1]))
        return results

    def rpn_heads(self, pyramid):
        return [self.rpn_head(p) for p in pyramid]            # one head, every level (RPNHead.forward :1561-1572)

    def proposals(self, heads, image_shapes):
        cfg = self.cfg
        n = len(heads)
        cells = [self.sd[f"proposal_generator.anchor_generator.cell_anchors.{i}"] for i in range(n)]
        strides = [4 * 2 ** i for i in range(n)]
        return multilevel_proposals([h[0] for h in heads], [h[1] for h in heads], cells, strides, image_shapes,
                                    cfg.RPN.PRE_NMS_TOPK_TEST, cfg.RPN.POST_NMS_TOPK_TEST, cfg.RPN.NMS_THRESH,
                                    cfg.PROPOSAL_GENERATOR.MIN_SIZE, cfg.RPN.BBOX_REG_WEIGHTS, cfg.ANCHOR_GENERATOR.OFFSET)

    def box_pool(self, pyramid, proposal_boxes):
        rois = torch.cat([torch.cat((torch.full((len(b), 1), float(i)), b), dim=1) for i, b in enumerate(proposal_boxes)], dim=0)
        P, sr = self.cfg.ROI_BOX_HEAD.POOLER_RESOLUTION, self.cfg.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO
        pooled, lv = multilevel_pool(pyramid[:4], [1 / 4, 1 / 8, 1 / 16, 1 / 32], rois, P, "align", sr, True)
        return (_h(pooled) if self.emulate else pooled), lv

    def box_head(self, pooled):
        """flatten (c, y, x) -> fc1 -> ReLU -> fc2 -> ReLU (detectron2 FastRCNNConvFCHead with NUM_FC fully connected layers);
        the last layer's output stays f32 (it is `roi_features`)."""
        q = _h if self.emulate else (lambda t: t)
        x = torch.flatten(pooled, start_dim=1)
        nfc = self.cfg.ROI_BOX_HEAD.NUM_FC
        for i in range(nfc):
            p = f"roi_heads.box_head.fc{i + 1}"
            x = F.relu(F.linear(q(x), q(self.sd[p + ".weight"])) + self.sd[p + ".bias"])
            if i + 1 < nfc:
                x = q(x)
        return x

    @torch.no_grad()
    def forward(self, images, image_shapes, scales_yx=None, return_stages=False):
        images = torch.as_tensor(images, dtype=torch.float32)
        image_shapes = [tuple(int(v) for v in s) for s in np.asarray(image_shapes).tolist()]
        st = OrderedDict()
        st["stages"] = self.backbone(images)
        st["pyramid"] = self.neck(st["stages"])
        st["rpn"] = self.rpn_heads(st["pyramid"])
        props = self.proposals(st["rpn"], image_shapes)
        proposal_boxes = [p[0] for p in props]
        st["proposal_boxes"], st["proposal_logits"] = proposal_boxes, [p[1] for p in props]
        st["pooled"], st["levels"] = self.box_pool(st["pyramid"], proposal_boxes)
        feat = self.box_head(st["pooled"])
        st["box_features"] = feat
        obj_logits, attr_logits, box_deltas = self.predictor(feat)
        st["obj_logits"], st["attr_logits"], st["box_deltas"] = obj_logits, attr_logits, box_deltas
        res = self.roi_outputs(obj_logits, attr_logits, box_deltas, proposal_boxes, feat, image_shapes, scales_yx)
        boxes, classes, probs, attrs, attr_probs, feats, ids = map(list, zip(*res))
        out = OrderedDict(obj_ids=classes, obj_probs=probs, attr_ids=attrs, attr_probs=attr_probs, boxes=boxes,
                          preds_per_image=torch.tensor([len(b) for b in boxes]), roi_features=feats)
        st["keep_ids"] = ids
        return (out, st) if return_stages else out
