"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's Faster R-CNN feature-extraction forward
(`/root/reference/vltk/modeling/frcnn.py`, FRCNN.inference :1942-2004), written
from scratch as plain functions over a state_dict in the reference's key
layout.  Only tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline`
leg may import this module; the product path (vltk_amd/) never does and fails
loudly when its HIP library is missing.

Pinning: this restatement is checked (tests/test_oracle_golden.py) against
golden vectors produced by the reference's OWN module loaded under stub
modules in the build container (tools/gen_golden.py -> tests/golden/*.npz).
The three torchvision operators (RoIPool, nms, batched_nms) are absent from
/root/reference and from this image: they are restated in oracle/tv_ops.c
from torchvision's published CPU kernels and are PARITY UNPINNED (the golden
vectors were produced with these same restatements plugged into the reference
module, so they pin everything *around* those three ops, not the ops).

Dense arithmetic (conv / linear / softmax / sort) uses torch CPU ATen ops --
the same provider the reference calls -- in fp32.

`emulate="fp16"` restates the arithmetic of the GPU *fast* mode (fp16 storage,
fp32 accumulate, BatchNorm folded into the convolution) so the fast HIP path
can be compared at tight tolerance; `emulate=None` is the reference's own fp32
arithmetic (unfolded BatchNorm) and is what the golden vectors pin.
"""
import ctypes
import math
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvko.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libvko.so missing: run `make -C oracle`")
        L = ctypes.CDLL(path)
        L.vko_roi_pool.restype = None
        L.vko_nms.restype = ctypes.c_int64
        L.vko_argsort_desc.restype = None
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------
# third-party ops (torchvision) -- see oracle/tv_ops.c
# --------------------------------------------------------------------------
def roi_pool(x, rois, output_size, spatial_scale):
    """torchvision.ops.RoIPool forward (frcnn.py:1179,1198).  x [N,C,H,W] f32, rois [K,5]."""
    x = np.ascontiguousarray(x.detach().cpu().numpy(), dtype=np.float32)
    r = np.ascontiguousarray(rois.detach().cpu().numpy(), dtype=np.float32)
    N, C, H, W = x.shape
    K = r.shape[0]
    PH = PW = int(output_size)
    out = np.zeros((K, C, PH, PW), dtype=np.float32)
    if K:
        _lib().vko_roi_pool(_fp(x), N, C, H, W, _fp(r), K, ctypes.c_float(spatial_scale), PH, PW, _fp(out))
    return torch.from_numpy(out)


def nms(boxes, scores, thr):
    """torchvision.ops.boxes.nms (frcnn.py:132): indices in score order, int64."""
    b = np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float32)
    s = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32)
    n = b.shape[0]
    keep = np.zeros((max(n, 1),), dtype=np.int64)
    k = _lib().vko_nms(_fp(b), _fp(s), ctypes.c_int64(n), ctypes.c_double(float(thr)), _fp(keep))
    return torch.from_numpy(keep[:k].copy())


def batched_nms(boxes, scores, idxs, thr):
    """torchvision.ops.boxes.batched_nms (frcnn.py:383), coordinate-offset form:
    boxes of different levels never overlap after adding idx*(max_coord+1)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1)
    return nms(boxes + offsets[:, None], scores, thr)


def argsort_desc(scores):
    """Stable descending argsort (ties -> lower index): the build's defined tie order."""
    s = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32)
    order = np.zeros((max(s.shape[0], 1),), dtype=np.int64)
    _lib().vko_argsort_desc(_fp(s), ctypes.c_int64(s.shape[0]), _fp(order))
    return torch.from_numpy(order[: s.shape[0]].copy())


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def _h(t):
    """round-trip through fp16 (the fast path's storage type)."""
    return t.to(torch.float16).to(torch.float32)


def fold_bn(w, gamma, beta, mean, var, eps=1e-5):
    """BN(eval) folded into the conv: float64 math -> f32."""
    w64 = w.double()
    s = gamma.double() / torch.sqrt(var.double() + eps)
    return (w64 * s.view(-1, 1, 1, 1)).float(), (beta.double() - mean.double() * s).float()


BLOCKS_PER_STAGE = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


class FRCNNOracle:
    def __init__(self, cfg, state_dict, emulate=None):
        assert emulate in (None, "fp16")
        self.cfg = cfg
        self.emulate = emulate
        self.sd = {k: (torch.from_numpy(np.asarray(v)) if not isinstance(v, torch.Tensor) else v)
                   for k, v in state_dict.items()}
        r = cfg.RESNETS
        self.groups = r.NUM_GROUPS
        self.stride_in_1x1 = r.STRIDE_IN_1X1
        self.nblocks = BLOCKS_PER_STAGE[r.DEPTH]
        # mutable, like the reference's model.roi_outputs.* (frcnn.py:1229-1240, tests/frcnn_test.py:16-19)
        nt = cfg.ROI_HEADS.NMS_THRESH_TEST
        self.nms_thresh = list(nt) if isinstance(nt, (list, tuple)) else [nt]
        self.score_thresh = cfg.ROI_HEADS.SCORE_THRESH_TEST
        self.min_detections = cfg.MIN_DETECTIONS
        self.max_detections = cfg.MAX_DETECTIONS
        self._folded = {}

    # ---- conv + BN (+relu) : Conv2d.forward frcnn.py:794-822 -------------
    def _conv_bn(self, x, prefix, stride=1, padding=0, dilation=1, groups=1, relu=False, residual=None, round_out=True):
        sd = self.sd
        if self.emulate is None:
            y = F.conv2d(x, sd[prefix + ".weight"], None, stride, padding, dilation, groups)
            y = F.batch_norm(y, sd[prefix + ".norm.running_mean"], sd[prefix + ".norm.running_var"],
                             sd[prefix + ".norm.weight"], sd[prefix + ".norm.bias"], False, 0.0, 1e-5)
            if residual is not None:
                y = y + residual          # `out += shortcut` frcnn.py:977
            return F.relu(y) if relu else y
        if prefix not in self._folded:
            w, b = fold_bn(sd[prefix + ".weight"], sd[prefix + ".norm.weight"], sd[prefix + ".norm.bias"],
                           sd[prefix + ".norm.running_mean"], sd[prefix + ".norm.running_var"])
            self._folded[prefix] = (_h(w), b)
        w, b = self._folded[prefix]
        y = F.conv2d(x, w, None, stride, padding, dilation, groups) + b.view(1, -1, 1, 1)
        if residual is not None:
            y = y + residual
        if relu:
            y = F.relu(y)
        return _h(y) if round_out else y

    def _conv_bias(self, x, prefix, padding=0, relu=False, round_out=True):
        w, b = self.sd[prefix + ".weight"], self.sd[prefix + ".bias"]
        if self.emulate is None:
            y = F.conv2d(x, w, b, 1, padding)
            return F.relu(y) if relu else y
        y = F.conv2d(x, _h(w), None, 1, padding) + b.view(1, -1, 1, 1)
        if relu:
            y = F.relu(y)
        return _h(y) if round_out else y

    # ---- BasicStem frcnn.py:872-879 -------------------------------------
    def stem(self, x):
        if self.emulate:
            x = _h(x)
        x = self._conv_bn(x, "backbone.stem.conv1", stride=2, padding=3, relu=True)
        if self.cfg.MODEL.MAX_POOL:   # caffe_maxpool frcnn.py:875-876
            return F.max_pool2d(x, kernel_size=3, stride=2, padding=0, ceil_mode=True)
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)

    # ---- BottleneckBlock.forward frcnn.py:963-979 -----------------------
    def bottleneck(self, x, prefix, stride, dilation=1, stride_shortcut=None):
        s1, s3 = (stride, 1) if self.stride_in_1x1 else (1, stride)   # frcnn.py:932
        out = self._conv_bn(x, prefix + ".conv1", stride=s1, relu=True)
        out = self._conv_bn(out, prefix + ".conv2", stride=s3, padding=dilation, dilation=dilation,
                            groups=self.groups, relu=True)
        if (prefix + ".shortcut.weight") in self.sd:
            ss = stride if stride_shortcut is None else stride_shortcut
            w3, wsc = self.sd[prefix + ".conv3.weight"], self.sd[prefix + ".shortcut.weight"]
            # fp16 emulation of the HIP path: a stride-1 projection shortcut is part of conv3's GEMM there
            # (csrc/model.hip can_fuse_shortcut), so it is never rounded to f16 on its own
            fused = (self.emulate is not None and ss == 1 and w3.shape[0] % 256 == 0 and w3.shape[1] % 32 == 0
                     and wsc.shape[1] % 32 == 0)
            sc = self._conv_bn(x, prefix + ".shortcut", stride=ss, round_out=not fused)
        else:
            sc = x
        return self._conv_bn(out, prefix + ".conv3", relu=True, residual=sc)

    # ---- ResNet.forward frcnn.py:1076-1090 (C4: stem + res2..res4) ------
    def backbone(self, images, return_stages=False):
        x = self.stem(images)
        stages = OrderedDict(stem=x)
        for si, name in enumerate(("res2", "res3", "res4")):
            for b in range(self.nblocks[si]):
                first_stride = 1 if si == 0 else 2          # frcnn.py:237
                x = self.bottleneck(x, f"backbone.{name}.{b}", first_stride if b == 0 else 1)
            stages[name] = x
        return stages if return_stages else x

    # ---- RPNHead.forward frcnn.py:1561-1572 -----------------------------
    def rpn_head(self, feat):
        t = self._conv_bias(feat, "proposal_generator.rpn_head.conv", padding=1, relu=True)
        obj = self._conv_bias(t, "proposal_generator.rpn_head.objectness_logits", round_out=False)
        dlt = self._conv_bias(t, "proposal_generator.rpn_head.anchor_deltas", round_out=False)
        return obj, dlt

    # ---- AnchorGenerator.grid_anchors frcnn.py:1463-1477, :176-197 ------
    def grid_anchors(self, Hf, Wf, stride=16):
        base = self.sd["proposal_generator.anchor_generator.cell_anchors.0"].float()
        off = self.cfg.ANCHOR_GENERATOR.OFFSET
        sx = torch.arange(off * stride, Wf * stride, step=stride, dtype=torch.float32)
        sy = torch.arange(off * stride, Hf * stride, step=stride, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        xx, yy = xx.reshape(-1), yy.reshape(-1)
        shifts = torch.stack((xx, yy, xx, yy), dim=1)
        return (shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)

    # ---- Box2BoxTransform.apply_deltas frcnn.py:548-584 -----------------
    @staticmethod
    def apply_deltas(deltas, boxes, weights):
        clamp = math.log(1000.0 / 16)                       # frcnn.py:510
        boxes = boxes.to(deltas.dtype)
        widths = boxes[:, 2] - boxes[:, 0]
        heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = weights
        dx = deltas[:, 0::4] / wx
        dy = deltas[:, 1::4] / wy
        dw = torch.clamp(deltas[:, 2::4] / ww, max=clamp)
        dh = torch.clamp(deltas[:, 3::4] / wh, max=clamp)
        pcx = dx * widths[:, None] + ctr_x[:, None]
        pcy = dy * heights[:, None] + ctr_y[:, None]
        pw = torch.exp(dw) * widths[:, None]
        ph = torch.exp(dh) * heights[:, None]
        out = torch.zeros_like(deltas)
        out[:, 0::4] = pcx - 0.5 * pw
        out[:, 1::4] = pcy - 0.5 * ph
        out[:, 2::4] = pcx + 0.5 * pw
        out[:, 3::4] = pcy + 0.5 * ph
        return out

    @staticmethod
    def clip_box(t, hw):
        """_clip_box frcnn.py:147-153 (in place); hw = (h, w)."""
        assert torch.isfinite(t).all(), "Box tensor contains infinite or NaN!"
        h, w = float(hw[0]), float(hw[1])
        t[:, 0].clamp_(min=0, max=w)
        t[:, 1].clamp_(min=0, max=h)
        t[:, 2].clamp_(min=0, max=w)
        t[:, 3].clamp_(min=0, max=h)

    # ---- RPNOutputs.predict_* frcnn.py:748-781 + find_top_rpn_proposals :264-390 + RPN.inference :1615-1638
    def rpn_proposals(self, obj, dlt, image_shapes):
        """obj [N,A,H,W], dlt [N,4A,H,W] -> list of ([<=R,4] boxes, [<=R] logits) per image."""
        cfg = self.cfg
        N, A, Hf, Wf = obj.shape
        anchors = self.grid_anchors(Hf, Wf)                                   # [HWA,4]
        d = dlt.view(N, A, 4, Hf, Wf).permute(0, 3, 4, 1, 2).reshape(-1, 4)   # frcnn.py:758-762
        anc = anchors.unsqueeze(0).expand(N, -1, -1).reshape(-1, 4)
        props = self.apply_deltas(d, anc, cfg.RPN.BBOX_REG_WEIGHTS).view(N, -1, 4)
        logits = obj.permute(0, 2, 3, 1).reshape(N, -1)                       # frcnn.py:776-780
        pre = min(cfg.RPN.PRE_NMS_TOPK_TEST, logits.shape[1])
        post = cfg.RPN.POST_NMS_TOPK_TEST
        res = []
        for n in range(N):
            order = argsort_desc(logits[n])[:pre]          # sort desc, defined tie order (frcnn.py:304-306)
            boxes = props[n][order].clone()
            scores = logits[n][order]
            self.clip_box(boxes, image_shapes[n])          # frcnn.py:369
            w = boxes[:, 2] - boxes[:, 0]
            h = boxes[:, 3] - boxes[:, 1]
            ms = cfg.PROPOSAL_GENERATOR.MIN_SIZE
            keep = (w > ms) & (h > ms)                      # _nonempty_boxes frcnn.py:156-160
            if int(keep.sum()) != len(boxes):
                boxes, scores = boxes[keep], scores[keep]
            lvl = torch.zeros(len(boxes), dtype=torch.int64)
            k = batched_nms(boxes, scores, lvl, cfg.RPN.NMS_THRESH)[:post]    # frcnn.py:383-384
            res.append((boxes[k], scores[k]))               # RPN.inference's re-sort :1633 is the identity here
        return res

    # ---- ROIPooler.forward frcnn.py:1183-1198 ---------------------------
    def pool(self, feat, proposal_boxes):
        rois = torch.cat([torch.cat((torch.full((len(b), 1), float(i)), b), dim=1)
                          for i, b in enumerate(proposal_boxes)], dim=0)      # frcnn.py:426-441
        res = self.cfg.ROI_BOX_HEAD.POOLER_RESOLUTION
        return roi_pool(feat, rois, res, 1.0 / 16)

    # ---- Res5ROIHeads frcnn.py:1344-1355, 1387-1403 ---------------------
    def res5(self, x):
        halve = self.cfg.ROI_BOX_HEAD.RES5HALVE
        for b in range(3):
            if halve:
                x = self.bottleneck(x, f"roi_heads.res5.{b}", 2 if b == 0 else 1)
            else:  # VG: stride 1 everywhere, conv2 dilation/padding 2
                x = self.bottleneck(x, f"roi_heads.res5.{b}", 1, dilation=2)
        return x

    # ---- FastRCNNOutputLayers.forward frcnn.py:1726-1740 ----------------
    def predictor(self, feats):
        sd, p = self.sd, "roi_heads.box_predictor."
        # the HIP path keeps the predictor in fp32 in both modes (csrc/model.hip vk_handle::pdt): nothing to emulate
        q = (lambda t: t)
        f = q(feats)
        scores = F.linear(f, q(sd[p + "cls_score.weight"])) + sd[p + "cls_score.bias"]
        deltas = F.linear(f, q(sd[p + "bbox_pred.weight"])) + sd[p + "bbox_pred.bias"]
        max_class = scores.argmax(-1)
        emb = q(sd[p + "cls_embedding.weight"])[max_class]
        h = F.linear(torch.cat([f, emb], -1), q(sd[p + "fc_attr.weight"])) + sd[p + "fc_attr.bias"]
        h = q(F.relu(h))
        attr = F.linear(h, q(sd[p + "attr_score.weight"])) + sd[p + "attr_score.bias"]
        return scores, attr, deltas

    # ---- do_nms frcnn.py:116-143 ----------------------------------------
    @staticmethod
    def do_nms(boxes, scores, image_shape, nms_thresh, mind, maxd):
        scores = scores[:, :-1]
        C = boxes.shape[1] // 4
        boxes = boxes.reshape(-1, 4).clone()
        FRCNNOracle.clip_box(boxes, image_shape)
        max_scores, max_classes = scores.max(1)
        idxs = torch.arange(scores.shape[0]) * C + max_classes
        max_boxes = boxes[idxs]
        keep = nms(max_boxes, max_scores, nms_thresh)[:maxd]
        stop = mind <= keep.shape[-1] <= maxd
        return stop, max_boxes[keep], max_scores[keep], max_classes[keep], keep

    # ---- ROIOutputs.inference frcnn.py:1262-1294 ------------------------
    def roi_outputs(self, obj_logits, attr_logits, box_deltas, proposal_boxes, features, sizes, scales=None):
        ppi = [len(p) for p in proposal_boxes]
        K = box_deltas.shape[0]
        C = box_deltas.shape[1] // 4
        props = torch.cat(proposal_boxes, 0).unsqueeze(-2).expand(K, C, 4).reshape(-1, 4)
        boxes = self.apply_deltas(box_deltas.reshape(K * C, 4), props,
                                  self.cfg.ROI_BOX_HEAD.BBOX_REG_WEIGHTS).view(K, C * 4).split(ppi, 0)
        probs = F.softmax(obj_logits, dim=-1).split(ppi, 0)
        ap, aid = attr_logits[..., :-1].softmax(-1).max(-1)
        ap, aid = ap.split(ppi, 0), aid.split(ppi, 0)
        feats = features.split(ppi, 0)
        out = []
        for i in range(len(ppi)):
            for t in self.nms_thresh:
                stop, mb, ms, cls, ids = self.do_nms(boxes[i], probs[i], sizes[i], t,
                                                     self.min_detections, self.max_detections)
                if stop:
                    break
            if scales is not None:
                mb = mb.clone()
                mb[:, 0::2] *= scales[i][1]
                mb[:, 1::2] *= scales[i][0]
            out.append((mb, cls, ms, aid[i][ids], ap[i][ids], feats[i][ids], ids))
        return out

    # ---- FRCNN.inference frcnn.py:1942-2004 ------------------------------
    @torch.no_grad()
    def forward(self, images, image_shapes, scales_yx=None, return_stages=False):
        images = torch.as_tensor(images, dtype=torch.float32)
        image_shapes = [tuple(int(v) for v in s) for s in np.asarray(image_shapes).tolist()]
        st = OrderedDict()
        st["res4"] = self.backbone(images)
        obj, dlt = self.rpn_head(st["res4"])
        st["rpn_objectness"], st["rpn_deltas"] = obj, dlt
        props = self.rpn_proposals(obj, dlt, image_shapes)
        proposal_boxes = [p[0] for p in props]
        st["proposal_boxes"], st["proposal_logits"] = proposal_boxes, [p[1] for p in props]
        pooled = self.pool(st["res4"], proposal_boxes)
        st["pooled"] = pooled
        box_feat = self.res5(pooled)
        feature_pooled = box_feat.mean(dim=[2, 3])                 # frcnn.py:1401
        st["feature_pooled"] = feature_pooled
        obj_logits, attr_logits, box_deltas = self.predictor(feature_pooled)
        st["obj_logits"], st["attr_logits"], st["box_deltas"] = obj_logits, attr_logits, box_deltas
        res = self.roi_outputs(obj_logits, attr_logits, box_deltas, proposal_boxes, feature_pooled,
                               image_shapes, scales_yx)
        boxes, classes, probs, attrs, attr_probs, feats, ids = map(list, zip(*res))
        out = OrderedDict(obj_ids=classes, obj_probs=probs, attr_ids=attrs, attr_probs=attr_probs,
                          boxes=boxes, preds_per_image=torch.tensor([len(b) for b in boxes]),
                          roi_features=feats)
        st["keep_ids"] = ids
        return (out, st) if return_stages else out
