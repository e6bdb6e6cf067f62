#!/usr/bin/env python3
"""One-time head calibration for the synthetic weights (build container; CPU).

Runs the ORACLE forward (test infrastructure) on a fixed synthetic batch with
the uncalibrated base weights and records, per prediction head, the gain and
bias that make its output zero-mean with the target std (see
vltk_amd.weights.HEAD_TARGET_STD).  Output: vltk_amd/data/head_calib_*.npz.

    python tools/calibrate_weights.py [--depth 101] [--groups 1] [--width 64] [--seed 1234]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.frcnn_oracle import FRCNNOracle          # noqa: E402
from vltk_amd.config import vg_c4_config             # noqa: E402
from vltk_amd import weights as W                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=101)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--groups", type=int, default=1, help="RESNETS.NUM_GROUPS (ResNeXt)")
    ap.add_argument("--width", type=int, default=64, help="RESNETS.WIDTH_PER_GROUP")
    ap.add_argument("--fpn", action="store_true", help="the FPN detector (vltk_amd.config.fpn_config)")
    a = ap.parse_args()
    if a.fpn:
        return main_fpn(a)
    cfg = vg_c4_config(depth=a.depth, num_groups=a.groups, width_per_group=a.width, post_nms_topk=64, detections=36)
    sd = W.make_state_dict(cfg, a.seed, calibrated=False)
    o = FRCNNOracle(cfg, sd)
    x = torch.from_numpy(W.synthetic_images(2, 192, 256, seed=a.seed + 1))
    out = {}

    def calib(head, y, reduce_dims):
        """y: raw head output with zero bias; channel dim = 1."""
        mean = y.mean(dim=reduce_dims)
        gain = W.HEAD_TARGET_STD[head] / float((y - mean.view(1, -1, *([1] * (y.dim() - 2)))).std())
        out[head + ".gain"] = np.float32(gain)
        out[head + ".bias"] = (-(mean * gain)).numpy().astype(np.float32)
        return gain

    with torch.no_grad():
        res4 = o.backbone(x)
        p = "proposal_generator.rpn_head."
        t = F.relu(F.conv2d(res4, o.sd[p + "conv.weight"], o.sd[p + "conv.bias"], 1, 1))
        for h in ("objectness_logits", "anchor_deltas"):
            calib(p + h, F.conv2d(t, o.sd[p + h + ".weight"]), (0, 2, 3))
        # proposals for the RoI heads: a fixed grid of boxes of mixed sizes (no dependence on the RPN heads)
        g = np.random.Generator(np.random.PCG64(a.seed))
        boxes = []
        for n in range(2):
            xy = g.uniform(0, [200, 140], size=(48, 2))
            wh = g.uniform(16, 120, size=(48, 2))
            b = np.concatenate([xy, np.minimum(xy + wh, [256, 192])], 1).astype(np.float32)
            boxes.append(torch.from_numpy(b))
        f = o.res5(o.pool(res4, boxes)).mean(dim=[2, 3])
        q = "roi_heads.box_predictor."
        g_cls = calib(q + "cls_score", F.linear(f, o.sd[q + "cls_score.weight"]), (0,))
        calib(q + "bbox_pred", F.linear(f, o.sd[q + "bbox_pred.weight"]), (0,))
        scores = F.linear(f, o.sd[q + "cls_score.weight"]) * g_cls + torch.from_numpy(out[q + "cls_score.bias"])
        emb = o.sd[q + "cls_embedding.weight"][scores.argmax(-1)]
        hid = F.relu(F.linear(torch.cat([f, emb], -1), o.sd[q + "fc_attr.weight"], o.sd[q + "fc_attr.bias"]))
        calib(q + "attr_score", F.linear(hid, o.sd[q + "attr_score.weight"]), (0,))
    os.makedirs(os.path.dirname(W.calib_path(cfg, a.seed)), exist_ok=True)
    np.savez(W.calib_path(cfg, a.seed), **out)
    print("wrote", W.calib_path(cfg, a.seed), {k: float(v) for k, v in out.items() if k.endswith("gain")})


def main_fpn(a):
    from oracle.fpn_oracle import FPNDetectorOracle
    from vltk_amd.config import fpn_config
    cfg = fpn_config(depth=a.depth, num_groups=a.groups, width_per_group=a.width, post_nms_topk=64, detections=36)
    sd = W.make_state_dict(cfg, a.seed, calibrated=False)
    o = FPNDetectorOracle(cfg, sd)
    x = torch.from_numpy(W.synthetic_images(2, 192, 256, seed=a.seed + 1))
    out = {}

    def calib(head, ys, channel_dim):
        """ys: raw head outputs (zero bias) of every level, [*, C, *]; one gain for the head, one bias per channel."""
        flat = torch.cat([y.transpose(channel_dim, -1).reshape(-1, y.shape[channel_dim]) for y in ys], 0)
        mean = flat.mean(0)
        gain = W.HEAD_TARGET_STD[head] / float((flat - mean).std())
        out[head + ".gain"] = np.float32(gain)
        out[head + ".bias"] = (-(mean * gain)).numpy().astype(np.float32)
        return gain

    with torch.no_grad():
        pyr = o.neck(o.backbone(x))
        p = "proposal_generator.rpn_head."
        ts = [F.relu(F.conv2d(f, o.sd[p + "conv.weight"], o.sd[p + "conv.bias"], 1, 1)) for f in pyr]
        for h in ("objectness_logits", "anchor_deltas"):
            calib(p + h, [F.conv2d(t, o.sd[p + h + ".weight"]) for t in ts], 1)
        g = np.random.Generator(np.random.PCG64(a.seed))
        boxes = []
        for n in range(2):        # boxes of mixed sizes so that every pyramid level is used
            xy = g.uniform(0, [200, 140], size=(64, 2))
            wh = np.exp(g.uniform(np.log(8), np.log(250), size=(64, 2)))
            b = np.concatenate([xy, np.minimum(xy + wh, [256, 192])], 1).astype(np.float32)
            boxes.append(torch.from_numpy(b))
        pooled, lv = o.box_pool(pyr, boxes)
        f = o.box_head(pooled)
        q = "roi_heads.box_predictor."
        g_cls = calib(q + "cls_score", [F.linear(f, o.sd[q + "cls_score.weight"])], 1)
        calib(q + "bbox_pred", [F.linear(f, o.sd[q + "bbox_pred.weight"])], 1)
        scores = F.linear(f, o.sd[q + "cls_score.weight"]) * g_cls + torch.from_numpy(out[q + "cls_score.bias"])
        emb = o.sd[q + "cls_embedding.weight"][scores.argmax(-1)]
        hid = F.relu(F.linear(torch.cat([f, emb], -1), o.sd[q + "fc_attr.weight"], o.sd[q + "fc_attr.bias"]))
        calib(q + "attr_score", [F.linear(hid, o.sd[q + "attr_score.weight"])], 1)
    np.savez(W.calib_path(cfg, a.seed), **out)
    print("wrote", W.calib_path(cfg, a.seed), {k: float(v) for k, v in out.items() if k.endswith("gain")},
          "levels used", lv.bincount().tolist())


if __name__ == "__main__":
    main()
