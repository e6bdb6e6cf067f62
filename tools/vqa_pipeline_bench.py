#!/usr/bin/env python3
"""BASELINE config 5 end to end on one MI355X: 32 synthetic 800x1333 images -> FRCNN (fp16, R = 300, 36 detections) ->
[32,36,2048] features + normalised boxes -> LXMERT question answering (bf16, default 9/5/5 geometry) -> answer ids.
Everything between the image tensor and the answer logits runs in libvltk_hip.so; seeded synthetic weights."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vltk_amd import FRCNN, make_state_dict, synthetic_images, vg_c4_config  # noqa: E402
from vltk_amd.lxmert import LxmertForQuestionAnswering, lxmert_config, make_lxmert_qa_state_dict  # noqa: E402

B, D, NQA = 32, 36, 3129
fcfg = vg_c4_config(post_nms_topk=300, detections=D, device="cuda:0")
det = FRCNN(fcfg).load_state_dict(make_state_dict(fcfg, seed=1234)).eval()
lcfg = lxmert_config()
qa = LxmertForQuestionAnswering(lcfg, NQA, precision="bf16").load_state_dict(make_lxmert_qa_state_dict(lcfg, NQA, 1))
images = torch.from_numpy(synthetic_images(B, 800, 1333, seed=0xF2C)).cuda()
shapes = torch.tensor([[800, 1333]] * B)
ids = torch.from_numpy(np.random.Generator(np.random.PCG64(0)).integers(1, lcfg["vocab_size"], (B, 20))).cuda()


def step():
    out = det(images, shapes, padding="max_detections", max_detections=D, return_tensors="pt", location="cuda")
    feats, boxes = out["roi_features"], out["normalized_boxes"]
    vmask = (torch.arange(D, device="cuda")[None] < out["preds_per_image"].cuda()[:, None]).float()
    return qa(ids, feats, boxes, visual_attention_mask=vmask)


for _ in range(2):
    score = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    score = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
assert score.shape == (B, NQA) and torch.isfinite(score.float()).all()
print(f"images -> FRCNN -> LXMERT-QA: {dt * 1e3:.1f} ms per batch of {B}  = {B / dt:.0f} images/s (answers {score.float().argmax(-1)[:6].tolist()} ...)")
