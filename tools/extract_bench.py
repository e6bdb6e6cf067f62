#!/usr/bin/env python3
"""End-to-end extraction rate on one MI355X, host side included (BASELINE config 3 shape, one rank's share):
raw uint8 images in host memory -> PCIe -> GPU resize/normalise/pad to 800x1333 -> FRCNN (fp16, R = 300, 36 detections)
-> device-to-host -> Arrow IPC file, through vltk_amd.pipeline.ExtractionPipeline (loader and writer threads).
usage: python tools/extract_bench.py [n_images=512] [batch=32]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vltk_amd import FRCNN, make_state_dict, vg_c4_config  # noqa: E402
from vltk_amd.pipeline import ExtractionPipeline  # noqa: E402
from vltk_amd.preprocess import Preprocess  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = vg_c4_config(post_nms_topk=300, detections=36, device="cuda:0")
model = FRCNN(cfg, precision="fp16").load_state_dict(make_state_dict(cfg, seed=1234)).eval()
pre = Preprocess(cfg)
g = np.random.Generator(np.random.PCG64(0xF2C))
pool = [g.integers(0, 256, (480, 640, 3), dtype=np.uint8) for _ in range(64)]      # GQA/VG-like raw size, 0.9 MB each
items = [(f"{i}", pool[i % 64]) for i in range(n)]
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    warm = ExtractionPipeline(model, pre, os.path.join(d, "warm.arrow"), batch_size=B)
    warm.run(items[:2 * B])
    torch.cuda.synchronize()
    pipe = ExtractionPipeline(model, pre, os.path.join(d, "train.arrow"), batch_size=B, dataset="synthetic")
    t0 = time.perf_counter()
    path = pipe.run(items)
    dt = time.perf_counter() - t0
    size = os.path.getsize(path)
print(f"pipeline: {n} raw 480x640 uint8 images -> {path.split('/')[-1]} ({size / 1e6:.0f} MB) in {dt:.2f} s = {n / dt:.1f} images/s "
      f"(batch {B}; upload + GPU pre-processing + forward + read-back + Arrow write; decode excluded)")
