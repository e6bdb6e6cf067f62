#!/usr/bin/env python3
"""Timing of the N4 ops at the bench geometry (needs a GPU): FPN neck over C2..C5 of 32 images of 800x1333 (fp16) and
multi-level RoIAlign of 32 x 1000 RoIs from P2..P5 (256 channels, 7x7, sampling ratio 2)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vltk_amd.fpn import FPNNeck, MultiLevelRoIAlign  # noqa: E402

dev = torch.device("cuda:0")
gen = np.random.Generator(np.random.PCG64(0))
B, Cc = 32, 256
chans, sizes = [256, 512, 1024, 2048], [(200, 334), (100, 167), (50, 84), (25, 42)]
feats = [torch.randn((B, h, w, c), device=dev).half() for c, (h, w) in zip(chans, sizes)]
lat = [((gen.standard_normal((Cc, c, 1, 1)) * (1.0 / c) ** 0.5).astype(np.float32), np.zeros(Cc, np.float32)) for c in chans]
out = [((gen.standard_normal((Cc, Cc, 3, 3)) * (1.0 / (9 * Cc)) ** 0.5).astype(np.float32), np.zeros(Cc, np.float32)) for _ in chans]
neck = FPNNeck(lat, out, precision="fp16")


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, r


t, ps = timeit(lambda: neck(feats))
px = sum(h * w for h, w in sizes)
gf = B * (2 * sum(h * w * c * Cc for c, (h, w) in zip(chans, sizes)) + 2 * px * Cc * 9 * Cc) / 1e9
print(f"FPN neck, {B} images: {t * 1e3:.2f} ms  ({gf / t / 1e3:.0f} TFLOP/s, {gf / B:.1f} GFLOP/image algorithmic)")
K = B * 1000
xy = gen.uniform(0, [1100, 650], (K, 2))
wh = np.exp(gen.uniform(np.log(16), np.log(600), (K, 2)))
rois = torch.from_numpy(np.concatenate([gen.integers(0, B, (K, 1)), xy, xy + wh], 1).astype(np.float32)).to(dev)
pool = MultiLevelRoIAlign(7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, True, precision="fp16")
t, (o, lv) = timeit(lambda: pool(ps[:4], rois))
wr = o.numel() * 2
print(f"RoIAlign, {K} RoIs x 7x7 x {Cc}: {t * 1e3:.2f} ms  ({wr / t / 1e9:.0f} GB/s written, {wr * 17 / t / 1e9:.0f} GB/s incl. the 16 taps read per output; "
      f"levels {np.bincount(lv.cpu().numpy()).tolist()})")
