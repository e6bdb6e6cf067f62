#!/usr/bin/env python3
"""res2 bottleneck blocks at bench size (32 x 200 x 333): the fused kernel (vk_bottleneck64) against the same block from the
layer-by-layer kernels, interleaved in one process on one device.  usage: python tools/bneck_bench.py [batch=32] [reps=10]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G                                   # noqa: E402
from vltk_amd import _lib as L                        # noqa: E402
import test_gpu_bneck_fused as T                       # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H, W = 200, 333
for proj in (False, True):
    cin = 64 if proj else 256
    g = np.random.Generator(np.random.PCG64(1))
    xd = torch.from_numpy(g.standard_normal((B, H, W, cin)).astype(np.float16)).to(G.DEV).relu_()
    p = T.make_block(3, cin, proj)
    w1d, b1d, w2d, b2d, w3d, b3d = T.packed(p, proj)
    t1 = torch.empty((B, H, W, 64), dtype=torch.float16, device=G.DEV)
    t2 = torch.empty_like(t1)
    y = torch.empty((B, H, W, 256), dtype=torch.float16, device=G.DEV)
    y2 = torch.empty_like(y)
    dt = L.VK_F16

    def fused():
        L.call("vk_bottleneck64", G.P(xd), B, H, W, cin, int(proj), G.P(w1d), G.P(b1d), G.P(w2d), G.P(b2d), G.P(w3d), G.P(b3d), G.P(y), G.stream())

    def layers():
        L.call("vk_conv2d", G.P(xd), B, H, W, cin, G.P(w1d), G.P(b1d), None, G.P(t1), 64, 64, 1, 1, 1, 0, 1, 1, 1, dt, dt, G.stream())
        L.call("vk_conv2d", G.P(t1), B, H, W, 64, G.P(w2d), G.P(b2d), None, G.P(t2), 64, 64, 3, 3, 1, 1, 1, 1, 1, dt, dt, G.stream())
        if proj:
            L.call("vk_conv1x1_dual", G.P(t2), 64, G.P(xd), cin, B * H * W, G.P(w3d), G.P(b3d), None, G.P(y2), 256, 1, G.stream())
        else:
            L.call("vk_conv2d", G.P(t2), B, H, W, 64, G.P(w3d), G.P(b3d), G.P(xd), G.P(y2), 256, 256, 1, 1, 1, 0, 1, 1, 1, dt, dt, G.stream())

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3
    for _ in range(3):
        fused(), layers()
    torch.cuda.synchronize()
    tf, tl, tt = [], [], []
    for _ in range(reps):
        os.environ["VK_BNECK_ROWS"] = "1"
        tf.append(timed(fused))
        os.environ["VK_BNECK_ROWS"] = "0"
        tt.append(timed(fused))
        tl.append(timed(layers))
    os.environ["VK_BNECK_ROWS"] = "1"
    fused()
    px = B * H * W
    byt = px * (cin * 2 + 512)
    print(f"res2 block {'0 (projection, cin 64)' if proj else '1/2 (identity, cin 256)'} batch {B}: fused rows {np.median(tf):8.1f} us (min {min(tf):8.1f}) = "
          f"{byt / np.median(tf) / 1e6:.2f} TB/s of x-in + y-out | fused tiles {np.median(tt):8.1f} us | layer by layer {np.median(tl):8.1f} us (min {min(tl):8.1f}) | equal: {torch.equal(y, y2)}")
