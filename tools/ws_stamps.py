#!/usr/bin/env python3
"""Where a wave of the weight-stationary 1x1 conv kernel spends its cycles (diagnostic; needs a GPU).

One stamped launch (VK_WS_STAMPS, conv_ws.hip DBG & 32) of the Res5 conv3 shape after warm-up launches; prints the median
per-wave cycle sums: waiting for DMA, at the stage barrier, stage body (fragment reads + MFMAs + DMA issue), waiting for the
residual rows, epilogue.
usage: VK_WS_WAVES=4|8 python tools/ws_stamps.py [shape]

Needs the tools build of the library (make -C vltk_amd/csrc clean && make -C vltk_amd/csrc -j8 ABLATION=1): the shipped
build has no stamp / ablation instantiations and ignores the VK_*_STAMPS / VK_*_DBG variables.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpu_util as G  # noqa: E402
from vltk_amd import _lib as L  # noqa: E402
from conv_bench import SHAPES  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "head_conv3"
    out = "/tmp/ws_stamps.txt"
    if os.path.exists(out):
        os.remove(out)
    N, H, W, cin, cout, k, stride, pad, dil, use_res = SHAPES[name]
    g = np.random.Generator(np.random.PCG64(0))
    w = (g.standard_normal((cout, cin, k, k)) * (2.0 / (cin * k * k)) ** 0.5).astype(np.float32)
    wd, bd = G.pack_conv(w, None, np.zeros(cout, np.float32), L.VK_F16)
    x = torch.randn((N, H, W, cin), device=G.DEV).half()
    y = torch.empty((N, H, W, cout), dtype=torch.float16, device=G.DEV)
    res = torch.randn((N, H, W, cout), device=G.DEV).half() if use_res else None

    def run():
        L.call("vk_conv2d", G.P(x), N, H, W, cin, G.P(wd), G.P(bd), G.P(res), G.P(y), cout, cout, k, k, stride, pad, dil, 1, 1,
               L.VK_F16, L.VK_F16, G.stream())
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    os.environ["VK_WS_STAMPS"] = out
    run()
    torch.cuda.synchronize()
    del os.environ["VK_WS_STAMPS"]
    rows = np.array([[int(v) for v in ln.split()] for ln in open(out) if not ln.startswith("#")], dtype=np.float64)
    rows = rows[rows[:, 8] > 0]
    names = ["whole wave", "wait for DMA", "stage barrier", "stage body", "wait for residual", "epilogue"]
    tiles = np.median(rows[:, 8])
    span_us = (rows[:, 9].max() - rows[:, 9].min()) / 100.0
    print(f"{name}: waves/workgroup {os.environ.get('VK_WS_WAVES', '4')}, {len(rows)} waves, {tiles:.0f} tiles per workgroup, "
          f"end-stamp spread {span_us:.1f} us")
    for i, nm in enumerate(names):
        v = rows[:, 2 + i]
        print(f"  {nm:18s} median {np.median(v):10.0f} cycles  ({np.median(v) / tiles:8.0f} per tile)   p10 {np.percentile(v, 10):10.0f}  p90 {np.percentile(v, 90):10.0f}")
    if rows.shape[1] >= 18:
        print("  per tile, by stage of the tile:  wait for DMA " + " ".join(f"{np.median(rows[:, 10 + q]) / tiles:6.0f}" for q in range(4))
              + "   stage barrier " + " ".join(f"{np.median(rows[:, 14 + q]) / tiles:6.0f}" for q in range(4)))


if __name__ == "__main__":
    main()
