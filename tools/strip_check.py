#!/usr/bin/env python3
"""conv_strip_kernel vs the two-per-CU kernel on the same 1x1 layer: bit-identical outputs, timing, stamps (GPU box)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpu_util as G  # noqa: E402
from vltk_amd import _lib as L  # noqa: E402

N, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 14, 14
for cin, cout in ((512, 2048), (256, 1024)):
    g = np.random.Generator(np.random.PCG64(0))
    w = (g.standard_normal((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5).astype(np.float32)
    wd, bd = G.pack_conv(w, None, g.standard_normal(cout).astype(np.float32), L.VK_F16)
    x = torch.randn((N, H, W, cin), device=G.DEV).half()
    res = torch.randn((N, H, W, cout), device=G.DEV).half()
    outs = {}
    for strip in ("0", "1"):
        os.environ["VK_CONV_STRIP"] = strip
        os.environ["VK_CONV_DUO"] = "1"
        y = torch.full((N, H, W, cout), float("nan"), dtype=torch.float16, device=G.DEV)

        def run():
            L.call("vk_conv2d", G.P(x), N, H, W, cin, G.P(wd), G.P(bd), G.P(res), G.P(y), cout, cout, 1, 1, 1, 0, 1, 1, 1,
                   L.VK_F16, L.VK_F16, G.stream())
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        outs[strip] = (y.clone(), e0.elapsed_time(e1) / 10)
    same = torch.equal(outs["0"][0], outs["1"][0])
    M = N * H * W
    print(f"K={cin} N={cout} M={M}: duo {outs['0'][1] * 1e3:.1f} us, strip {outs['1'][1] * 1e3:.1f} us, bit-identical {same}, finite {bool(torch.isfinite(outs['1'][0].float()).all())}")
    if cin == 512:
        sf = "/tmp/strip_stamps.txt"
        if os.path.exists(sf):
            os.remove(sf)
        os.environ["VK_STRIP_STAMPS"] = sf
        run()
        torch.cuda.synchronize()
        del os.environ["VK_STRIP_STAMPS"]
        rows = np.array([[int(v) for v in ln.split()] for ln in open(sf) if not ln.startswith("#")], dtype=np.float64)
        print(f"  stamps: prologue median {np.median(rows[:, 1]) / 100:.2f} us, sweep median {np.median(rows[:, 2]) / 100:.2f} us per 128 x {cout} strip "
              f"({np.median(rows[:, 2]) / 100 / (cout / 256):.2f} us per 128x256 block), {np.median(rows[:, 3] / rows[:, 4]):.0f} core cycles per stage incl. epilogues, "
              f"clock {np.median(rows[:, 3] / (rows[:, 2] * 10)):.2f} GHz; wave 0: K loops {np.median((rows[:, 3] - rows[:, 5]) / rows[:, 4]):.0f} cycles per stage, "
              f"epilogue {np.median(rows[:, 5]) / (cout / 256):.0f} cycles per block")
