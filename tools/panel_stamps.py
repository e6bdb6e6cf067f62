#!/usr/bin/env python3
"""In-kernel clock, cycles per step and share of the K loop of the 3x3 panel kernel (diagnostic; needs a GPU).

Two seconds of back-to-back launches on random data, then one stamped launch (VK_PANEL_STAMPS, conv3x3_panel.hip DBG 4).
A step (one tap of one 32-channel stage) is 32 MFMAs per wave on two waves per SIMD = 1024 matrix-pipe cycles.
usage: python tools/panel_stamps.py [shape]
Needs the tools build of the library (make -C vltk_amd/csrc clean && make -C vltk_amd/csrc -j8 ABLATION=1): the shipped
build has no stamp / ablation instantiations and ignores the VK_*_STAMPS / VK_*_DBG variables.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpu_util as G  # noqa: E402
from vltk_amd import _lib as L  # noqa: E402
from conv_bench import SHAPES  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "head_conv2"
    out = "/tmp/panel_stamps.txt"
    if os.path.exists(out):
        os.remove(out)
    N, H, W, cin, cout, k, stride, pad, dil, use_res = SHAPES[name]
    g = np.random.Generator(np.random.PCG64(0))
    w = (g.standard_normal((cout, cin, k, k)) * (2.0 / (cin * k * k)) ** 0.5).astype(np.float32)
    wd, bd = G.pack_conv(w, None, np.zeros(cout, np.float32), L.VK_F16)
    x = torch.randn((N, H, W, cin), device=G.DEV).half()
    y = torch.empty((N, H, W, cout), dtype=torch.float16, device=G.DEV)

    def run():
        L.call("vk_conv2d", G.P(x), N, H, W, cin, G.P(wd), G.P(bd), None, G.P(y), cout, cout, k, k, stride, pad, dil, 1, 1,
               L.VK_F16, L.VK_F16, G.stream())
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    os.environ["VK_PANEL_STAMPS"] = out
    run()
    torch.cuda.synchronize()
    del os.environ["VK_PANEL_STAMPS"]
    rows = np.array([[int(v) for v in ln.split()] for ln in open(out) if not ln.startswith("#")], dtype=np.float64)
    rows = rows[rows[:, 3] > 0]          # spare workgroups of a dynamic tail leave no stamp
    cyc, kt, wt, pre = rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4]
    steps = 9 * cin // 32
    clk = np.median(cyc / kt) * 100e6
    print(f"{name}: {len(rows)} workgroups, {steps} steps each; in-kernel clock {clk / 1e9:.3f} GHz")
    print(f"  K loop: median {np.median(cyc):.0f} cycles = {np.median(cyc) / steps:.0f} per step (1024 per 32 MFMAs per wave = matrix pipe always busy; 36 per wave and step with 288-pixel tiles: "
          f"{1024 * steps / np.median(cyc) * 100:.1f} %)")
    print(f"  per workgroup (us): before the K loop {np.median(pre) / 100:.2f}, K loop {np.median(kt) / 100:.2f}, after it "
          f"{np.median(wt - kt - pre) / 100:.2f} (epilogue incl. store acknowledgements): K loop = {np.median(kt / wt) * 100:.1f} % of the workgroup")
    if rows.shape[1] >= 7:      # when each XCD (blockIdx mod 8) runs dry: its share of the grid is fixed by the dispatcher
        wg, t0, t1 = rows[:, 0].astype(int), rows[:, 5], rows[:, 6]
        span = (t1.max() - t0.min()) / 100
        ends = [(t1[wg % 8 == x].max() - t0.min()) / 100 for x in range(8)]
        print(f"  kernel span {span:.1f} us; last workgroup end by XCD (us): " + " ".join(f"{e:.0f}" for e in ends) +
              f"; CU time idle behind them: {100 * (1 - np.mean(ends) / max(ends)):.1f} % of the kernel")


if __name__ == "__main__":
    main()
