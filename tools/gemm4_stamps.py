#!/usr/bin/env python3
"""In-kernel clock and cycles per K stage of the four-wave 1x1 GEMM kernel (diagnostic; needs a GPU).

Two seconds of back-to-back launches on random data, then one stamped launch (VK_GEMM4_STAMPS, conv_gemm4.hip STAMP build):
clock = s_memtime ticks / s_memrealtime ticks x 100 MHz around the K loop, median over workgroups; a stage is 64 MFMAs per wave
= 1024 matrix-pipe cycles.  usage: python tools/gemm4_stamps.py [shape]
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
    name = sys.argv[1] if len(sys.argv) > 1 else "head_conv1"
    out = "/tmp/gemm4_stamps.txt"
    if os.path.exists(out):
        os.remove(out)
    N, H, W, cin, cout, k, stride, pad, dil, use_res = SHAPES[name]
    g = np.random.Generator(np.random.PCG64(0))
    w = (g.standard_normal((cout, cin, k, k)) * (2.0 / (cin * k * k)) ** 0.5).astype(np.float32)
    wd, bd = G.pack_conv(w, None, np.zeros(cout, np.float32), L.VK_F16)
    x = torch.randn((N, H, W, cin), device=G.DEV).half()
    y = torch.empty((N, H, W, cout), dtype=torch.float16, device=G.DEV)
    res = torch.randn((N, H, W, cout), device=G.DEV).half() if use_res else None
    os.environ["VK_CONV_GEMM4"] = "2"

    def run():
        L.call("vk_conv2d", G.P(x), N, H, W, cin, G.P(wd), G.P(bd), G.P(res), G.P(y), cout, cout, k, k, stride, pad, dil, 1, 1,
               L.VK_F16, L.VK_F16, G.stream())
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
    os.environ["VK_GEMM4_STAMPS"] = out   # (VK_GEMM4_DBG picks a timing-only ablation build)
    run()
    torch.cuda.synchronize()
    del os.environ["VK_GEMM4_STAMPS"]
    rows = np.array([[int(v) for v in ln.split()] for ln in open(out) if not ln.startswith("#")], dtype=np.float64)
    cyc, rt = rows[:, 1], rows[:, 2]
    S = cin // 32
    clk = np.median(cyc / rt) * 100e6
    print(f"{name}: {len(rows)} workgroups, {S:.0f} stages each; in-kernel clock {clk / 1e9:.3f} GHz (p10 {np.percentile(cyc / rt, 10) / 10:.3f}, "
          f"p90 {np.percentile(cyc / rt, 90) / 10:.3f})")
    print(f"  K loop: median {np.median(cyc):.0f} cycles = {np.median(cyc) / S:.0f} per stage (1024 = matrix pipe always busy: "
          f"{1024 * S / np.median(cyc) * 100:.1f} %), p10 {np.percentile(cyc, 10) / S:.0f}, p90 {np.percentile(cyc, 90) / S:.0f}")
    if rows.shape[1] >= 9:
        tiles = rows[:, 8]
        print(f"  kernel span (first workgroup start to last workgroup end) {(rows[:, 4].max() - rows[:, 3].min()) / 100:.1f} us; workgroup start spread "
              f"{(rows[:, 3].max() - rows[:, 3].min()) / 100:.1f} us, end spread {(rows[:, 4].max() - rows[:, 4].min()) / 100:.1f} us")
        print(f"  per tile (us): zero + wait for stage 0 {np.median(rows[:, 5] / tiles) / 100:.2f}, K loop {np.median(rows[:, 6] / tiles) / 100:.2f}, "
              f"epilogue incl. its store acknowledgements {np.median(rows[:, 7] / tiles) / 100:.2f}  ({np.median(tiles):.0f} tiles per workgroup)")
        # where the end spread comes from: a workgroup's tiles all run on one XCD (blockIdx mod 8)
        end = (rows[:, 4] - rows[:, 3].min()) / 100
        wg = rows[:, 0].astype(int)
        per = [(x, end[wg % 8 == x].mean(), end[wg % 8 == x].min(), end[wg % 8 == x].max(), np.median(tiles[wg % 8 == x])) for x in range(8)]
        print("  end of a workgroup after the first start (us), by XCD: " + "  ".join(f"{x}: {m:.0f} ({lo:.0f}-{hi:.0f}, {t:.0f} tiles)" for x, m, lo, hi, t in per))
        print(f"  idle CU time behind the workgroups' ends: {100 * (end.max() - end.mean()) / end.max():.1f} % of the kernel")
    print(f"  MFMA peak at this clock: {clk * 1024 * 1024 / 1e12:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
