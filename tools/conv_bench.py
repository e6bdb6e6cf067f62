#!/usr/bin/env python3
"""Micro-benchmark of vk_conv2d on the FRCNN layer shapes (GPU box).  Random data (never zeros:
zero operands clock higher, cdna guide rule 25).  Usage: python tools/conv_bench.py [shape-name ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from vltk_amd import _lib as L   # noqa: E402
import gpu_util as G             # noqa: E402

# name: (N, H, W, cin, cout, k, stride, pad, dil, residual)
SHAPES = {
    "head_conv2": (1024, 14, 14, 512, 512, 3, 1, 2, 2, False),
    "head_conv2_full": (9600, 14, 14, 512, 512, 3, 1, 2, 2, False),
    "head_conv3": (1024, 14, 14, 512, 2048, 1, 1, 0, 1, True),
    "head_conv3_full": (9600, 14, 14, 512, 2048, 1, 1, 0, 1, True),
    "head_conv1": (1024, 14, 14, 2048, 512, 1, 1, 0, 1, False),
    "head_conv1_full": (9600, 14, 14, 2048, 512, 1, 1, 0, 1, False),
    "head_short": (1024, 14, 14, 1024, 2048, 1, 1, 0, 1, False),
    "head_short_full": (9600, 14, 14, 1024, 2048, 1, 1, 0, 1, False),
    "res4_conv3": (32, 50, 84, 256, 1024, 1, 1, 0, 1, True),
    "res4_conv2": (32, 50, 84, 256, 256, 3, 1, 1, 1, False),
    "res4_conv1": (32, 50, 84, 1024, 256, 1, 1, 0, 1, False),
    "res2_conv3": (32, 200, 333, 64, 256, 1, 1, 0, 1, True),
    "res3_conv3": (32, 100, 167, 128, 512, 1, 1, 0, 1, True),
    "rpn_conv": (32, 50, 84, 1024, 512, 3, 1, 1, 1, False),
}


def main():
    variants = [v for v in os.environ.get("VARIANTS", "").split(",") if v]   # e.g. VARIANTS=0,4,8: VK_CONV256_DBG values, interleaved rounds
    # ENVVARIANTS="VK_GEMM4_MINK=512+VK_CONV_WS=0,VK_GEMM4_MINK=1024": arbitrary environment settings per variant, interleaved
    envvariants = [v for v in os.environ.get("ENVVARIANTS", "").split(",") if v]
    if envvariants:
        variants = ["e:" + v for v in envvariants]
    names = sys.argv[1:] or list(SHAPES)
    iters = int(os.environ.get("ITERS", "10"))
    g = np.random.Generator(np.random.PCG64(0))
    torch.manual_seed(0)
    for name in names:
        N, H, W, cin, cout, k, stride, pad, dil, use_res = SHAPES[name]
        w = (g.standard_normal((cout, cin, k, k)) * (2.0 / (cin * k * k)) ** 0.5).astype(np.float32)
        wd, bd = G.pack_conv(w, None, np.zeros(cout, np.float32), L.VK_F16)
        x = (torch.randn((N, H, W, cin), device=G.DEV) * 1.0).half()
        Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
        Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
        y = torch.empty((N, Ho, Wo, cout), dtype=torch.float16, device=G.DEV)
        res = torch.randn((N, Ho, Wo, cout), device=G.DEV).half() if use_res else None

        def run():
            L.call("vk_conv2d", G.P(x), N, H, W, cin, G.P(wd), G.P(bd), G.P(res), G.P(y), cout, cout, k, k, stride, pad,
                   dil, 1, 1, L.VK_F16, L.VK_F16, G.stream())
        M = N * Ho * Wo
        if os.environ.get("CHECKSUM") == "1":       # two builds / two switch settings must give the same bits: compare these lines
            run()
            torch.cuda.synchronize()
            v = y.view(torch.int16).to(torch.int64)
            print(f"{name:12s} checksum {int(v.sum())} {int((v * (torch.arange(v.numel(), device=v.device).view(v.shape) % 65521)).sum())}")
            continue
        if variants:      # interleaved A/B rounds in ONE process on ONE device (cdna guide rule 24)
            res_ms = {v: [] for v in variants}
            for rnd in range(int(os.environ.get("ROUNDS", "5"))):
                for vkey in variants:
                    v = vkey
                    if v.startswith("e:"):
                        for ev in envvariants:                         # clear what the other variants set
                            for kv in ev.split("+"):
                                os.environ.pop(kv.split("=")[0], None)
                        for kv in v[2:].split("+"):
                            os.environ[kv.split("=")[0]] = kv.split("=")[1]
                        v = "0"
                    # a variant is "<dbg>" or "<kernel><dbg>", e.g. "0", "a0", "p0", "d0"
                    # "p..." = LDS-panel 3x3 kernel enabled, otherwise the im2col ring kernels
                    if not vkey.startswith("e:"):
                        os.environ["VK_CONV3X3_PANEL"] = "1" if v[0] == "p" else "0"
                        os.environ["VK_CONV_DUO"] = "1" if v[0] == "d" else "0"
                        os.environ["VK_CONV256_DBG"] = v.lstrip("apd") or "0"
                    run()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    res_ms[vkey].append(e0.elapsed_time(e1) / iters)
            fl = 2.0 * M * cout * cin * k * k
            print(name, " ".join(f"v{v}: med {np.median(t) * 1e3:.1f} us min {min(t) * 1e3:.1f} us "
                                 f"({fl / np.median(t) / 1e9:.0f} TF)" for v, t in res_ms.items()), flush=True)
            continue
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 2.0 * M * cout * cin * k * k
        by = 2.0 * (M * cin + M * cout * (2 if use_res else 1) + cout * cin * k * k)
        print(f"{name:12s} M={M:8d} cout={cout:5d} K={cin * k * k:5d} {ms * 1e3:9.1f} us {fl / ms / 1e9:8.1f} TFLOP/s "
              f"{by / ms / 1e6:8.1f} GB/s(alg)", flush=True)


if __name__ == "__main__":
    main()
