#!/usr/bin/env python3
"""Per-layer roofline table of the ResNet-101-C4 backbone at the bench size (32 images of 800x1333, fp16): every distinct
conv shape through vk_conv2d / vk_conv1x1_dual on random data, alone on the GPU, with its two floors beside it --
MFMA (2*M*Cout*K flop at the dense f16 peak) and HBM (input + output (+ residual) + weights, each once, at the achievable
6.3 TB/s of MI355X_MICROARCH.md) -- and how often the layer occurs per forward.  GPU box only.
    python tools/backbone_layers.py [--batch 32] [--iters 10] [--json out.json]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from vltk_amd import _lib as L   # noqa: E402
import gpu_util as G             # noqa: E402

PEAK_TF, HBM_TBS = 2500.0, 6.3

# name: (H, W, cin, cout, k, stride, pad, residual, count per forward, cin2 of a fused shortcut or 0)
LAYERS = [
    ("res2.0.conv1", 200, 333, 64, 64, 1, 1, 0, False, 1, 0),
    ("res2.x.conv2", 200, 333, 64, 64, 3, 1, 1, False, 3, 0),
    ("res2.0.conv3+sc", 200, 333, 64, 256, 1, 1, 0, False, 1, 64),
    ("res2.x.conv1", 200, 333, 256, 64, 1, 1, 0, False, 2, 0),
    ("res2.x.conv3", 200, 333, 64, 256, 1, 1, 0, True, 2, 0),
    ("res3.0.conv1", 200, 333, 256, 128, 1, 2, 0, False, 1, 0),
    ("res3.0.shortcut", 200, 333, 256, 512, 1, 2, 0, False, 1, 0),
    ("res3.x.conv2", 100, 167, 128, 128, 3, 1, 1, False, 4, 0),
    ("res3.x.conv3", 100, 167, 128, 512, 1, 1, 0, True, 4, 0),
    ("res3.x.conv1", 100, 167, 512, 128, 1, 1, 0, False, 3, 0),
    ("res4.0.conv1", 100, 167, 512, 256, 1, 2, 0, False, 1, 0),
    ("res4.0.shortcut", 100, 167, 512, 1024, 1, 2, 0, False, 1, 0),
    ("res4.x.conv2", 50, 84, 256, 256, 3, 1, 1, False, 23, 0),
    ("res4.x.conv3", 50, 84, 256, 1024, 1, 1, 0, True, 23, 0),
    ("res4.x.conv1", 50, 84, 1024, 256, 1, 1, 0, False, 22, 0),
]


# ResNeXt-152 32x8d grouped conv2 layers (batch 16, as bench.py --arch x152): name: (N, H, W, C, groups, dil, count)
X152_GROUPED = [
    ("x152.res2.conv2", 16, 200, 333, 256, 32, 1, 3),
    ("x152.res3.conv2", 16, 100, 167, 512, 32, 1, 8),
    ("x152.res4.conv2", 16, 50, 84, 1024, 32, 1, 36),
    ("x152.res5.conv2", 4800, 14, 14, 2048, 32, 2, 3),
]


def grouped(a):
    g = np.random.Generator(np.random.PCG64(1))
    tot = 0.0
    for name, N, H, W, C_, groups, dil, count in X152_GROUPED:
        if a.names and name not in a.names:
            continue
        cg = C_ // groups
        w = (g.standard_normal((C_, cg, 3, 3)) * (2.0 / (cg * 9)) ** 0.5).astype(np.float32)
        wd, bd = G.pack_conv(w, None, np.zeros(C_, np.float32), L.VK_F16, groups)
        x = torch.randn((N, H, W, C_), device=G.DEV).half()
        y = torch.empty_like(x)

        def run():
            L.call("vk_conv2d", G.P(x), N, H, W, C_, G.P(wd), G.P(bd), None, G.P(y), C_, C_, 3, 3, 1, dil, dil, groups, 1,
                   L.VK_F16, L.VK_F16, G.stream())
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        M = N * H * W
        flop, byts = 2.0 * M * C_ * 9 * cg, 2.0 * M * C_ * 2
        tot += ms * count
        print(f"{name:18s} x{count:2d}  M={M:8d} C={C_:5d} cg={cg:3d}  {ms * 1e3:8.1f} us  {flop / ms / 1e9:7.1f} TFLOP/s  {byts / ms / 1e6:7.1f} GB/s(alg)"
              f"   floors: mfma {flop / (PEAK_TF * 1e9) * 1e3:7.1f} us, hbm {byts / (HBM_TBS * 1e9) * 1e3:7.1f} us", flush=True)
    print(f"grouped 3x3 layers of one x152 forward (batch 16, R = 300): {tot:.2f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--json", default=None)
    ap.add_argument("names", nargs="*")
    ap.add_argument("--x152", action="store_true", help="the grouped conv2 layers of ResNeXt-152 32x8d instead")
    a = ap.parse_args()
    if a.x152:
        return grouped(a)
    g = np.random.Generator(np.random.PCG64(0))
    N = a.batch
    rows, tot = [], {"ms": 0.0, "mfma_ms": 0.0, "hbm_ms": 0.0, "floor_ms": 0.0, "gflop": 0.0}
    for name, H, W, cin, cout, k, stride, pad, use_res, count, cin2 in LAYERS:
        if a.names and name not in a.names:
            continue
        w = (g.standard_normal((cout, cin, k, k)) * (2.0 / (cin * k * k)) ** 0.5).astype(np.float32)
        wd, bd = G.pack_conv(w, None, np.zeros(cout, np.float32), L.VK_F16)
        x = torch.randn((N, H, W, cin), device=G.DEV).half()
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        M = N * Ho * Wo
        y = torch.empty((N, Ho, Wo, cout), dtype=torch.float16, device=G.DEV)
        res = torch.randn((N, Ho, Wo, cout), device=G.DEV).half() if use_res else None
        if cin2:
            w2 = (g.standard_normal((cout, cin2, 1, 1)) * (2.0 / cin2) ** 0.5).astype(np.float32)
            wd2, bd2 = G.pack_conv(w2, None, np.zeros(cout, np.float32), L.VK_F16)
            rows_p = bd.shape[0]
            wcat = torch.cat([wd.view(rows_p, -1), wd2.view(rows_p, -1)], dim=1).contiguous()
            x2 = torch.randn((N, H, W, cin2), device=G.DEV).half()

            def run():
                L.call("vk_conv1x1_dual", G.P(x), cin, G.P(x2), cin2, M, G.P(wcat), G.P(bd + bd2), None, G.P(y), cout, 1, G.stream())
        else:
            def run():
                L.call("vk_conv2d", G.P(x), N, H, W, cin, G.P(wd), G.P(bd), G.P(res), G.P(y), cout, cout, k, k, stride, pad, 1, 1, 1,
                       L.VK_F16, L.VK_F16, G.stream())
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        K = (cin + cin2) * k * k
        flop = 2.0 * M * cout * K
        # strided 1x1 convs read only the pixels they use, but whole 128-byte lines of each
        in_rows = N * H * W if (stride == 1 or k > 1) else M
        byts = 2.0 * (in_rows * cin + M * cin2 + M * cout * (2 if use_res else 1) + cout * K)
        mfma_ms, hbm_ms = flop / (PEAK_TF * 1e9), byts / (HBM_TBS * 1e9)
        r = {"layer": name, "count": count, "M": M, "cout": cout, "K": K, "ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1),
             "alg_gbs": round(byts / ms / 1e6, 1), "mfma_floor_ms": round(mfma_ms, 4), "hbm_floor_ms": round(hbm_ms, 4),
             "bound": "hbm" if hbm_ms > mfma_ms else "mfma", "x_floor": round(ms / max(mfma_ms, hbm_ms), 2)}
        rows.append(r)
        tot["ms"] += ms * count
        tot["mfma_ms"] += mfma_ms * count
        tot["hbm_ms"] += hbm_ms * count
        tot["floor_ms"] += max(mfma_ms, hbm_ms) * count
        tot["gflop"] += flop * count / 1e9
        print(f"{name:18s} x{count:2d}  M={M:8d} cout={cout:5d} K={K:5d}  {ms * 1e3:8.1f} us  {r['tflops']:7.1f} TFLOP/s  {r['alg_gbs']:7.1f} GB/s(alg)"
              f"   floors: mfma {mfma_ms * 1e3:7.1f} us, hbm {hbm_ms * 1e3:7.1f} us -> {r['bound']:4s}  x{r['x_floor']:.2f} of floor", flush=True)
    print(f"sum over the forward's layers (stem + pool excluded): {tot['ms']:.2f} ms measured one layer at a time; floors: mfma {tot['mfma_ms']:.2f} ms, "
          f"hbm {tot['hbm_ms']:.2f} ms, per-layer max {tot['floor_ms']:.2f} ms; {tot['gflop'] / N:.1f} GFLOP/image")
    if a.json:
        json.dump({"batch": N, "layers": rows, "total": {k: round(v, 3) for k, v in tot.items()}}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
