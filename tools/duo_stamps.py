#!/usr/bin/env python3
"""Phase timeline of the two-workgroups-per-CU 1x1 conv kernel (diagnostic; needs a GPU).

Runs one stamped launch (VK_DUO_STAMPS) of a tools/conv_bench.py shape after warm-up launches and prints, per
phase, the median duration, plus how the two workgroups that share a CU overlap.
usage: python tools/duo_stamps.py head_conv3 [out_file]

Needs the tools build of the library (make -C vltk_amd/csrc clean && make -C vltk_amd/csrc -j8 ABLATION=1): the shipped
build has no stamp / ablation instantiations and ignores the VK_*_STAMPS / VK_*_DBG variables.
"""
import collections
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
    name = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/duo_stamps.txt"
    if os.path.exists(out):
        os.remove(out)
    N, H, W, cin, cout, k, stride, pad, dil, use_res = SHAPES[name]
    g = np.random.Generator(np.random.PCG64(0))
    w = (g.standard_normal((cout, cin, k, k)) * (2.0 / (cin * k * k)) ** 0.5).astype(np.float32)
    wd, bd = G.pack_conv(w, None, np.zeros(cout, np.float32), L.VK_F16)
    x = torch.randn((N, H, W, cin), device=G.DEV).half()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((N, Ho, Wo, cout), dtype=torch.float16, device=G.DEV)
    res = torch.randn((N, Ho, Wo, cout), device=G.DEV).half() if use_res else None

    def run():
        L.call("vk_conv2d", G.P(x), N, H, W, cin, G.P(wd), G.P(bd), G.P(res), G.P(y), cout, cout, k, k, stride, pad, dil, 1, 1,
               L.VK_F16, L.VK_F16, G.stream())
    os.environ["VK_CONV_DUO"] = "1"
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    os.environ["VK_DUO_STAMPS"] = out
    run()
    torch.cuda.synchronize()
    del os.environ["VK_DUO_STAMPS"]
    rows = np.array([[int(v) for v in ln.split()] for ln in open(out) if not ln.startswith("#")], dtype=np.int64)
    ts = rows[:, 1:6].astype(np.float64) / 100.0          # us
    t00 = ts[:, 0].min()
    ts -= t00
    names = ["prologue", "mfma loop", "epilogue half 0", "epilogue half 1"]
    print(f"{name}: {len(rows)} workgroups, kernel span {ts[:, 4].max():.1f} us")
    for i, nm in enumerate(names):
        d = ts[:, i + 1] - ts[:, i]
        print(f"  {nm:16s} median {np.median(d):7.2f} us   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}")
    d = ts[:, 4] - ts[:, 0]
    print(f"  {'whole workgroup':16s} median {np.median(d):7.2f} us")
    S = int(rows[0, 12])
    loop, vm, bar = rows[:, 9].astype(np.float64), rows[:, 10].astype(np.float64), rows[:, 11].astype(np.float64)
    clk = np.median(loop / np.maximum((ts[:, 2] - ts[:, 1]) * 1e-6, 1e-12)) / 1e9
    print(f"  mfma loop (wave 0): {np.median(loop) / S:7.0f} core cycles per stage ({S} stages, 512 = MFMA-bound), clock {clk:.2f} GHz;"
          f" waiting for DMA {np.median(vm / loop):.1%}, at the barrier {np.median(bar / loop):.1%}")
    # co-residency: workgroups by (XCC, SE, SH, CU)
    cu = ((rows[:, 7] & 0xF) << 8) | ((rows[:, 6] >> 8) & 0xFF)
    by = collections.defaultdict(list)
    for b in range(len(rows)):
        by[int(cu[b])].append(b)
    print(f"  distinct CUs {len(by)}; workgroups per CU: min {min(map(len, by.values()))} max {max(map(len, by.values()))}")
    # per CU: fraction of one workgroup's epilogue time during which another workgroup of that CU is in its MFMA loop
    cov, tot, both_epi = 0.0, 0.0, 0.0
    for wl in by.values():
        for b in wl:
            e0, e1 = ts[b, 2], ts[b, 4]
            tot += e1 - e0
            for c in wl:
                if c == b:
                    continue
                cov += max(0.0, min(e1, ts[c, 2]) - max(e0, ts[c, 1]))
                both_epi += max(0.0, min(e1, ts[c, 4]) - max(e0, ts[c, 2]))
    print(f"  epilogue time under the co-resident workgroup's MFMA loop: {cov / tot:.1%}; under its epilogue: {both_epi / tot:.1%}")
    first = sorted(by.items(), key=lambda kv: min(kv[1]))[:4]
    for key, wl in first:
        print(f"  CU {key:#06x}: first workgroups {wl[:6]}  starts {[round(float(ts[b, 0]), 1) for b in wl[:6]]}")


if __name__ == "__main__":
    main()
