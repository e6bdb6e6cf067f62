#!/usr/bin/env python3
"""Where a tile of the fused res2 bottleneck kernel spends its cycles (diagnostic; needs a GPU and the tools build of the library:
make -C vltk_amd/csrc clean && make -C vltk_amd/csrc -j8 ABLATION=1).  One stamped launch per block kind at bench size after a
second of back-to-back launches; per wave: core cycles of phases A / B / C summed over its tiles.
usage: python tools/bneck_stamps.py [batch=32]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G                                   # noqa: E402
from vltk_amd import _lib as L                        # noqa: E402
import test_gpu_bneck_fused as T                       # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = 200, 333
DBGS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]      # rows form, identity block: VK_BNECK_DBG ablations
for proj, dbg in [(False, d) for d in DBGS] + [(True, 0)]:
    os.environ["VK_BNECK_DBG"] = str(dbg)
    cin = 64 if proj else 256
    g = np.random.Generator(np.random.PCG64(1))
    xd = torch.from_numpy(g.standard_normal((B, H, W, cin)).astype(np.float16)).to(G.DEV).relu_()
    p = T.make_block(3, cin, proj)
    w = T.packed(p, proj)
    y = torch.empty((B, H, W, 256), dtype=torch.float16, device=G.DEV)

    def run():
        L.call("vk_bottleneck64", G.P(xd), B, H, W, cin, int(proj), G.P(w[0]), G.P(w[1]), G.P(w[2]), G.P(w[3]), G.P(w[4]), G.P(w[5]), G.P(y), G.stream())
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    path = "/tmp/bneck_stamps.txt"
    if os.path.exists(path):
        os.remove(path)
    os.environ["VK_BNECK_STAMPS"] = path
    run()
    torch.cuda.synchronize()
    del os.environ["VK_BNECK_STAMPS"]
    rows = np.array([[float(v) for v in ln.split()] for ln in open(path) if not ln.startswith("#")])
    a, b, c, n, cyc, ticks = (rows[:, i] for i in range(2, 8))
    clock = np.median(cyc / ticks) * 0.1
    print(f"proj={proj} dbg={dbg}: clock {clock:.2f} GHz; per tile (median over waves, core cycles): phase A {np.median(a / n):.0f}  B {np.median(b / n):.0f}  "
          f"C {np.median(c / n):.0f}  sum {np.median((a + b + c) / n):.0f}; tiles per wave {np.median(n):.0f}; kernel {np.median(ticks) / 100:.1f} us")
