#!/usr/bin/env python3
"""Forward time of the LXMERT-style encoder on one MI355X (N3; BASELINE config 5's second half), bf16.
usage: python tools/lxmert_bench.py [batch ...]     default geometry: transformers' LxmertConfig (9/5/5 layers, hidden 768),
20 language tokens + 36 visual tokens per example."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vltk_amd.lxmert import LxmertEncoder, lxmert_config, make_lxmert_state_dict  # noqa: E402


def gflop(cfg, B, Lq, V):
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    def layer(T): return 2 * T * (4 * H * H + 2 * H * I)                 # q, k, v, output dense + FFN (2*MAC)
    def att(Tq, Tk): return 2 * 2 * Tq * Tk * H                          # QK^T and PV
    f = B * V * 2 * cfg["visual_feat_dim"] * H
    f += cfg["l_layers"] * (layer(B * Lq) + B * att(Lq, Lq)) + cfg["r_layers"] * (layer(B * V) + B * att(V, V))
    x = 2 * (B * Lq + B * V) * 2 * H * H + 2 * (B * Lq + B * V) * H * H * 1      # cross: q, kv, output dense (both directions)
    x += B * (att(Lq, V) + att(V, Lq)) + layer(B * Lq) + layer(B * V) + B * (att(Lq, Lq) + att(V, V))
    return (f + cfg["x_layers"] * x) / 1e9


def main():
    cfg = lxmert_config()
    m = LxmertEncoder(cfg, precision="bf16").load_state_dict(make_lxmert_state_dict(cfg, 1))
    gen = np.random.Generator(np.random.PCG64(0))
    for B in [int(a) for a in sys.argv[1:]] or [32, 256, 1024]:
        ids = torch.from_numpy(gen.integers(1, cfg["vocab_size"], (B, 20))).cuda()
        feats = torch.from_numpy(np.maximum(gen.standard_normal((B, 36, 2048)), 0).astype(np.float32)).cuda().bfloat16()
        pos = torch.from_numpy(gen.uniform(0, 1, (B, 36, 4)).astype(np.float32)).cuda().bfloat16()
        for _ in range(2):
            m(ids, feats, pos)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            m(ids, feats, pos)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        gf = gflop(cfg, B, 20, 36)
        replay = m.capture(ids, feats, pos)
        replay(ids, feats, pos)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            replay(ids, feats, pos)
        torch.cuda.synchronize()
        dg = (time.perf_counter() - t0) / n
        print(f"B={B:5d}: {dt * 1e3:8.2f} ms/forward  {B / dt:9.0f} examples/s  {gf / dt / 1e3:7.1f} TFLOP/s (algorithmic {gf:.0f} GFLOP)"
              f"   | HIP graph replay: {dg * 1e3:7.2f} ms  {B / dg:9.0f} examples/s")


if __name__ == "__main__":
    main()
