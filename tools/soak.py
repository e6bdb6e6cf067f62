#!/usr/bin/env python3
"""Soak (GPU box): N forwards of the bench batch back to back; every output block must equal the first one bit for bit
(dynamic tile tails, self-resetting counters, two-phase NMS: nothing may depend on which workgroup ran what)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vltk_amd import FRCNN, make_state_dict, synthetic_images, vg_c4_config  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    cfg = vg_c4_config(post_nms_topk=300, detections=100, device="cuda:0")
    m = FRCNN(cfg, precision="fp16", device="cuda:0").load_state_dict(make_state_dict(cfg, seed=1234)).eval()
    x = torch.from_numpy(synthetic_images(32, 800, 1333, seed=0xF2C)).cuda()
    shapes = torch.tensor([[800, 1333]] * 32)
    m(x, shapes)
    ref = {k: v.clone() for k, v in m.forward_padded().items()}
    t0 = time.time()
    bad = 0
    for i in range(n):
        m(x, shapes)
        cur = m.forward_padded()
        if not all(torch.equal(cur[k], ref[k]) for k in ref):
            bad += 1
        if i % 50 == 49:
            print(f"{i + 1} forwards, {bad} differing, {32 * (i + 1) / (time.time() - t0):.1f} images/s incl. the comparison", flush=True)
    print("soak:", "OK" if bad == 0 else f"{bad} of {n} forwards differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
