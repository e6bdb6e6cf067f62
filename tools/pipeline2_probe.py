#!/usr/bin/env python3
"""Probe (GPU box): does running consecutive batches on TWO streams (batch i+1's backbone / RPN / proposals / RoIPool beside batch
i's Res5 head) raise images/s?  Two model instances, one stream each, against one instance on one stream, interleaved."""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from vltk_amd import FRCNN, make_state_dict, synthetic_images, vg_c4_config  # noqa: E402


def main():
    B, steps = int(os.environ.get("B", "32")), int(os.environ.get("STEPS", "12"))
    cfg = vg_c4_config(post_nms_topk=300, detections=100, device="cuda:0")
    sd = make_state_dict(cfg, seed=1234)
    models = [FRCNN(cfg, precision="fp16", device="cuda:0").load_state_dict(sd).eval() for _ in range(2)]
    images = torch.from_numpy(synthetic_images(B, 800, 1333, seed=0xF2C)).cuda(0)
    shapes = torch.tensor([[800, 1333]] * B)
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]

    def run(two, n):
        infl = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            k = i % 2 if two else 0
            with torch.cuda.stream(streams[k]):
                infl.append(models[k].forward_async(images, shapes))
            if len(infl) > 1:
                infl.pop(0).wait()
        while infl:
            infl.pop(0).wait()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    for two in (False, True):
        run(two, 4)
    for rnd in range(3):
        a = run(False, steps)
        b = run(True, steps)
        print(f"round {rnd}: one stream {a * 1e3:.2f} ms/step ({B / a:.1f} img/s)   two streams {b * 1e3:.2f} ms/step ({B / b:.1f} img/s)   {100 * (a / b - 1):+.1f} %", flush=True)


if __name__ == "__main__":
    main()
