#!/usr/bin/env python3
"""Where a step's wall time goes outside the kernels: wall ms per forward vs the HIP-event span first-to-last kernel,
with and without the per-launch kernel timers (diagnostic; GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vltk_amd import FRCNN, make_state_dict, synthetic_images, vg_c4_config  # noqa: E402

cfg = vg_c4_config(post_nms_topk=300, detections=100, device="cuda:0")
model = FRCNN(cfg, precision="fp16").load_state_dict(make_state_dict(cfg, seed=1234)).eval()
x = torch.from_numpy(synthetic_images(32, 800, 1333, seed=0xF2C)).cuda()
sh = torch.tensor([[800, 1333]] * 32)
for kt in (False, True, False, True):
    model.enable_kernel_timing(kt)
    model.enable_stage_timing(True)
    for _ in range(2):
        model(x, sh, padding="max_detections", return_tensors="pt", location="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        model(x, sh, padding="max_detections", return_tensors="pt", location="cuda")
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    print(f"kernel timers {'on ' if kt else 'off'}: wall {wall:.2f} ms/step, event span of the last forward {model.stage_timing_ms()['total']:.2f} ms, "
          f"outside {wall - model.stage_timing_ms()['total']:.2f} ms")
