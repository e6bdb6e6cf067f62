"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/frcnn_oracle.py for the rules).

CPU restatement of the reference's legacy image pre-processing, the step immediately before the hot path
(SURVEY.md §8f N2): `ResizeShortestEdge` + `Preprocess` of /root/reference/vltk/legacy/processing.py:29-150.
Pinned by tests/golden/preprocess.npz, produced by the reference's own classes (tools/gen_golden.py --preprocess).
"""
import numpy as np
import torch
import torch.nn.functional as F


def resized_hw(h, w, min_size, max_size):
    """ResizeShortestEdge.__call__ size rule (processing.py:41-60): Python float arithmetic, int(x + 0.5)."""
    size = min_size
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh = newh * scale
        neww = neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def preprocess(raws, min_size, max_size, pixel_mean, pixel_std, pad_value=0.0):
    """raws: list of float HWC (BGR, 0-255) tensors -> (images [N,3,Hmax,Wmax], sizes [N,2], scales_yx [N,2])."""
    mean = torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1)
    imgs = []
    for img in raws:
        img = torch.as_tensor(img, dtype=torch.float32)
        h, w = img.shape[:2]
        nh, nw = resized_hw(h, w, min_size, max_size)
        x = F.interpolate(img.permute(2, 0, 1).unsqueeze(0), (nh, nw), mode="bilinear", align_corners=False).squeeze(0)
        imgs.append((x - mean) / std)                       # normalizer processing.py:96
    hmax, wmax = max(i.shape[-2] for i in imgs), max(i.shape[-1] for i in imgs)
    sizes = torch.tensor([list(i.shape[-2:]) for i in imgs])
    out = torch.stack([F.pad(i, [0, wmax - i.shape[-1], 0, hmax - i.shape[-2]], value=pad_value) for i in imgs])
    raw_sizes = torch.tensor([list(torch.as_tensor(r).shape[:2]) for r in raws])
    return out, sizes, torch.true_divide(raw_sizes, sizes)  # processing.py:98-110, :149
