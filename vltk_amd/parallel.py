"""Image sharding across the GPUs of one node and re-collection of the feature arrays.

The reference has no distributed code at all: extraction is a single-process
loop over files (vltk/abc/extraction.py:142-220).  Images are independent (no cross-image op
in FRCNN.inference, frcnn.py:1942-2004), so the path shards by image with NO data-path
collective; the only exchange is one all-gather (RCCL over xGMI with backend "nccl", gloo on
CPU in the tests) of the fixed-size per-rank output blocks, so that every rank -- in particular
the rank that writes the Arrow table -- sees the whole step.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist

OUTPUT_KEYS = ("obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "preds_per_image", "roi_features")


def shard_indices(n_items, rank, world_size):
    """Contiguous block partition: rank r takes items [lo, hi).  Sizes differ by at most 1."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_outputs(padded, group=None):
    """All-gather the fixed-capacity output block of one step.

    padded: dict of [B, D, ...] tensors (FRCNN.forward_padded()) with identical shapes on every rank.
    Returns a dict of [world*B, D, ...] tensors ordered by rank (== image order under shard_indices
    when every rank holds an equal block)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return OrderedDict((k, padded[k]) for k in OUTPUT_KEYS)
    world = dist.get_world_size(group)
    out = OrderedDict()
    for k in OUTPUT_KEYS:
        t = padded[k].contiguous()
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        out[k] = torch.cat(parts, dim=0)
    return out
