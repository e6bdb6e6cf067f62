"""Image sharding across the GPUs of one node and re-collection of the feature arrays.

The reference has no distributed code at all: extraction is a single-process
loop over files (vltk/abc/extraction.py:142-220).  Images are independent (no cross-image op
in FRCNN.inference, frcnn.py:1942-2004), so the path shards by image with NO data-path
collective; the only exchange is ONE all-gather per step (RCCL over xGMI with backend "nccl",
gloo on CPU in the tests) of the fixed-size per-rank output block, so that every rank -- in
particular the rank that writes the Arrow table -- sees the whole step.

The seven output arrays of a step live in one flat byte block (`OutputBlock`: 256-byte aligned
segments, the tensors are views), so the exchange is a single `all_gather_into_tensor` of that
block -- no packing copies, one collective instead of seven -- and it can run on RCCL's own
stream under the next step's forward (`gather_outputs_async`).
"""
import os
from collections import OrderedDict

import torch
import torch.distributed as dist

OUTPUT_KEYS = ("obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "preds_per_image", "roi_features")
_ALIGN = 256


def shard_indices(n_items, rank, world_size):
    """Contiguous block partition: rank r takes items [lo, hi).  Sizes differ by at most 1."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def output_spec(N, D, F):
    """(key, shape, dtype) of the fixed-capacity output arrays of one step of N images (FRCNN.forward_padded())."""
    return (("obj_ids", (N, D), torch.int64), ("obj_probs", (N, D), torch.float32),
            ("attr_ids", (N, D), torch.int64), ("attr_probs", (N, D), torch.float32),
            ("boxes", (N, D, 4), torch.float32), ("preds_per_image", (N,), torch.int64),
            ("roi_features", (N, D, F), torch.float32))


def _layout(spec):
    offs, off = [], 0
    for _, shape, dtype in spec:
        n = torch.empty((), dtype=dtype).element_size()
        for s in shape:
            n *= s
        offs.append((off, n))
        off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
    return offs, max(off, _ALIGN)


class OutputBlock(OrderedDict):
    """key -> tensor views into one flat uint8 buffer (`.flat`), in OUTPUT_KEYS order."""

    def __init__(self, spec, device=None, flat=None):
        super().__init__()
        self.spec = tuple(spec)
        offs, total = _layout(self.spec)
        self.flat = flat if flat is not None else torch.empty(total, dtype=torch.uint8, device=device)
        assert self.flat.numel() == total and self.flat.dtype == torch.uint8 and self.flat.is_contiguous()
        for (key, shape, dtype), (off, n) in zip(self.spec, offs):
            self[key] = self.flat[off:off + n].view(dtype).view(shape)

    @classmethod
    def from_tensors(cls, padded):
        """Pack a plain dict of output arrays (copies; for callers that did not get an OutputBlock)."""
        spec = tuple((k, tuple(padded[k].shape), padded[k].dtype) for k in OUTPUT_KEYS)
        blk = cls(spec, device=padded[OUTPUT_KEYS[0]].device)
        for k in OUTPUT_KEYS:
            blk[k].copy_(padded[k])
        return blk


class _Gathered:
    """Handle of an all-gather in flight: wait() -> OrderedDict of [world*B, D, ...] tensors ordered by rank."""

    def __init__(self, block, flat_all, work, world):
        self.block, self.flat_all, self.work, self.world = block, flat_all, work, world

    def wait_flat(self):
        """(flat uint8 buffer, world): the gathered blocks back to back, rank by rank (the rank's own block when no
        collective ran).  Each rank's slice parses with OutputBlock(spec, flat=slice)."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.flat_all is None:
            blk = self.block if isinstance(self.block, OutputBlock) else OutputBlock.from_tensors(self.block)
            return blk.flat, 1
        return self.flat_all, self.world

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.flat_all is None:
            return OrderedDict((k, self.block[k]) for k in OUTPUT_KEYS)
        per_rank = self.flat_all.view(self.world, -1)
        offs, _ = _layout(self.block.spec)
        out = OrderedDict()
        for (key, shape, dtype), (off, n) in zip(self.block.spec, offs):
            # [world, n bytes] strided view -> [world * B, ...] (one small device copy per key)
            t = per_rank[:, off:off + n].contiguous().view(dtype)
            out[key] = t.view((self.world * shape[0],) + tuple(shape[1:]))
        return out


def gather_outputs_async(padded, group=None, out=None):
    """Start the all-gather of one step's output block; returns a handle whose wait() gives the gathered arrays.

    padded: an OutputBlock (FRCNN.forward_padded(): sent as is) or a dict of [B, D, ...] tensors with identical
    shapes on every rank (packed first).  out: optional preallocated uint8 destination of world * block bytes
    (a steady-state loop passes its own rotating buffers so that no step allocates)."""
    # VLTK_AMD_FORCE_COLLECTIVE=1: run the collective even in a one-rank group (exercises the RCCL path on a one-GPU box)
    force = os.environ.get("VLTK_AMD_FORCE_COLLECTIVE") == "1"
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return _Gathered(padded, None, None, 1)
    world = dist.get_world_size(group)
    block = padded if isinstance(padded, OutputBlock) else OutputBlock.from_tensors(padded)
    nbytes = world * block.flat.numel()
    flat_all = out if out is not None and out.numel() == nbytes else torch.empty(nbytes, dtype=torch.uint8, device=block.flat.device)
    work = dist.all_gather_into_tensor(flat_all, block.flat, group=group, async_op=True)
    return _Gathered(block, flat_all, work, world)


def gather_outputs(padded, group=None):
    """All-gather the fixed-capacity output block of one step (blocking form).

    Returns a dict of [world*B, D, ...] tensors ordered by rank (== image order under shard_indices when every
    rank holds an equal block)."""
    return gather_outputs_async(padded, group).wait()
