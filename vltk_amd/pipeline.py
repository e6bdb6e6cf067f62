"""Pipelined extraction loop: raw images in, `<split>.arrow` out, with the host work hidden behind the GPU.

The reference's loop (vltk/abc/extraction.py:142-220) is strictly serial per image: load + process on the CPU,
`forward` (batch 1), `.tolist()` into Python lists, and one in-memory Arrow table written at the very end.  At
hundreds of images per second on an MI355X everything around the forward has to leave the critical path:

    loader thread      next batch of raw HWC images (uint8 or float, BGR 0-255) is collected while the GPU works
    main thread        upload -> GPU pre-processing (vltk_amd.Preprocess: the legacy contract, scales_yx = raw/size)
                       -> FRCNN forward with `scales_yx` (boxes come back in raw-image coordinates, frcnn.py:1280-1283)
                       -> [world > 1: ONE all-gather of the flat output block, parallel.py]
                       -> asynchronous device-to-host copy of the flat block into a pinned ring slot (copy stream)
    writer thread      waits for the slot's event, rounds the boxes (adapters/frcnn.py:57) and streams the rows to the
                       Arrow file (extraction.ExtractionWriter); only rank 0 writes

Images shard across ranks by contiguous blocks (parallel.shard_indices); every rank runs the same number of steps (the
last batches are padded with a repeat of the rank's last image and dropped at the writer), so the per-step collective
always matches.  Rows are written in (step, rank, position) order; `img_to_row_map` makes the order irrelevant to readers
(abc/adapter.py:382-409).
"""
import queue
import threading

import numpy as np
import torch
import torch.distributed as dist

from .extraction import ExtractionWriter
from .parallel import OutputBlock, gather_outputs_async, output_spec, shard_indices

_STOP = object()


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class ExtractionPipeline:
    """model: vltk_amd.FRCNN (or any callable with the same `__call__`/`forward_padded()` surface);
    preprocess: vltk_amd.preprocess.Preprocess (or a callable (raws, ids) -> (ids, images, sizes, scales_yx))."""

    def __init__(self, model, preprocess, savefile, batch_size=32, visual_dim=2048, dataset=None, processor_args=None,
                 model_config=None, group=None, depth=2):
        self.model, self.preprocess = model, preprocess
        self.B, self.F = int(batch_size), int(visual_dim)
        self.D = int(model.roi_outputs.max_detections)
        self.group = group
        self.rank, self.world = _world(group)
        self.depth = max(2, int(depth))
        self.savefile = savefile
        self.writer = None
        if self.rank == 0:
            self.writer = ExtractionWriter(savefile, self.D, self.F, dataset=dataset, processor_args=processor_args,
                                           model_config=model_config)
        self.spec = output_spec(self.B, self.D, self.F)
        self.images_done = 0

    # ---- loader thread: batches of (ids, raws, n_valid) ----
    def _load(self, items, n_steps, q):
        try:
            it = iter(items)
            last = None
            for _ in range(n_steps):
                ids, raws = [], []
                for _ in range(self.B):
                    nxt = next(it, None)
                    if nxt is None:
                        break
                    ids.append(str(nxt[0]))
                    raws.append(np.ascontiguousarray(nxt[1]))
                    last = raws[-1]
                n_valid = len(ids)
                if last is None:
                    raise ValueError("a rank received no image at all: fewer images than ranks")
                while len(raws) < self.B:                      # keep the step's shape: repeat the last image, drop at the writer
                    raws.append(last)
                q.put((ids, raws, n_valid))
            q.put(_STOP)
        except BaseException as e:                              # surfaces in the main thread
            q.put(e)

    # ---- writer thread (rank 0): pinned slot -> Arrow rows ----
    def _write(self, q, free):
        try:
            while True:
                job = q.get()
                if job is _STOP:
                    return
                slot, host, event, ids_per_rank = job
                event.synchronize()
                nb = host.numel() // self.world
                for r in range(self.world):
                    ids = ids_per_rank[r]
                    if ids:
                        blk = OutputBlock(self.spec, flat=host[r * nb:(r + 1) * nb])
                        n = len(ids)
                        boxes = np.round(blk["boxes"][:n].numpy())          # adapters/frcnn.py:57 (half to even, as torch.round)
                        self.writer.write_batch(ids, blk["obj_ids"][:n].numpy().astype(np.float32),
                                                blk["attr_ids"][:n].numpy().astype(np.float32), boxes,
                                                blk["roi_features"][:n].numpy())
                free.put(slot)
        except BaseException as e:
            self._writer_error = e
            while True:                                         # keep draining so that the main thread never blocks
                job = q.get()
                if job is _STOP:
                    return
                free.put(job[0])

    def run(self, items, n_items=None):
        """items: this RANK's iterable of (imgid, raw HWC image) -- e.g. `all_items[slice(*shard_indices(n, rank, world))]`;
        n_items: total number of images over all ranks (default: len(items) when world == 1).
        Returns the Arrow path on rank 0, None elsewhere."""
        if n_items is None:
            if self.world > 1:
                raise ValueError("n_items (the global image count) is required with more than one rank")
            n_items = len(items)
        spans = [shard_indices(n_items, r, self.world) for r in range(self.world)]
        if n_items < self.world:
            # every rank sees the same numbers and raises here, before any thread or collective exists: a rank with an
            # empty shard could not pad its steps (no image to repeat) and its peers would wait for it in the all-gather
            raise ValueError(f"{n_items} image(s) for {self.world} ranks: every rank needs at least one image")
        n_steps = max((hi - lo + self.B - 1) // self.B for lo, hi in spans)
        dev = self.model.device
        load_q = queue.Queue(maxsize=2)
        loader = threading.Thread(target=self._load, args=(items, n_steps, load_q), daemon=True)
        loader.start()
        write_q, free = queue.Queue(), queue.Queue()
        self._writer_error = None
        wthread = None
        host, gbuf = [], []
        nbytes = OutputBlock(self.spec, device="cpu").flat.numel()
        if self.rank == 0:
            for s in range(self.depth):
                h = torch.empty(self.world * nbytes, dtype=torch.uint8)
                host.append(h.pin_memory() if dev.type == "cuda" else h)
                free.put(s)
            wthread = threading.Thread(target=self._write, args=(write_q, free), daemon=True)
            wthread.start()
        if self.world > 1:
            gbuf = [torch.empty(self.world * nbytes, dtype=torch.uint8, device=dev) for _ in range(self.depth)]
        on_gpu = dev.type == "cuda"
        copy_stream = torch.cuda.Stream(device=dev) if on_gpu else None
        upload_stream = torch.cuda.Stream(device=dev) if on_gpu else None

        def finish(p, ids, n_valid, step):
            blk = p.wait_raw()
            slot = step % self.depth
            if self.rank == 0:
                # back-pressure: at most `depth` batches in flight; slot s owns pinned buffer s AND gather buffer s, so
                # a gather never lands in a buffer whose device-to-host copy is still running
                slot = free.get()
                if self._writer_error is not None:
                    raise self._writer_error
            handle = gather_outputs_async(blk, self.group, out=gbuf[slot] if gbuf else None)
            flat, _ = handle.wait_flat()
            if self.rank == 0:
                if copy_stream is not None:
                    copy_stream.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(copy_stream):
                        host[slot].copy_(flat, non_blocking=True)
                        flat.record_stream(copy_stream)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                else:
                    host[slot].copy_(flat)
                    ev = _DoneEvent()
                # which ids each rank's block holds in this step (all ranks derive it from the spans alone)
                ids_per_rank = []
                for r, (lo, hi) in enumerate(spans):
                    a, b = min(lo + step * self.B, hi), min(lo + (step + 1) * self.B, hi)
                    ids_per_rank.append(ids[:n_valid] if r == self.rank else self._ids_of(r, a, b))
                write_q.put((slot, host[slot], ev, ids_per_rank))
            self.images_done += n_valid

        try:
            prev = None
            for step in range(n_steps):
                job = load_q.get()
                if isinstance(job, BaseException):
                    raise job
                ids, raws, n_valid = job
                # uint8 stays uint8 over PCIe; the copies run on their own stream (a pageable host-to-device copy on the
                # compute stream would make the host wait for the forward that is still running there)
                if on_gpu:
                    with torch.cuda.stream(upload_stream):
                        dev_raws = [torch.from_numpy(r).to(dev) for r in raws]
                    torch.cuda.current_stream(dev).wait_stream(upload_stream)
                    for t in dev_raws:
                        t.record_stream(torch.cuda.current_stream(dev))
                else:
                    dev_raws = [torch.from_numpy(r) for r in raws]
                _, images, sizes, scales_yx = self.preprocess(dev_raws, list(range(self.B)))
                if hasattr(self.model, "forward_async"):      # enqueue this batch behind the previous one, then finish that one
                    p = self.model.forward_async(images, sizes, scales_yx=scales_yx)
                else:
                    self.model(images, sizes, scales_yx=scales_yx)
                    p = _Finished(self.model.forward_padded())
                if prev is not None:
                    finish(*prev)
                prev = (p, ids, n_valid, step)
            if prev is not None:
                finish(*prev)
        finally:
            if wthread is not None:
                write_q.put(_STOP)
                wthread.join()
        if self._writer_error is not None:
            raise self._writer_error
        return self.writer.close() if self.rank == 0 else None

    # ids held by other ranks: rank 0 needs them to label the gathered rows.  Default: the caller registered the
    # global id list with set_global_ids(); a one-rank run never gets here.
    def set_global_ids(self, ids):
        self._global_ids = [str(i) for i in ids]
        return self

    def _ids_of(self, rank, a, b):
        g = getattr(self, "_global_ids", None)
        if g is None:
            raise ValueError("with more than one rank, rank 0 needs the global id list: call set_global_ids(all_ids)")
        return g[a:b]


class _DoneEvent:
    def synchronize(self):
        pass


class _Finished:
    """A forward that has already completed (models without forward_async)."""

    def __init__(self, block):
        self.block = block

    def wait_raw(self):
        return self.block


def extract_images(model, preprocess, items, savedir, split="train", dataset=None, batch_size=32, processor_args=None,
                   group=None, global_ids=None, n_items=None):
    """Convenience wrapper: run the pipeline over this rank's `items` and return `<savedir>/<split>.arrow` (rank 0)."""
    import os
    cfgd = model.config.to_dict() if hasattr(getattr(model, "config", None), "to_dict") else None
    pipe = ExtractionPipeline(model, preprocess, os.path.join(savedir, f"{split}.arrow"), batch_size=batch_size,
                              visual_dim=getattr(model, "visual_dim", 2048), dataset=dataset, processor_args=processor_args,
                              model_config=cfgd, group=group)
    if global_ids is not None:
        pipe.set_global_ids(global_ids)
    return pipe.run(items, n_items=n_items)
