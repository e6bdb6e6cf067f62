#!/usr/bin/env python3
"""BASELINE configs[2] as ONE command: an M-image GQA-shaped extract, image-sharded over N ranks of one node.

    python tools/extract_bench.py --gpus N --images M [--batch 32] [--detections 36] [--proposals 300]

Every rank: raw uint8 images in host memory -> PCIe -> GPU resize / normalise / pad to 800x1333 (vltk_amd.Preprocess, legacy
contract) -> FRCNN fp16 (R proposals, D detections) -> ONE all-gather of the flat output block per step (RCCL; the reference's
feature arrays, vltk/abc/extraction.py:142-246) -> rank 0: device-to-host -> Arrow IPC file, through
vltk_amd.pipeline.ExtractionPipeline (loader and writer threads).  Images shard by contiguous blocks (parallel.shard_indices).
Called plainly with --gpus N > 1 the process starts the ranks itself through bench.py's launcher (a parent that never touches
HIP -> torch.distributed.run on 127.0.0.1).  Rank 0 prints ONE JSON line: whole-job images/s with everything in the timed
region (upload, pre-processing, forward, exchange, read-back, Arrow write; JPEG decode excluded and measured apart), the ranks
seen by the collective, and what the host side costs per core -- SURVEY.md 8e's risk "host feeding at 8 x 470 images/s" as a number.

Validation without an 8-GPU node: VLTK_AMD_BENCH_ONE_GPU=1 puts every rank on device 0 and exchanges over gloo (RCCL needs one
GPU per rank); --selftest runs the same control flow on the CPU with a stand-in model (tests/test_extract_bench.py).
"""
import argparse
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def host_feed_costs(raw, n=24):
    """What feeding one image costs on ONE host core: JPEG decode (PIL) of a 480x640 image, and the uint8 -> device upload."""
    import numpy as np
    out = {}
    try:
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(raw[..., ::-1]).save(buf, format="JPEG", quality=90)
        data = buf.getvalue()
        t0 = time.perf_counter()
        for _ in range(n):
            np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        dt = (time.perf_counter() - t0) / n
        out["jpeg_decode_ms_per_image_one_core"] = round(dt * 1e3, 3)
        out["jpeg_decode_images_per_s_per_core"] = round(1.0 / dt, 1)
        out["jpeg_bytes"] = len(data)
    except Exception as e:          # PIL missing: say so, measure the rest
        out["jpeg_decode"] = f"not measured ({e!r})"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--images", type=int, default=2048, help="images of the whole job (BASELINE configs[2]: 50000)")
    ap.add_argument("--batch", type=int, default=32, help="images per rank per step")
    ap.add_argument("--detections", type=int, default=36)
    ap.add_argument("--proposals", type=int, default=300)
    ap.add_argument("--raw-hw", default="480x640", help="raw image size (GQA / VG-like)")
    ap.add_argument("--selftest", action="store_true", help="CPU: stand-in model and pre-processing, gloo (control flow only)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import bench
        sys.exit(bench.launch_ranks(a, script=__file__))

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")          # bench.py: RCCL's streams must not share the forward's queues
    json_fd = os.dup(1)                                        # RCCL's banner and everything else: stderr
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    one_gpu = os.environ.get("VLTK_AMD_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.selftest or one_gpu:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from vltk_amd.parallel import shard_indices
    from vltk_amd.pipeline import ExtractionPipeline
    rh, rw = (int(v) for v in a.raw_hw.split("x"))
    g = np.random.Generator(np.random.PCG64(0xF2C + rank))
    pool = [g.integers(0, 256, (rh, rw, 3), dtype=np.uint8) for _ in range(64)]
    if a.selftest:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        ns = {}
        import test_pipeline
        exec(test_pipeline._FAKE, ns)
        model, pre, F = ns["FakeModel"](), ns["fake_preprocess"], 8
        pool = [np.full((4, 6, 3), i, dtype=np.uint8) for i in range(64)]
    else:
        torch.cuda.set_device(local_rank)
        from vltk_amd import FRCNN, make_state_dict, vg_c4_config
        from vltk_amd.preprocess import Preprocess
        cfg = vg_c4_config(post_nms_topk=a.proposals, detections=a.detections, device=f"cuda:{local_rank}")
        model = FRCNN(cfg, precision="fp16", device=f"cuda:{local_rank}").load_state_dict(make_state_dict(cfg, seed=1234)).eval()
        pre, F = Preprocess(cfg, device=model.device), 2048
    lo, hi = shard_indices(a.images, rank, world)
    ids = [str(i) for i in range(a.images)]

    class Items:        # this rank's shard, drawn from a pool of 64 images: no M x 0.9 MB resident
        def __iter__(self):
            return ((ids[i], pool[i % 64]) for i in range(lo, hi))

        def __len__(self):
            return hi - lo

    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        if not a.selftest:      # warm-up: arena, streams, the first RCCL collective
            w = ExtractionPipeline(model, pre, os.path.join(d, f"warm{rank}.arrow"), batch_size=a.batch, visual_dim=F)
            w.set_global_ids([f"w{i}" for i in range(2 * a.batch * world)])
            wl, wh = shard_indices(2 * a.batch * world, rank, world)
            w.run([(f"w{i}", pool[i % 64]) for i in range(wl, wh)], n_items=2 * a.batch * world)
            torch.cuda.synchronize()
        pipe = ExtractionPipeline(model, pre, os.path.join(d, "train.arrow"), batch_size=a.batch, visual_dim=F, dataset="synthetic")
        pipe.set_global_ids(ids)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        path = pipe.run(Items(), n_items=a.images)
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        seen = 1
        if world > 1:
            t = torch.ones(1, device=None if (a.selftest or one_gpu) else f"cuda:{local_rank}")
            dist.all_reduce(t)
            seen = int(t.item())
        if rank == 0:
            from vltk_amd.extraction import load_extraction
            table, _ = load_extraction(path)
            assert table.num_rows == a.images, (table.num_rows, a.images)
            size = os.path.getsize(path)
            line = {"metric": "images/sec FRCNN feature extraction, end to end (upload + GPU pre-processing + forward + all-gather + read-back + Arrow write)",
                    "value": round(a.images / dt, 2), "unit": "images/sec", "n_gpus": world, "n_ranks_seen": seen, "images": a.images,
                    "seconds": round(dt, 3), "arrow_mb": round(size / 1e6, 1), "rows_written": table.num_rows,
                    "config": {"workload": f"configs[2] shape: {a.images} synthetic {rh}x{rw} uint8 images over {world} rank(s), ResNet-101-C4 fp16, "
                                           f"R={a.proposals}, {a.detections} detections/img, batch {a.batch} per rank" + (" [CPU selftest: stand-in model]" if a.selftest else ""),
                               "exchange": "one all_gather_into_tensor of the flat output block per step (" + ("gloo" if (a.selftest or one_gpu) else "RCCL") + ")" if world > 1 else "none (one rank)"},
                    "data": "synthetic"}
            if not a.selftest:
                feed = host_feed_costs(pool[0])
                # upload: what the loop's own host-to-device copies cost (uint8, pageable -> device), one batch
                raws = [pool[i % 64] for i in range(a.batch)]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    _ = [torch.from_numpy(r).to(model.device) for r in raws]
                    torch.cuda.synchronize()
                up = (time.perf_counter() - t1) / 5 / a.batch
                feed["upload_ms_per_image"] = round(up * 1e3, 3)
                feed["upload_images_per_s_one_thread"] = round(1.0 / up, 1)
                rate = a.images / dt / world
                if "jpeg_decode_images_per_s_per_core" in feed:
                    feed["decode_cores_per_gpu_at_this_rate"] = round(rate / feed["jpeg_decode_images_per_s_per_core"], 2)
                    feed["decode_cores_for_8_gpus_at_470_img_s"] = round(8 * 470 / feed["jpeg_decode_images_per_s_per_core"], 1)
                line["host_feed"] = feed
            os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
