#!/usr/bin/env python3
"""bench.py -- images/sec of FRCNN feature extraction (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Called plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the process starts the N ranks itself, as children
of a parent that never touches HIP, and relays rank 0's JSON line and the job's return code.

A step = one vk_forward over one per-GPU batch of synthetic 800x1333 images already resident in HBM
(ResNet-101-C4 fp16, R = 300 RPN proposals through the Res5 head, up to 100 detections per image),
followed -- for N > 1 -- by ONE all-gather of the flat output block (RCCL), overlapped with the next step's forward and
completed inside the timed region.  Images shard across ranks
(weak scaling: fixed per-GPU batch).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (dense, no sparsity)
PEAK_F16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def conv_gflop_per_image(R, arch="r101"):
    """SURVEY.md §8d algorithmic work (2*MAC) per 800x1333 image: backbone + RPN head + Res5 head (+ predictor)."""
    if arch == "x152":
        return 943.5 + 40.0 + 11.25 * R + 0.03554 * R
    return 292.4 + 40.0 + 5.857 * R + 0.03554 * R


KERNEL_NAMES = {   # kernel_timing() bucket -> (rocprofv3 kernel name, description)
    "conv_mfma256": ("conv_mfma256_kernel", "256x256 LDS-ring implicit-GEMM conv"),
    "conv3x3_panel": ("conv3x3_panel_kernel", "3x3 conv from an LDS-resident input panel, 256x256 tile"),
    "conv_duo": ("conv_duo_kernel", "1x1 conv / dual-source GEMM, 128x256 tile, two workgroups per CU"),
    "conv_mfma_f16": ("conv_mfma_kernel", "128x{64,128} tile conv"),
    "conv_mfma_f16_f32out": ("conv_mfma_kernel", "128x{64,128} tile conv, f32 out"),
    "other": ("conv_mfma_kernel", "stem / strict-mode convs"),
    "conv_ws": ("conv_ws_kernel", "1x1 conv with K <= 512, weight-stationary: 256 x K weights in registers, pixels through an LDS-DMA ring"),
    "conv_mfma256_dual": ("conv_mfma256_kernel<0, true>", "256x256 LDS-ring GEMM with two inputs (conv3 + projection shortcut, K = Cin | Cin2)"),
    "conv_gemm4": ("conv_gemm4_kernel", "1x1 conv with K >= 1024 (one or two inputs): 256x256 tile, four waves of 128x128, LDS-DMA ring"),
    "conv3x3_blk": ("conv3x3_blk_kernel", "3x3 conv over 64-channel slabs (ResNeXt grouped conv2, dense 64->64), weights in registers"),
}


def fpn_gflop_per_image(R, H=800, W=1333):
    """Algorithmic 2*MAC of the ResNet-101-FPN detector per image, by stage (analytic; bottom-up res2-res4 from SURVEY.md 8d)."""
    def ceil2(v):
        return (v + 1) // 2
    h, w = ceil2(ceil2(H)), ceil2(ceil2(W))                 # stride 4 (pad-1 max-pool)
    px = []                                                 # pixels of C2..C5 = P2..P5
    for _ in range(4):
        px.append(h * w)
        h, w = ceil2(h), ceil2(w)
    res5 = px[3] * 2 * (512 * 1024 + 9 * 512 * 512 + 512 * 2048 + 1024 * 2048 + 2 * (2048 * 512 + 9 * 512 * 512 + 512 * 2048)) / 1e9
    neck = sum(p * 2 * (c * 256 + 9 * 256 * 256) for p, c in zip(px, (256, 512, 1024, 2048))) / 1e9
    rpn_px = sum(px) + h * w                                # + P6 (every second pixel of P5)
    rpn = rpn_px * 2 * (9 * 256 * 256 + 256 * 15) / 1e9
    box = R * 2 * (12544 * 1024 + 1024 * 1024) / 1e9
    pred = R * 2 * (1024 * 1601 + 1152 * 256 + 256 * 401 + 4 * 1024) / 1e9
    return {"backbone": 292.4 + res5, "neck": neck, "rpn_head": rpn, "box_head": box, "predictor_outputs": pred}


def stage_fractions(st, B, R, arch):
    """Per stage of the forward: ms, algorithmic TFLOP/s and fraction of the f16 MFMA peak (backbone and whole
    model are asked for separately); the index stages (proposals, outputs) are latency-bound: ms only."""
    gf = {"backbone": (943.5 if arch == "x152" else 292.4) * B, "rpn_head": 40.0 * B,
          "roi_heads": (11.25 if arch == "x152" else 5.857) * R * B}
    out = {}
    for k, ms in st.items():
        e = {"ms": round(float(ms), 3)}
        if k in gf and ms > 0:
            e["tflops"] = round(gf[k] / ms, 1)
            e["frac_of_mfma_peak"] = round(gf[k] / ms / PEAK_F16_TFLOPS, 4)
        out[k] = e
    return out


PMC_TRAFFIC_FILES = ("r02_pmc_traffic.json", "r01_pmc_traffic.json")


def pmc_traffic(batch, proposals, kernel):
    """(HBM bytes per launch of the dominant kernel, source) from the committed rocprofv3 PMC passes (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950; tools/refresh_profiles.sh writes the file).  PMC counters cannot be read from inside the process, so the
    figure is NOT measured in this run: the source file is named next to it.  (None, None) if no file matches."""
    for fn in PMC_TRAFFIC_FILES:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except Exception:
            continue
        for e in (d.get("kernels") or [d]):
            if e.get("batch", d.get("batch")) == batch and e.get("proposals", d.get("proposals")) == proposals \
                    and e.get("kernel") == kernel:
                return e["hbm_bytes_per_launch"], "profiles/" + fn
    return None, None


def cpu_baseline(cfg, sd, R, det, seed, n_images=4, repeats=3):
    """The oracle (CPU restatement of the reference path, torch CPU ops) on a batch of N=4 images, median of 3 runs
    (SURVEY.md 8d).  Threads = the cores this process may use, capped at the one-GPU box's CPU share of 16 (the box shows
    every core of the host; oversubscribing them was measured 10x slower) -- VLTK_AMD_CPU_THREADS overrides."""
    import statistics
    import torch
    from oracle.frcnn_oracle import FRCNNOracle
    from vltk_amd import synthetic_images
    visible = os.cpu_count() or 1
    try:
        visible = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = int(os.environ.get("VLTK_AMD_CPU_THREADS", min(visible, 16)))
    torch.set_num_threads(cores)
    o = FRCNNOracle(cfg, sd)
    x = torch.from_numpy(synthetic_images(n_images, 800, 1333, seed=seed))
    shapes = [[800, 1333]] * n_images
    times = []
    for _ in range(repeats):
        t0 = time.time()
        out = o.forward(x, shapes)
        times.append(time.time() - t0)
    dt = statistics.median(times)
    return {"value": round(n_images / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n_images} images 800x1333 in one batch, R={R}, D={det}, fp32, oracle/frcnn_oracle.py, median of "
                      f"{repeats} runs ({', '.join('%.1f' % t for t in times)} s), {cores} threads of {visible} visible cores, "
                      f"{[int(v) for v in out['preds_per_image']]} detections"}


def selftest_launch(a):
    """What every rank does under --selftest-launch: join a gloo group, take part in one all-reduce, rank 0 prints one
    JSON line.  Exercises launch_ranks() (children, relayed line, return code) where no GPU exists."""
    import torch
    import torch.distributed as dist
    json_fd = os.dup(1)             # as main(): stdout carries the ONE JSON line, everything else goes to stderr
    os.dup2(2, 1)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.ones(1)
    if world > 1:
        dist.all_reduce(t)
    print("noise on stdout from rank", rank, flush=True)           # must not reach the caller's stdout as a JSON line
    if a.selftest_launch == "fail" and rank == world - 1:
        return 3
    if rank == 0:
        os.write(json_fd, (json.dumps({"metric": "selftest", "n_gpus": world, "n_ranks_seen": int(t.item())}) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()
    return 0


def launch_ranks(a):
    """--gpus N > 1 without a launcher: start the N ranks as children (torch.distributed.run) from this process, which
    has not imported torch or touched HIP, relay rank 0's one JSON line on stdout and return the job's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for raw in proc.stdout:
        txt = raw.decode(errors="replace").rstrip("\n")
        if txt.startswith("{") and '"metric"' in txt:
            line = txt
        elif txt:
            print(txt, file=sys.stderr)
    rc = proc.wait()
    if rc != 0:                       # a rank failed: no number is reported for the job
        return rc
    if line is None:
        return 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--proposals", type=int, default=None,
                    help="RPN.POST_NMS_TOPK_TEST = RoIs through the box head (default 300; 1000 for r101-fpn)")
    ap.add_argument("--detections", type=int, default=100)
    ap.add_argument("--head-chunk", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arch", default="r101", choices=["r101", "x152", "r101-fpn"],
                    help="r101 = the BASELINE workload (configs[1], C4: the model the reference has); x152 = ResNeXt-152 32x8d C4 "
                         "(SURVEY.md 8d config c4, extra); r101-fpn = the FPN detector (build extension, UNPINNED vs the reference, extra)")
    ap.add_argument("--selftest-launch", default=None, choices=["ok", "fail"],
                    help="CPU check of the rank launcher only: gloo ranks, no GPU, no model (tests/test_bench_launcher.py)")
    a = ap.parse_args()
    if a.proposals is None:
        a.proposals = 1000 if a.arch == "r101-fpn" else 300
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    if a.selftest_launch:
        sys.exit(selftest_launch(a))

    # HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With RCCL's streams alive, the forward's second
    # stream (res3/res4 half-batches) lands on the main stream's queue and the halves serialise: measured 374 instead of
    # 402 images/s on one GPU with a one-rank RCCL group; 16 queues restore it.  Must be set before HIP initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    # RCCL prints a version banner on stdout at communicator creation: keep stdout for the ONE JSON line
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    # validation only (one-GPU box): VLTK_AMD_BENCH_ONE_GPU=1 puts every rank on device 0 and uses gloo for the exchange
    # (RCCL needs one GPU per rank), so that the N > 1 control flow of this file can be run where only one GPU exists
    one_gpu = os.environ.get("VLTK_AMD_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_pg = world > 1 or os.environ.get("VLTK_AMD_FORCE_COLLECTIVE") == "1"     # the latter: one-rank RCCL group (validation)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from vltk_amd import FRCNN, fpn_config, make_state_dict, synthetic_images, vg_c4_config
    from vltk_amd.parallel import gather_outputs_async
    arch = dict(depth=152, num_groups=32, width_per_group=8) if a.arch == "x152" else {}
    if a.arch == "r101-fpn":
        cfg = fpn_config(post_nms_topk=a.proposals, detections=a.detections, device=f"cuda:{local_rank}")
    else:
        cfg = vg_c4_config(post_nms_topk=a.proposals, detections=a.detections, device=f"cuda:{local_rank}", **arch)
    sd = make_state_dict(cfg, seed=1234)
    model = FRCNN(cfg, precision="fp16", device=f"cuda:{local_rank}").load_state_dict(sd).eval()
    if a.head_chunk >= 0:
        model.set_option("head_chunk", a.head_chunk)
    B = a.batch
    images = torch.from_numpy(synthetic_images(B, 800, 1333, seed=0xF2C, rank=rank)).cuda(local_rank)
    shapes = torch.tensor([[800, 1333]] * B)

    pending = []
    gbuf = {}          # rotating destinations of the gathers in flight (allocated once)

    def step_gather(i):
        blk = model.forward_padded()
        if use_pg and i % 3 not in gbuf:
            gbuf[i % 3] = torch.empty(dist.get_world_size() * blk.flat.numel(), dtype=torch.uint8, device=blk.flat.device)
        return gather_outputs_async(blk, out=gbuf.get(i % 3))

    nstep = [0]
    inflight = []

    def finish(p):
        # format like a blocking call would (padded "pt" tensors on the device), then start this step's all-gather: ONE
        # collective over the flat output block on RCCL's own stream, which runs under the next step's kernels
        p.wait(padding="max_detections", return_tensors="pt", location="cuda")
        pending.append(step_gather(nstep[0]))
        nstep[0] += 1
        return pending.pop(0).wait() if len(pending) > 1 else None

    def step():
        # vk_forward_begin of step i+1 is enqueued before step i is waited for, so the device never idles while the host
        # formats a batch and launches the next one; every forward and every gather completes inside the timed region
        # (drain() before the closing barrier)
        inflight.append(model.forward_async(images, shapes))
        if len(inflight) > 1:
            finish(inflight.pop(0))

    def drain():
        out = None
        while inflight:
            finish(inflight.pop(0))
        while pending:
            out = pending.pop(0).wait()
        return out

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    drain()
    # headline pass: EXACTLY a.steps steps, no per-launch timers, no stage events
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    out = drain()
    barrier()
    dt = time.perf_counter() - t0
    # breakdown pass (outside the headline): the same steps again with a HIP event pair around every conv launch and
    # every stage; its own wall time is reported as ms_per_step_with_timers
    model.enable_kernel_timing(True)
    model.enable_stage_timing(True)
    model.kernel_timing(reset=True)
    barrier()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()
    barrier()
    dt_timers = time.perf_counter() - t1
    n_ranks_seen = dist.get_world_size() if use_pg else 1
    if use_pg:
        t = torch.tensor([dt, dt_timers], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_timers = float(t[0].item()), float(t[1].item())
    kt = model.kernel_timing()
    st = model.stage_timing_ms()                 # HIP events of the LAST step's stages (backbone / RPN head / proposals / RoI heads / outputs)
    assert out["roi_features"].shape[0] == world * B

    if rank == 0 and a.arch == "r101-fpn":
        # the FPN detector is composed on the host from stage-level C-ABI calls: stage timers only (no per-kernel buckets)
        gf = fpn_gflop_per_image(a.proposals)
        tot = sum(gf.values())
        stages = {}
        for k, ms in st.items():
            e = {"ms": round(float(ms), 3)}
            if k in gf and ms > 0:
                e["tflops"] = round(gf[k] * B / ms, 1)
                e["frac_of_mfma_peak"] = round(gf[k] * B / ms / PEAK_F16_TFLOPS, 4)
            stages[k] = e
        ach = tot * world * B * a.steps / dt / 1e3
        line = {
            "metric": "images/sec FRCNN feature extraction, 800x1333 batch",
            "value": round(world * B * a.steps / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "ms_per_step_with_timers": round(dt_timers / a.steps * 1e3, 3), "n_ranks_seen": n_ranks_seen,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "EXTRA, UNPINNED vs the reference (it has no FPN model, SURVEY.md D1): ResNet-101-FPN Faster R-CNN fp16 "
                                   f"(detectron2 layout: FPN neck, RPN over P2-P6, RoIAlign 7x7, 2-FC box head), {B} synthetic 800x1333 "
                                   f"images per GPU per step, R={a.proposals} proposals, max {a.detections} detections/img, seeded synthetic weights",
                       "global_batch": world * B, "parallelism": f"image-sharded x{world}, all-gather of output blocks"},
            "roofline": {"bound": "mfma", "achieved": round(ach / world, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / world / PEAK_F16_TFLOPS, 4), "traffic": None,
                         "kernel": "whole forward (algorithmic conv / GEMM flops of the model over the step time; this path has stage "
                                   "timers only)",
                         "alg_gflop_per_image": {k: round(v, 1) for k, v in dict(gf, total=tot).items()},
                         "stages_last_step": stages},
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    elif rank == 0:
        # res3 / res4 run as two half-batches on two streams: those launches overlap in time (their summed durations are
        # not wall time), so they are reported apart and the per-kernel figures cover the launches that ran alone
        conc = kt.pop("two_stream_backbone")
        dom_key = max(kt, key=lambda k: kt[k]["ms"])          # the kernel with the most GPU time in the timed region
        dom = kt[dom_key]
        dom_name, dom_desc = KERNEL_NAMES[dom_key]
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        traffic, traffic_src = pmc_traffic(B, a.proposals, dom_name)
        all_ms = sum(v["ms"] for v in kt.values())
        all_fl = sum(v["flops"] for v in kt.values())
        line = {
            "metric": "images/sec FRCNN feature extraction, 800x1333 batch",
            "value": round(world * B * a.steps / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "ms_per_step_with_timers": round(dt_timers / a.steps * 1e3, 3), "n_ranks_seen": n_ranks_seen,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": ("configs[1]: ResNet-101-C4 fp16" if a.arch == "r101" else "EXTRA (SURVEY.md 8d c4): ResNeXt-152 32x8d C4 fp16") +
                                   " (the reference has no FPN: SURVEY.md D1), "
                                   f"{B} synthetic 800x1333 images per GPU per step, R={a.proposals} RPN proposals "
                                   f"through the Res5 head, max {a.detections} detections/img, seeded synthetic weights",
                       "global_batch": world * B, "parallelism": f"image-sharded x{world}, all-gather of output blocks"},
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_F16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_launch": round(dom["bytes"] / max(dom["launches"], 1)),
                         "alg_gflop_per_launch": round(dom["flops"] / max(dom["launches"], 1) / 1e9, 2),
                         "kernel": f"{dom_name} ({dom_desc}, f16 in / f32 acc; all its launches in the timed region)",
                         "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 5),
                         "alg_gflop_per_image": round(conv_gflop_per_image(a.proposals, a.arch), 1),
                         "per_kernel": {KERNEL_NAMES[k][0] + ("" if k in ("conv_mfma256", "conv3x3_panel", "conv_duo", "conv3x3_blk", "conv_ws", "conv_mfma256_dual", "conv_gemm4") else ":" + k):
                                        {"launches": v["launches"], "ms_per_step": round(v["ms"] / a.steps, 3),
                                         "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                                        for k, v in kt.items() if v["ms"] > 0},
                         "two_stream_backbone_launches": {"launches": conc["launches"], "summed_ms_per_step": round(conc["ms"] / a.steps, 3),
                                                          "gflop_per_step": round(conc["flops"] / a.steps / 1e9, 1),
                                                          "note": "res3/res4 half-batches on two streams; durations overlap, see "
                                                                  "stages_last_step.backbone for their wall time"},
                         "stages_last_step": stage_fractions(st, B, a.proposals, a.arch),
                         "all_conv_kernels_that_ran_alone": {"ms_per_step": round(all_ms / a.steps, 3),
                                              "tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2) if all_ms else 0.0,
                                              "share_of_step": round(all_ms / (dt_timers * 1e3), 4)},
                         "note": "per-kernel and per-stage figures are from the breakdown pass (timers on), value / ms_per_step "
                                 "from the headline pass (timers off)"},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, a.proposals, a.detections, seed=0xF2C)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
