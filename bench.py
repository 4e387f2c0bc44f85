#!/usr/bin/env python3
"""Headline benchmark: images/s end-to-end (640 -> 224 detect + classify), BASELINE.json configs[1]:
YOLOv8n + ViT-B/16, 640x640, batch 32 per GPU, bf16, synthetic data, random-init weights.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; images shard across ranks with no
   data-path collective -> "scaling": "weak")

One step = the whole hot path over one batch already resident in HBM:
stem+backbone+neck+head -> DFL decode -> EfficientNMS(0.25/0.65/100) -> restore/filter(0.35)/int ->
custom_nms(0.45) dedupe -> inflate -> crop+nearest-resize+normalize -> ViT-B/16 -> wrapper head -> argmax.
Crops per image are capped at --crops (default 4, SURVEY.md 8(d) "fixed-R mode") so the
classifier work per step is deterministic; the cap is part of the reported config.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0        # bf16 dense, MI355X_MICROARCH.md chip table
DOMINANT_KERNEL = "gemm_p9_kernel"       # what profiles/pmc_traffic.json must have been collected for
ROUND_TAG = "r03"


def _cpu_loop(imgs, yolo_sd, vit_sd, vit_name, crops, budget_s, max_images):
    """The oracle's batch-1 loop (oracle/pipeline.py::run_image, stage by stage so that each stage can be timed)."""
    import numpy as np
    from oracle import boxes as ob, pipeline as op, vit as ov, yolo as oy
    st = {"detect": 0.0, "nms_post": 0.0, "crop": 0.0, "classify": 0.0}
    n, t0 = 0, time.perf_counter()
    while n < max_images and (time.perf_counter() - t0) < budget_s:
        img = imgs[n]
        S = img.shape[0]
        ta = time.perf_counter()
        boxes, scores = oy.decode(oy.forward_raw(yolo_sd, oy.blob(img[None]), "n", 5), 5, S)
        tb = time.perf_counter()
        num, bb, sc, lb = ob.efficient_nms(boxes, scores)
        dets = op.post_stages(num[0, 0], bb[0], sc[0], lb[0], 1.0, (0.0, 0.0), (S, S), max_crops=crops)
        tc = time.perf_counter()
        cr = [ob.crop_resize_normalize(img.numpy(), d["rect"]) for d in dets if d["ok"]]
        td = time.perf_counter()
        if cr:
            ov.wrapper_forward(vit_sd, torch.from_numpy(np.stack(cr)), vit_name).argmax(1)
        te = time.perf_counter()
        st["detect"] += tb - ta; st["nms_post"] += tc - tb; st["crop"] += td - tc; st["classify"] += te - td
        n += 1
    dt = time.perf_counter() - t0
    return n, dt, {k: round(v / max(n, 1) * 1e3, 2) for k, v in st.items()}


def cpu_baseline(yolo_sd, vit_sd, vit_name, crops, budget_s=12.0, max_images=8):
    """The oracle (CPU restatement, fp32, batch-1 loop like the reference's per-image loop) timed on this box's host cores:
    all cores (the headline `value`) and ONE thread, each with per-stage milliseconds."""
    from oracle import pipeline as op
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randint(0, 256, (max_images + 1, 640, 640, 3), generator=g, dtype=torch.uint8)
    all_cores = torch.get_num_threads()
    with torch.no_grad():
        op.run_image(imgs[0], yolo_sd, [vit_sd], vit_name, max_crops=crops)      # warm-up (page-in, thread pool)
        n, dt, stages = _cpu_loop(imgs[1:], yolo_sd, vit_sd, vit_name, crops, budget_s, max_images)
        torch.set_num_threads(1)
        try:
            n1, dt1, stages1 = _cpu_loop(imgs[1:], yolo_sd, vit_sd, vit_name, crops, budget_s * 0.7, 3)
        finally:
            torch.set_num_threads(all_cores)
    return {"value": n / dt, "unit": "images/s", "cores": all_cores, "kind": "port",
            "note": "oracle port; the reference's own CPU path is not runnable here (timm / ultralytics / TensorRT absent)",
            "sample": f"{n} synthetic 640x640 images, batch-1 fp32 loop, {crops} crops/image, {dt:.1f}s",
            "ms_per_image_by_stage": stages,
            "single_thread": {"value": n1 / dt1, "unit": "images/s", "cores": 1,
                              "sample": f"{n1} images, {dt1:.1f}s", "ms_per_image_by_stage": stages1}}


# HIP maps streams round-robin onto GPU_MAX_HW_QUEUES hardware queues (default 4); the schedule below uses 4 streams next to
# the default one, and two streams that share a queue do not overlap.  Must be set before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def bench_train(args, rank, world, dev, dist):
    """BASELINE.json configs[2]: ViT-B/16 fine-tune fwd+bwd+SGD, 224x224, 32 crops per GPU (256 at DP=8), bf16 compute,
    fp32 master weights, gradient all-reduce over RCCL overlapped with backward."""
    import yvhip
    from yvhip import engines
    from yvhip.dist import max_over_ranks
    from yvhip.training import VitTrainer
    name, R = "vit_base_patch16_224", args.batch
    tr = VitTrainer(engines.init_vit_wrapper_state(name, 5, seed=42), name, 5, device=str(dev))
    g = torch.Generator().manual_seed(4321 + rank)
    patches = (torch.rand(R * tr.tok, 768, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    labels = torch.randint(0, 5, (R,), generator=g, dtype=torch.int32).to(dev)
    for _ in range(max(args.warmup, 1)):
        tr.step(patches, labels, 1e-4)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = tr.step(patches, labels, 1e-4)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = max_over_ranks(time.perf_counter() - t0, None if os.environ.get("YV_BENCH_REHEARSAL") == "1" else dev)
    if rank == 0:
        flop = 3.0 * 35.13e9 * R                       # fwd + bwd ~ 3 x forward
        print(json.dumps({"metric": "ViT-B/16 fine-tune images/sec (fwd+bwd+SGD)", "value": world * R * args.steps / dt,
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "ViT-B/16 fine-tune fwd+bwd, 224x224 (BASELINE.json configs[2])",
                                     "batch_per_gpu": R, "global_batch": R * world, "parallelism": f"dp{world}",
                                     "loss": float(loss[0])},
                          # whole-step figure: model flops (3 x forward) over the step time, against the dense bf16 MFMA peak; the
                          # per-kernel split is profiles/r03_train_vit_kernel_stats.csv (12 GEMMs per block of 22 GFLOP each at
                          # 6,304 rows: forward on gemm_p9 / gemm_dma, data gradients on the transposed weight mirror, weight
                          # gradients on gemm_tn)
                          "roofline": {"bound": "mfma", "achieved": flop * args.steps / dt / 1e12, "peak": MFMA_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": flop * args.steps / dt / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None,
                                       "kernel": "whole step (model flops / step time)"},
                          "model_tflops_per_gpu": flop * args.steps / dt / 1e12}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_train_yolo(args, rank, world, dev, dist):
    """BASELINE.json configs[3]: YOLOv8s (nc 80) backbone+head training step at 640x640, 16 images per GPU (128 at
    DP=8): un-fused forward (BatchNorm batch statistics), v8 detection loss, backward, gradient all-reduce, SGD."""
    from yvhip.dist import max_over_ranks
    from yvhip.yolo_training import YoloTrainer, init_yolo_train_state
    scale, nc, S = "s", 80, 640
    B = 16 if args.batch == 32 else args.batch
    tr = YoloTrainer(init_yolo_train_state(scale, nc, seed=42), scale=scale, nc=nc, size=S, batch=B, lr=1e-4, device=str(dev))
    g = torch.Generator().manual_seed(4321 + rank)
    images = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to(dev)
    G = 8
    ctr = torch.rand(B, G, 2, generator=g) * S
    wh = torch.rand(B, G, 2, generator=g) * 240 + 16
    gtb = torch.cat([(ctr - wh / 2).clamp(0, S), (ctr + wh / 2).clamp(0, S)], -1).to(dev)
    gtl = torch.randint(0, nc, (B, G), generator=g, dtype=torch.int32).to(dev)
    gtn = torch.full((B,), G, dtype=torch.int32).to(dev)
    for _ in range(max(args.warmup, 1)):
        tr.step(images, gtb, gtl, gtn)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step(images, gtb, gtl, gtn)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = max_over_ranks(time.perf_counter() - t0, None if os.environ.get("YV_BENCH_REHEARSAL") == "1" else dev)
    if rank == 0:
        flop = 3.0 * 28.60e9 * B                       # fwd + bwd ~ 3 x forward
        print(json.dumps({"metric": "YOLOv8s training images/sec (fwd+loss+bwd+SGD)", "value": world * B * args.steps / dt,
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "YOLOv8s(nc=80) train step, 640x640 (BASELINE.json configs[3])",
                                     "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                                     "boxes_per_image": G, "loss": [float(v) for v in loss.cpu()]},
                          "roofline": {"bound": "mfma", "achieved": flop * args.steps / dt / 1e12, "peak": MFMA_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": flop * args.steps / dt / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None,
                                       "kernel": "whole step (model flops / step time); per-kernel split: profiles/r03_train_yolo_kernel_stats.csv"},
                          "model_tflops_per_gpu": flop * args.steps / dt / 1e12}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_postproc(args, dev):
    """NMS and crop-gather against the HBM roof (north_star: >= 60 % of peak HBM on NMS / crop at batch 256).  Inputs are the
    SURVEY.md 8(d) synthetic candidate sets (independent of the random-weight detector): per image A = 8400 boxes, centres
    uniform in [0,640)^2, w,h ~ U[16,256] clipped to the image, scores ~ Beta(0.5,4) (3-5 % above 0.25), nc = 5, seed 4321;
    crop-gather: 4 crops per image from those boxes.  Algorithmic bytes: NMS 304.8 kB per image (boxes + scores
    read once + the 2,404-byte result), crop 451,584 B per crop (224*224*3 u8 read + bf16 write).  Each stage is timed with
    HIP events around `steps` back-to-back launches on the current stream."""
    import yvhip
    B, A, nc, S = args.batch, 8400, 5, 640
    g = torch.Generator().manual_seed(4321)
    ctr = torch.rand(B, A, 2, generator=g) * S
    wh = torch.rand(B, A, 2, generator=g) * 240 + 16
    boxes = torch.cat([(ctr - wh / 2).clamp(0, S), (ctr + wh / 2).clamp(0, S)], -1).contiguous().to(dev)
    scores = torch.distributions.Beta(0.5, 4.0).sample((B, A, nc)).to(torch.float32)
    frac = float((scores > 0.25).float().mean())
    scores = scores.to(dev)
    images = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to(dev)
    R = args.crops
    cl = torch.zeros(B * R, 6, dtype=torch.int32)
    for b in range(B):
        for k in range(R):
            x0, y0, x1, y1 = [int(v) for v in boxes[b, k].tolist()]
            cl[b * R + k] = torch.tensor([b, x0, y0, max(x1, x0 + 8), max(y1, y0 + 8), k])
    cl = cl.to(dev)
    total = torch.tensor([B * R], dtype=torch.int32, device=dev)
    out = torch.zeros((B * R * 196, 768), dtype=torch.bfloat16, device=dev)

    def timed(fn):
        for _ in range(max(args.warmup, 2)):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.steps * 1e3            # us per call

    res = {}
    res["nms"] = timed(lambda: yvhip.efficient_nms(boxes, scores))
    res["nms_single_kernel"] = timed(lambda: yvhip.efficient_nms(boxes, scores, single_kernel=True))
    res["crop"] = timed(lambda: yvhip.crop_resize_norm(images, cl, total, B * R, 224, 16, layout=2, out=out))
    num = yvhip.efficient_nms(boxes, scores)[0]
    nms_bytes = B * (A * (16 + 4 * nc) + 2404)
    crop_bytes = B * R * 451584
    peak = 8000.0
    line = {"metric": "NMS + crop-gather HBM throughput (post-processing stages of the hot path)", "unit": "GB/s",
            "value": nms_bytes / res["nms"] * 1e-3, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "dtype": "f32 boxes / u8 pixels", "data": "synthetic",
            "config": {"workload": f"EfficientNMS(0.25/0.65/100) on {B} x 8400 x {nc} candidates + {R} crops/image 640->224, "
                                   "SURVEY 8(d) synthetic sets", "batch": B, "candidates_above_threshold": frac,
                       "mean_num_dets": float(num.float().mean())},
            "roofline": {"bound": "hbm", "achieved": nms_bytes / res["nms"] * 1e-3, "peak": peak, "unit": "GB/s",
                         "frac": nms_bytes / res["nms"] * 1e-3 / peak, "traffic": None,
                         "kernel": "en2_filter + en2_front + en2_tail (EfficientNMS, per call)",
                         "us_per_call": res["nms"], "alg_bytes_per_call": nms_bytes},
            "crop_roofline": {"bound": "hbm", "achieved": crop_bytes / res["crop"] * 1e-3, "peak": peak, "unit": "GB/s",
                              "frac": crop_bytes / res["crop"] * 1e-3 / peak, "kernel": "crop_kernel<2>",
                              "us_per_call": res["crop"], "alg_bytes_per_call": crop_bytes},
            "nms_single_kernel_us": res["nms_single_kernel"]}
    print(json.dumps(line), flush=True)


def launch_ranks(n: int) -> int:
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, the contract of
    torch.distributed.run), same command line; rank 0 prints the JSON line.  Returns the worst child exit code."""
    import socket
    import subprocess
    rehearsal = os.environ.get("YV_BENCH_REHEARSAL") == "1"
    have = torch.cuda.device_count()
    if have < n and not rehearsal:
        raise SystemExit(f"--gpus {n}: only {have} device(s) visible (YV_BENCH_REHEARSAL=1 maps every rank to cuda:0 over gloo)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def live_traffic(argv_tail):
    """HBM-side bytes per launch of the dominant kernel, measured NOW: two short child runs of this very command under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (their own passes, with --kernel-trace only), started before this process
    has touched the GPU.  bytes = 2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE reads half of a wide
    coalesced read stream) + WRITE_SIZE, averaged over the kernel's dispatches.  None if the profiler is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    import signal
    prof = shutil.which("rocprofv3")
    if prof is None:
        return {"failed": "rocprofv3 not found"}
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return {"failed": "already running under a profiler"}
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="yv_pmc_", dir="/tmp")
        proc = None
        try:
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__)] + argv_tail + ["--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-traffic"]
            env = dict(os.environ, TMPDIR="/tmp")
            # own session: on a timeout the WHOLE group (profiler + the profiled bench it started) is killed and reaped before the
            # timed run begins - a surviving child would share the GPU with it
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rc = proc.wait(timeout=150)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                print(f"[bench] live traffic pass {counter}: timed out, process group killed", file=sys.stderr)
                return {"failed": f"{counter} pass timed out"}
            if rc != 0:
                print(f"[bench] live traffic pass {counter}: rocprofv3 exit code {rc}", file=sys.stderr)
                return {"failed": f"{counter} pass: rocprofv3 exit code {rc}"}
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                rows += [float(x["Counter_Value"]) for x in csv.DictReader(open(f))
                         if DOMINANT_KERNEL in x["Kernel_Name"] and x["Counter_Name"] == counter]
            if not rows:
                print(f"[bench] live traffic pass {counter}: no {DOMINANT_KERNEL} dispatches in the counter file", file=sys.stderr)
                return {"failed": f"{counter} pass: no {DOMINANT_KERNEL} dispatches recorded"}
            vals[counter] = (sum(rows) / len(rows), len(rows))
        except Exception as e:                                    # noqa: BLE001 - any failure here only costs the live number
            if proc is not None and proc.poll() is None:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
            print(f"[bench] live traffic pass {counter}: {type(e).__name__}: {e}", file=sys.stderr)
            return {"failed": f"{counter} pass: {type(e).__name__}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"bytes_per_launch": (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0,
            "launches": [vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--crops", type=int, default=4, help="crops classified per image (cap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic live (two short rocprofv3 --pmc child runs before the timed run)")
    ap.add_argument("--no-overlap", action="store_true", help="single stream: no detector/classifier overlap across batches")
    ap.add_argument("--no-split", action="store_true", help="keep the classifier of a batch on one stream (no half-batch overlap)")
    ap.add_argument("--models", choices=["base", "large"], default="base",
                    help="base = YOLOv8n + ViT-B/16 (configs[1], the headline); large = YOLOv8m + ViT-L/16 in bf16 "
                         "(the model pair of configs[4]; its FP8 arithmetic is not built, so this is NOT that config's number)")
    ap.add_argument("--dtype", choices=["bf16", "mxfp8"], default="bf16",
                    help="arithmetic of the classifier's block linears; mxfp8 = OCP e4m3 + E8M0 block scales (configs[4]); the "
                         "headline metric is defined on bf16")
    ap.add_argument("--mode", choices=["infer", "train", "train-yolo", "postproc"], default="infer",
                    help="infer = headline metric (configs[1]); train = ViT-B/16 fine-tune step (configs[2]); "
                         "train-yolo = YOLOv8s training step (configs[3])")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched HIP yet (importing torch
        # and counting devices does not initialise it), and the ranks are started as CHILD processes, never by exec.
        return launch_ranks(args.gpus)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} disagrees with WORLD_SIZE={os.environ['WORLD_SIZE']}")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    live = None
    if (args.mode == "infer" and args.models == "base" and args.dtype == "bf16" and world == 1 and "RANK" not in os.environ
            and not args.no_traffic and os.environ.get("YV_BENCH_DRY") != "1"):
        # before this process initialises HIP (importing torch and counting devices does not): the children are started, never exec'ed
        tail = [a for f in ("--batch", "--crops") for a in (f, str(getattr(args, f[2:])))]
        tail += [f for f, on in (("--no-overlap", args.no_overlap), ("--no-split", args.no_split)) if on]
        live = live_traffic(tail)
    if os.environ.get("YV_BENCH_DRY") == "1":        # launcher test (tests/test_dist_cpu.py): report the rank layout, touch no GPU
        print(json.dumps({"rank": rank, "world": world, "local": local, "gpus": args.gpus}), flush=True)
        return 0
    # rehearsal of the multi-rank control flow on a ONE-GPU box: YV_BENCH_REHEARSAL=1 maps every rank to cuda:0 and uses
    # gloo (RCCL refuses two ranks on one device); never used by the driver
    rehearsal = os.environ.get("YV_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import yvhip
    from yvhip import engines
    from yvhip.pipeline import DetectClassifyPipeline, PipelinedRunner

    if args.mode == "postproc":
        return bench_postproc(args, dev)
    if args.mode == "train":
        return bench_train(args, rank, world, dev, dist)
    if args.mode == "train-yolo":
        return bench_train_yolo(args, rank, world, dev, dist)

    large = args.models == "large"
    vit_name = "vit_large_patch16_224" if large else "vit_base_patch16_224"
    yscale = "m" if large else "n"
    yolo_sd = engines.init_yolo_state(yscale, 5, seed=42, head_gain=4.0)
    vit_sd = engines.init_vit_wrapper_state(vit_name, 5, seed=42)
    yolo = engines.YoloEngine(yolo_sd, yscale, 5, 640, device=str(dev))
    vit = engines.VitEngine(vit_sd, vit_name, 5, device=str(dev), dtype=args.dtype)
    B, R = args.batch, args.crops
    pipe = DetectClassifyPipeline(yolo, [vit], max_crops_per_image=R)
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.randint(0, 256, (B, 640, 640, 3), generator=g, dtype=torch.uint8).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    runner = None if args.no_overlap else PipelinedRunner(pipe, split_classifier=not args.no_split)
    step = pipe if runner is None else runner.submit
    for _ in range(max(args.warmup, 1)):
        out = step(images)
    torch.cuda.synchronize()
    crops_step = int(out["crop_total"][0])

    # HIP events around every launch of the dominant kernel (the 128x128 MFMA GEMM instance)
    # HIP events of the dominant kernel are attached to its launches (hipExtLaunchKernel: timestamps of the kernel's own
    # dispatch packet).  A hipEventRecord pair around every launch would be two barrier packets per launch: ~1.5 ms per
    # step and it serialises the concurrent half-batches (measured: 9.77 ms vs 8.22 ms clean).  As a further precaution
    # the timed region carries the events on a sample of its steps only (2 of them for K >= 6).
    recs = []
    hook = lambda M, N, K, e0, e1: recs.append((2.0 * M * N * K, e0, e1))
    sampled = {args.steps // 3, (2 * args.steps) // 3} if args.steps >= 6 else {args.steps - 1}
    if os.environ.get("BENCH_NO_HOOK"):
        sampled = set()
    yvhip.reserve_events(2 * 128 * max(len(sampled), 1))            # ~100 instrumented launches per sampled step
    barrier()
    torch.cuda.synchronize()
    base_ev = yvhip.HipEvent()
    base_ev.record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        yvhip.LINEAR_HOOK = hook if i in sampled else None
        out = step(images)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    yvhip.LINEAR_HOOK = None
    from yvhip.dist import max_over_ranks
    dt = max_over_ranks(dt, None if os.environ.get("YV_BENCH_REHEARSAL") == "1" else dev)

    if rank == 0:
        flops = sum(f for f, _, _ in recs)
        ms = sum(e0.elapsed_time(e1) for _, e0, e1 in recs)
        n_launch = len(recs)
        # The classifier runs as two concurrent half-batches, so launches of the kernel overlap in time and each one's
        # event-to-event duration covers a period in which it owns only part of the chip.  Rate of the kernel = its
        # algorithmic flops / the time during which at least one launch of it is executing (union of the intervals).
        from yvhip.dist import union_length
        busy = union_length((base_ev.elapsed_time(e0), base_ev.elapsed_time(e1)) for _, e0, e1 in recs)
        achieved = flops / (busy * 1e-3) / 1e12 if busy > 0 else 0.0
        # HBM bytes per launch of the dominant kernel come from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over THIS command
        # (tools/profile_summary.py -> profiles/pmc_traffic.json).  The file names the kernel and the round it was collected
        # for; anything else (other kernel, older round) is reported as null instead of a stale number.
        traffic, traffic_source = None, None
        live_failed = live.get("failed") if live is not None else None
        if live is not None and live_failed is None:
            traffic = live["bytes_per_launch"]
            traffic_source = (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command before the timed run, "
                              f"{live['launches'][0]} / {live['launches'][1]} dispatches; 2 x FETCH + WRITE")
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if traffic is None and os.path.exists(pmc) and args.dtype == "bf16" and not large:
            try:
                tj = json.load(open(pmc))
                if tj.get("kernel") == DOMINANT_KERNEL and tj.get("round") == ROUND_TAG:
                    traffic = tj.get("bytes_per_launch")
                    traffic_source = ("fallback: profiles/pmc_traffic.json (rocprofv3 passes of this round over this command)" +
                                      (f"; live passes failed: {live_failed}" if live_failed else "; live passes not requested"))
            except Exception:
                traffic = None
        line = {
            "metric": "images/sec end-to-end (640->224 detect+classify)",
            "value": world * B * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "mxfp8 (classifier block linears) + bf16",
            "data": "synthetic",
            "config": {"workload": ("YOLOv8m(nc=5)+ViT-L/16 end-to-end inference, 640x640 (model pair of BASELINE.json "
                                    "configs[4]); classifier block linears in " + args.dtype) if large else
                                   ("YOLOv8n(nc=5)+ViT-B/16 end-to-end inference, 640x640, bf16 "
                                    "(BASELINE.json configs[1])" if args.dtype == "bf16" else
                                    "YOLOv8n(nc=5)+ViT-B/16 end-to-end inference, 640x640, classifier block linears in "
                                    "mxfp8 (NOT the headline configuration, which is bf16)"), "batch_per_gpu": B, "global_batch": B * world,
                       "crops_per_image": R, "crops_per_step_rank0": crops_step, "parallelism": f"dp{world}",
                       "schedule": "single stream" if runner is None else
                                   ("HIP streams: detector of batch i+1 (high priority) overlaps classifier of batch i" +
                                    ("" if args.no_split else "; classifier runs as two concurrent half-batches")),
                       "weights": "random-init, seed 42",
                       "gemm": "persistent free-running kernel (gemm_p9), tile height per launch" +
                               ("" if runner is None else ", 208 of 256 CUs (the rest stay free for the concurrent streams)")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS * (2.0 if args.dtype == "mxfp8" else 1.0),
                         "unit": "TFLOP/s", "frac": achieved / (MFMA_PEAK_TFLOPS * (2.0 if args.dtype == "mxfp8" else 1.0)),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": ("gemm_p9_kernel (qkv / proj / fc1 / fc2) + gemm_dma_kernel<128,128> (patch-embed, head)" if args.dtype == "bf16" else
                                    "gemm_mx_kernel<128,128> (block linears, block-scaled MFMA peak) + gemm_dma_kernel (patch-embed, head)"),
                         "launches": n_launch,
                         "avg_launch_us": ms * 1e3 / max(n_launch, 1), "kernel_busy_ms_per_step": busy / max(len(sampled), 1), "instrumented_steps": len(sampled),
                         "concurrent_launches": runner is not None and not args.no_split,
                         "alg_flop_per_launch": flops / max(n_launch, 1)},
        }
        if world == 1 and not args.no_cpu_baseline and not large:
            line["cpu_baseline"] = cpu_baseline(yolo_sd, vit_sd, vit_name, R)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
