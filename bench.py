"""Headline benchmark: images/sec of YOLOv3-416 inference including head decode + NMS on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (fused stem -> Darknet-53 -> FPN -> heads -> decode -> batched NMS
-> top-100 gather) over one resident batch of 64 synthetic uint8 images per GPU.  Steps run in the engine's
throughput mode (two sets of activation buffers: the host reads step i's candidate count while step i+1's
convolutions are already queued); all K steps, including the last one's NMS, finish inside the timed region.  Images shard across
GPUs with no data-path collective (weak scaling); the only collectives are the barriers and the MAX over
ranks of the timed region.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 64
SIZE = 416
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak for bf16 on MI355X


def _dist_setup(n_gpus: int):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # YOLO_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks
    # (ranks then share cards); the measured configuration is always nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("YOLO_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local if world > 1 else 0)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    return world, rank, local


def _barrier(world: int):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def _dominant_kernel_roofline(eng, plan, convs, images, mean, std, passes: int = 3):
    """The convolutions run as instances of one implicit-GEMM kernel that differ in their workgroup tile
    (conv_igemm_kernel<bf16, channels, pixels, ...>; yolo_conv_tile_shape says which launch uses which).  Every
    launch of the 128-channel instances is timed where it runs: the whole plan is replayed in order with a HIP event
    pair (on the launch stream) around each of them, so every launch sees the cache state it sees in a real step;
    the cost of an empty event pair is measured the same way and subtracted.  The instance with the most time is
    the dominant kernel: achieved = its algorithmic FLOPs (2 * pixels * Cout * Cin * k * k per launch) / its time.
    The other instances are reported next to it."""
    from yolo_amd import ops
    stream = torch.cuda.current_stream()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    timed = []  # (step index, tile pixels, flops, bytes)
    for i, (meta, step) in enumerate(zip(plan.meta, plan.steps)):
        c = convs[meta["name"]]
        if c.in_channels % 64 != 0:
            continue
        up = 2 if meta["up"] else 1
        M = plan.N * (meta["y"].H // up) * (meta["y"].W // up)
        tc, tp = ops.conv_tile_shape(M, c.in_channels, c.out_channels, c.kernel_size, meta["x"].buf.dtype)
        if tc != 128:
            continue
        es = meta["x"].buf.element_size()
        by = (plan.N * meta["x"].H * meta["x"].W * c.in_channels * es
              + c.out_channels * c.in_channels * c.kernel_size ** 2 * es
              + plan.N * meta["y"].H * meta["y"].W * c.out_channels * meta["y"].buf.element_size()
              + (M * c.out_channels * es if meta["residual"] is not None else 0))
        timed.append((i, tp, 2.0 * M * c.out_channels * c.in_channels * c.kernel_size ** 2, by, c.kernel_size))
    which = {i: k for k, (i, _, _, _, _) in enumerate(timed)}
    ms = [0.0] * len(timed)
    empty = 0.0
    for _ in range(passes):
        pairs = [(ev(), ev()) for _ in timed]
        e0, e1 = ev(), ev()
        eng.run_backbone_prefix(images, mean, std, plan)  # fused stem + first-stage launches
        for i, step in enumerate(plan.steps):
            k = which.get(i)
            if k is None:
                step()
            else:
                pairs[k][0].record(stream)
                step()
                pairs[k][1].record(stream)
        e0.record(stream)
        e1.record(stream)
        torch.cuda.synchronize()
        empty += e0.elapsed_time(e1) / passes
        for k, (a, b) in enumerate(pairs):
            ms[k] += a.elapsed_time(b) / passes
    acc, by_k = {}, {}
    for (i, tp, fl, by, ks), t in zip(timed, ms):
        for a in (acc.setdefault(tp, [0.0, 0.0, 0, 0.0]), by_k.setdefault((tp, ks), [0.0, 0.0, 0, 0.0])):
            a[0] += fl
            a[1] += max(t - empty, 1e-6)
            a[2] += 1
            a[3] += by
    dom = max(acc, key=lambda tp: acc[tp][1])
    flops, t_ms, launches, by = acc[dom]
    achieved = flops / (t_ms * 1e-3) / 1e12
    out = dict(bound="mfma", achieved=round(achieved, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
               frac=round(achieved / PEAK_BF16_TFLOPS, 4), traffic=None,
               kernel=f"conv_igemm_kernel<bf16,128,{dom},2,2,false>", launches_per_step=launches,
               avg_launch_us=round(t_ms * 1e3 / launches, 2), flop_per_launch=round(flops / launches / 1e9, 3),
               algorithmic_bytes_per_launch=round(by / launches), event_pair_overhead_us=round(empty * 1e3, 2))
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):  # HBM bytes per launch from separate rocprofv3 --pmc passes of this same command
        with open(pmc) as f:
            rec = json.load(f)
        hit = rec.get("kernels", {}).get(out["kernel"])
        if hit:
            out["traffic"] = hit["hbm_bytes_per_launch"]
            out["traffic_source"] = "profiles/pmc_traffic.json: " + rec.get("source", "")
    # the same instance serves long-K (3x3, matrix-bound) and short-K (1x1, memory-bound) layers: both parts of its time
    out["by_filter_size"] = {
        f"{ks}x{ks}": dict(achieved=round(a[0] / (a[1] * 1e-3) / 1e12, 1), launches_per_step=a[2],
                           avg_launch_us=round(a[1] * 1e3 / a[2], 2),
                           algorithmic_gbytes_per_s=round(a[3] / (a[1] * 1e-3) / 1e9))
        for (tp, ks), a in sorted(by_k.items()) if tp == dom}
    out["other_instances"] = {
        f"128x{tp}": dict(achieved=round(a[0] / (a[1] * 1e-3) / 1e12, 1), launches_per_step=a[2],
                          avg_launch_us=round(a[1] * 1e3 / a[2], 2))
        for tp, a in sorted(acc.items()) if tp != dom}
    return out


def _cpu_baseline(model, cfg, n_images: int = 16, reps: int = 14):
    """The CPU oracle (plain torch fp32 restatement of the reference + C NMS) on a bounded sample of the
    same workload, timed on this box's host cores."""
    import workload
    from oracle import nms_c, yolo_ref as R
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    imgs = [im.float() for im in workload.synthetic_images(n_images, SIZE, SIZE, seed=1)]
    kw = dict(num_classes=cfg.MODEL.YOLOV3.NUM_CLASSES, score_thr=cfg.MODEL.YOLOV3.TEST_SCORE_THRESH,
              nms_thr=cfg.MODEL.YOLOV3.TEST_NMS_THRESH, nms_fn=nms_c.nms, device_kind="gpu")
    with torch.no_grad():
        R.inference(imgs[:2], sd, **kw)  # warm-up (thread pool, allocator)
        t0 = time.time()
        for _ in range(reps):
            R.inference(imgs, sd, **kw)
        dt = time.time() - t0
    return dict(value=round(n_images * reps / dt, 2), unit="images/sec", cores=cores, kind="port",
                sample=f"{reps} x batch {n_images} of the same 416x416 workload, torch {torch.__version__} fp32 CPU ops + C NMS")


def _train_throughput(steps: int = 4, warmup: int = 2, batch: int = 64):
    """Secondary metric of the same hot path (training images per second): forward + backward + SGD step of
    configs/yolov3.yaml at 416x416, batch 64, bf16, synthetic COCO-shaped targets, on this one GPU."""
    import workload
    from yolo_amd.trainer import Trainer, synthetic_batches
    model, cfg = workload.build_model(config="yolov3.yaml", dtype="bf16", device=f"cuda:{torch.cuda.current_device()}",
                                      opts=["OUTPUT_DIR", "", "SOLVER.CHECKPOINT_PERIOD", 0, "SOLVER.BASE_LR", 1e-4])
    data = next(synthetic_batches(cfg, batch, SIZE, seed=7, device=model.device))
    tr = Trainer(cfg, model, iter(lambda: data, None), log=lambda m: None)
    for _ in range(warmup):
        tr.run_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = tr.run_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(metric="train images/sec (fwd + bwd + SGD step)", value=round(batch * steps / dt, 1), unit="images/sec",
                ms_per_step=round(dt / steps * 1e3, 2), steps=steps, warmup=warmup,
                config=dict(workload="configs/yolov3.yaml (20 classes, FREEZE_AT 2) 416x416 batch 64 bf16, synthetic targets",
                            conv_tflops=round(187.357e9 * batch * steps / dt / 1e12, 1)),
                finite_losses=bool(all(torch.isfinite(v) for v in losses.values())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-throughput measurement")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU is visible and there is no CPU fallback")
    world, rank, local = _dist_setup(args.gpus)
    import workload

    model, cfg = workload.calibrated_model(size=SIZE, calib_batch=4, config="yolov3-aug.yaml", dtype="bf16",
                                           head_dtype="f32", device=f"cuda:{torch.cuda.current_device()}")
    eng = model.backbone.engine()
    # every rank gets its own shard of images (different seeds), already resident in HBM
    imgs = workload.synthetic_images(BATCH, SIZE, SIZE, seed=1 + rank, device=model.device)
    mean, std = cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD
    thr, nms_thr = cfg.MODEL.YOLOV3.TEST_SCORE_THRESH, cfg.MODEL.YOLOV3.TEST_NMS_THRESH

    def run(k):
        """k complete steps in the engine's throughput mode: step i+1's convolutions are queued before the host
        reads step i's candidate count; every step's NMS + gather is enqueued before this returns."""
        out = None
        for out in eng.detect_pipelined((imgs for _ in range(k)), score_thr=thr, nms_thr=nms_thr, mean=mean,
                                        std=std, max_det=100):
            pass
        return out

    for slot in (0, 1):  # buffers of both pipeline slots exist before anything is timed
        eng.plan(BATCH, SIZE, SIZE, slot)
    det = n_det = None
    if args.warmup:
        det, n_det = run(args.warmup)
    _barrier(world)
    t0 = time.perf_counter()
    det, n_det = run(args.steps)
    _barrier(world)
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=model.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    plan = eng.plan(BATCH, SIZE, SIZE)
    out = None
    if rank == 0:
        convs = dict(model.backbone.darknet_modules())
        roof = _dominant_kernel_roofline(eng, plan, convs, imgs, mean, std)
        n_cand = int(plan.cand.img_offsets[BATCH].item())
        out = {
            "metric": "images/sec YOLOv3-416 inference incl. decode+NMS",
            "value": round(world * BATCH * args.steps / elapsed, 1),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": "YOLOv3-416 (Darknet-53 + FPN, 80 classes) batch-64 bf16 inference on one MI355X: "
                            "preprocess + backbone + neck + heads + decode + batched NMS + top-100",
                "batch_per_gpu": BATCH, "global_batch": BATCH * world, "input": [SIZE, SIZE],
                "score_thresh": thr, "nms_thresh": nms_thr, "parallelism": f"image-sharded x{world}, no collective",
                "weights": "random init (reference initialisers), BatchNorm statistics calibrated on the synthetic images",
                "candidates_per_batch": n_cand, "detections_per_batch": int(n_det.sum().item()),
                "conv_gflop_per_image": round(plan.flops / BATCH / 1e9, 3),
                "conv_tflops_end_to_end": round(plan.flops * world * args.steps / elapsed / 1e12, 1),
            },
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = _cpu_baseline(model, cfg)
        if world == 1 and not args.no_train:
            del det, n_det
            eng._plans.clear()
            torch.cuda.empty_cache()
            try:
                out["train"] = _train_throughput()
            except Exception as e:  # the headline line must not depend on the secondary measurement
                out["train"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
