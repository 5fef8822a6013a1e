"""Headline benchmark: images/sec of YOLOv3-416 inference including head decode + NMS on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment, this process starts N fresh rank processes itself (one per GPU,
RCCL) BEFORE it touches the GPU, relays rank 0's output and exits with the worst exit code -- the way the
reference's ``launch(main, num_gpus, ...)`` starts its ranks (tools/train_net.py:140-147).  Started under
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (WORLD_SIZE set) it is one rank.

One "step" = one pass of the hot path (fused stem -> Darknet-53 -> FPN -> heads -> decode -> batched NMS
-> top-100 gather) over one resident batch of 64 synthetic uint8 images per GPU.  Steps run in the engine's
throughput mode; all timed steps, including the last one's NMS, finish inside the timed region.  Images shard across
GPUs with no data-path collective (weak scaling); the only collectives are the barriers and the MAX over ranks of
the timed region.  K steps at 5 ms are too short a region to compare boxes (box-to-box spread is a few per cent), so
the K steps are repeated back to back until the region is at least ``--min-seconds`` long and the time is divided by
the number of steps actually run (``timed_steps``).  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 64
SIZE = 416
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak for bf16 on MI355X
PEAK_HBM_GBS = 8000.0


# ------------------------------------------------------------------------------------------------------------
# launcher: N ranks from one command
# ------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv) -> int:
    """Start n rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and
    wait for them.  The parent never initialises the GPU.  Returns the worst exit code."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(_free_port())
    env["WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e))
    worst = 0
    deadline = time.time() + float(os.environ.get("YOLO_BENCH_RANK_TIMEOUT", "900"))
    alive = list(procs)
    while alive:
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in alive:  # a rank died: the others would wait for it in a collective for ever
                    q.terminate()
        if alive and time.time() > deadline:
            worst = worst or 124
            for q in alive:
                q.kill()
        time.sleep(0.05)
    return worst


# ------------------------------------------------------------------------------------------------------------
# the measured workload
# ------------------------------------------------------------------------------------------------------------
class Yolov3Inference:
    """Batch-64 YOLOv3-416 bf16 inference on this rank's GPU.  (Any object with these methods can be benchmarked:
    YOLO_BENCH_WORKLOAD=module:factory replaces it -- the CPU tests of the launcher use a stub.)"""

    metric = "images/sec YOLOv3-416 inference incl. decode+NMS"
    unit = "images/sec"
    dtype = "bf16"
    data = "synthetic"
    units_per_step = BATCH

    def __init__(self, rank: int, world: int, local: int):
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no GPU is visible and there is no CPU fallback")
        backend = os.environ.get("YOLO_BENCH_BACKEND", "nccl")
        if backend != "nccl":  # rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local if world > 1 else 0)
        self.torch = torch
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", torch.cuda.current_device())

    def prepare(self):
        import workload
        torch = self.torch
        self.model, self.cfg = workload.calibrated_model(size=SIZE, calib_batch=4, config="yolov3-aug.yaml", dtype="bf16",
                                                         head_dtype="f32", device=str(self.device))
        self.eng = self.model.backbone.engine()
        # every rank gets its own shard of images (different seeds), already resident in HBM
        self.imgs = workload.synthetic_images(BATCH, SIZE, SIZE, seed=1 + self.rank, device=self.model.device)
        cfg = self.cfg
        self.mean, self.std = cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD
        self.thr, self.nms_thr = cfg.MODEL.YOLOV3.TEST_SCORE_THRESH, cfg.MODEL.YOLOV3.TEST_NMS_THRESH
        for slot in (0, 1):  # buffers of both pipeline slots exist before anything is timed
            self.eng.plan(BATCH, SIZE, SIZE, slot)
        self.last = None
        torch.cuda.synchronize()

    def run(self, k: int, lanes: int = 1):
        """k complete steps in the engine's throughput mode; every step's NMS + gather is enqueued before this
        returns.  The headline runs ONE lane (one batch's convolutions on the GPU at a time), so that the per-kernel
        durations behind ``roofline`` are those of kernels that have the chip to themselves."""
        for out in self.eng.detect_pipelined((self.imgs for _ in range(k)), score_thr=self.thr, nms_thr=self.nms_thr,
                                             mean=self.mean, std=self.std, max_det=100, lanes=lanes):
            self.last = out

    def two_lanes(self, k: int = 100) -> dict:
        """The same steps with consecutive batches alternating between two main streams (Yolov3Engine.detect_pipelined,
        ``lanes=2``): what this GPU delivers when two batches share it.  Secondary: not the headline."""
        torch = self.torch
        self.eng.plan(BATCH, SIZE, SIZE, 2)
        self.run(6, lanes=2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.run(k, lanes=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"value": round(BATCH * k / dt, 1), "unit": "images/sec on this GPU", "steps": k,
                "ms_per_step": round(dt / k * 1e3, 3)}

    def sync(self):
        self.torch.cuda.synchronize()

    def reduce_device(self):
        return self.device

    def device_label(self) -> str:
        """Which physical GPU this rank ran on (ordinal, PCI address, name): two ranks on one card show up as equal
        labels in the line's per_rank_device."""
        i = self.device.index
        pr = self.torch.cuda.get_device_properties(i)
        pci = "%04x:%02x:%02x" % tuple(getattr(pr, k, 0) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        return f"cuda:{i} {pci} {pr.name}"

    # ---- reporting (rank 0) ----
    def config(self, world: int, steps_total: int, elapsed: float) -> dict:
        plan = self.eng.plan(BATCH, SIZE, SIZE)
        det, n_det = self.last
        return {
            "workload": "YOLOv3-416 (Darknet-53 + FPN, 80 classes) batch-64 bf16 inference on one MI355X: "
                        "preprocess + backbone + neck + heads + decode + batched NMS + top-100",
            "batch_per_gpu": BATCH, "global_batch": BATCH * world, "input": [SIZE, SIZE],
            "score_thresh": self.thr, "nms_thresh": self.nms_thr, "parallelism": f"image-sharded x{world}, no collective",
            "weights": "random init (reference initialisers), BatchNorm statistics calibrated on the synthetic images; "
                       "the three head convolutions rescaled so that their logits have unit standard deviation on those "
                       "images and the objectness bias set to -4.5 (that fixes the post-processing load: see "
                       "candidates_per_batch / detections_per_batch)",
            "candidates_per_batch": int(plan.cand.img_offsets[BATCH].item()),
            "detections_per_batch": int(n_det.sum().item()),
            "conv_gflop_per_image": round(plan.flops / BATCH / 1e9, 3),
            "conv_tflops_end_to_end": round(plan.flops * world * steps_total / elapsed / 1e12, 1),
        }

    def roofline(self) -> dict:
        plan = self.eng.plan(BATCH, SIZE, SIZE)
        convs = dict(self.model.backbone.darknet_modules())
        return _dominant_kernel_roofline(self.eng, plan, convs, self.run)

    def postproc(self) -> dict:
        return _postproc_in_step(self)

    def cpu_baseline(self) -> dict:
        return _cpu_baseline(self.model, self.cfg)

    def train(self) -> dict:
        """Collective when world > 1: every rank calls it (data-parallel training step, gradients exchanged over
        the process group); rank 0's record is the one printed."""
        self.last = None
        self.eng._plans.clear()
        self.torch.cuda.empty_cache()
        try:
            return _train_throughput(rank=self.rank, world=self.world)
        except Exception as e:  # the headline line must not depend on the secondary measurement
            if self.world > 1:
                raise  # a rank that left the collective would leave its peers waiting: fail the job instead
            return {"error": f"{type(e).__name__}: {e}"[:200]}


def _dominant_kernel_roofline(eng, plan, convs, step_fn, in_step_steps: int = 12, passes: int = 3):
    """The convolutions run as instances of one implicit-GEMM kernel that differ in their workgroup tile
    (yolo_conv_kernel_name says which launch uses which).  Every launch of the instances with 128 or more channels
    per tile is timed where it runs IN THE BENCHMARK'S OWN STEPS: ``step_fn(n)`` runs n pipelined steps while the engine
    records a HIP event pair (on the launch stream) around each convolution, so every launch sees the cache state and
    the co-running post-processing kernels it sees in the timed region; the cost of an empty event pair is measured
    and subtracted.  The instance with the most time is the dominant kernel: achieved = its algorithmic FLOPs
    (2 * pixels * Cout * Cin * k * k per launch) / its time.  The other instances are reported next to it."""
    import torch
    from yolo_amd import ops
    stream = torch.cuda.current_stream()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    timed = []  # (step index, kernel name, flops, bytes, filter size)
    for i, (meta, step) in enumerate(zip(plan.meta, plan.steps)):
        c = convs[meta["name"]]
        if c.in_channels % 64 != 0:
            continue
        up = 2 if meta["up"] else 1
        M = plan.N * (meta["y"].H // up) * (meta["y"].W // up)
        name = ops.conv_kernel_name(meta["x"], meta["y"], c.in_channels, c.out_channels, c.kernel_size, c.stride,
                                    residual=meta["residual"] is not None, upsample2x=bool(meta["up"]))
        if ",128," not in name and ",256," not in name:
            continue
        es = meta["x"].buf.element_size()
        by = (plan.N * meta["x"].H * meta["x"].W * c.in_channels * es
              + c.out_channels * c.in_channels * c.kernel_size ** 2 * es
              + plan.N * meta["y"].H * meta["y"].W * c.out_channels * meta["y"].buf.element_size()
              + (M * c.out_channels * es if meta["residual"] is not None else 0))
        timed.append((i, name, 2.0 * M * c.out_channels * c.in_channels * c.kernel_size ** 2, by, c.kernel_size))
    which = {i: k for k, (i, _, _, _, _) in enumerate(timed)}
    # (1) inside real pipelined steps: the engine records an event pair around every convolution launch on the stream
    #     it runs on, while decode / NMS of the previous batch run on their streams -- what a profiler of the bench sees
    samples = [[] for _ in timed]
    eng.trace_events = []
    step_fn(in_step_steps)
    torch.cuda.synchronize()
    events, eng.trace_events = eng.trace_events, None
    for label, a, b in events:
        if isinstance(label, tuple) and label[0] == "conv" and label[1] in which:
            samples[which[label[1]]].append(a.elapsed_time(b))
    # (2) the cost of an empty event pair (median of many, recorded back to back behind a kernel so that the stream
    #     is busy as in a step), subtracted from every timed launch
    pairs = []
    busy = torch.zeros(1 << 20, device=eng.device)
    for _ in range(16 * passes):
        busy.add_(1.0)   # (a torch kernel, not one of the library's: it must not enter the profiler's per-kernel averages)
        e0, e1 = ev(), ev()
        e0.record(stream)
        e1.record(stream)
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    empty = statistics.median(a.elapsed_time(b) for a, b in pairs)
    ms = [statistics.median(v) if v else 0.0 for v in samples]   # per launch: median over the traced steps
    acc, by_k = {}, {}
    for (i, name, fl, by, ks), t in zip(timed, ms):
        for a in (acc.setdefault(name, [0.0, 0.0, 0, 0.0]), by_k.setdefault((name, ks), [0.0, 0.0, 0, 0.0])):
            a[0] += fl
            a[1] += max(t - empty, 1e-6)
            a[2] += 1
            a[3] += by
    dom = max(acc, key=lambda n: acc[n][1])
    flops, t_ms, launches, by = acc[dom]
    achieved = flops / (t_ms * 1e-3) / 1e12
    out = dict(bound="mfma", achieved=round(achieved, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
               frac=round(achieved / PEAK_BF16_TFLOPS, 4), traffic=None,
               kernel=dom, launches_per_step=launches,
               avg_launch_us=round(t_ms * 1e3 / launches, 2), flop_per_launch=round(flops / launches / 1e9, 3),
               algorithmic_bytes_per_launch=round(by / launches), event_pair_overhead_us=round(empty * 1e3, 2))
    # HBM bytes per launch come from separate rocprofv3 --pmc passes of this command (tools/pmc_summary.py writes the
    # file together with the git commit of the tree it profiled); a record of another kernel state is not replayed
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        with open(pmc) as f:
            rec = json.load(f)
        hit = rec.get("kernels", {}).get(dom)
        if hit and rec.get("kernel_source_sha") == _kernel_source_sha():
            out["traffic"] = hit["hbm_bytes_per_launch"]
            out["traffic_source"] = "profiles/pmc_traffic.json: " + rec.get("source", "")
        else:
            out["traffic_note"] = ("profiles/pmc_traffic.json was collected on other kernel sources (or holds no row "
                                   "for this kernel): not replayed")
    # the same instance serves long-K (3x3, matrix-bound) and short-K (1x1, memory-bound) layers: both parts of its time
    out["by_filter_size"] = {
        f"{ks}x{ks}": dict(achieved=round(a[0] / (a[1] * 1e-3) / 1e12, 1), launches_per_step=a[2],
                           avg_launch_us=round(a[1] * 1e3 / a[2], 2),
                           algorithmic_gbytes_per_s=round(a[3] / (a[1] * 1e-3) / 1e9))
        for (name, ks), a in sorted(by_k.items()) if name == dom}
    out["other_instances"] = {
        name: dict(achieved=round(a[0] / (a[1] * 1e-3) / 1e12, 1), launches_per_step=a[2],
                   avg_launch_us=round(a[1] * 1e3 / a[2], 2))
        for name, a in sorted(acc.items()) if name != dom}
    tot_f = sum(a[0] for a in acc.values())
    tot_t = sum(a[1] for a in acc.values())
    out["all_timed_instances"] = dict(achieved=round(tot_f / (tot_t * 1e-3) / 1e12, 1), launches_per_step=len(timed),
                                      frac=round(tot_f / (tot_t * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4))
    return out


def _kernel_source_sha() -> str:
    """Identifies the kernel sources a PMC record belongs to: hash of the convolution kernel's source
    (yolo_amd/csrc/conv.hip) and the headers next to it."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(ROOT, "yolo_amd", "csrc", "conv.hip")) +
                       glob.glob(os.path.join(ROOT, "yolo_amd", "csrc", "*.h"))):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _postproc_source_sha() -> str:
    """Identifies the decode / NMS kernel sources the post-processing rows of a PMC record belong to."""
    import hashlib
    h = hashlib.sha1()
    for name in ("decode.hip", "nms.hip", "common.h"):
        with open(os.path.join(ROOT, "yolo_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _postproc_in_step(w: "Yolov3Inference", steps: int = 6) -> dict:
    """Decode and NMS + gather timed INSIDE pipelined steps (HIP events on the streams they run on, next to the
    following batch's convolutions), against the HBM roofline on two byte counts: the algorithmic bytes (every head
    value once + the candidate arrays) and the bytes the kernels cannot avoid touching (one 64-byte sector per anchor
    row for the objectness test + the whole rows of cells that yield candidates + the candidate arrays)."""
    import torch
    eng = w.eng
    eng.trace_events = []
    w.run(steps)
    torch.cuda.synchronize()
    ev, eng.trace_events = eng.trace_events, None
    acc = {}
    for label, a, b in ev:
        acc.setdefault(label, []).append(a.elapsed_time(b) * 1e3)
    plan = eng.plan(BATCH, SIZE, SIZE)
    n_cand = int(plan.cand.img_offsets[BATCH].item())
    n_det = int(w.last[1].sum().item())
    rows = sum(h.H * h.W for h in plan.heads) * eng.num_anchors * BATCH
    K = 5 + eng.num_classes
    es = plan.heads[0].buf.element_size()
    alg_dec = rows * K * es + n_cand * 28
    touched_dec = rows * 64 + min(n_cand, rows) * K * es + n_cand * 28
    alg_nms = n_cand * 24 + 8 * n_det + 2 * n_cand * 8
    out = {}

    def record(label, key, alg, touched, events, where):
        ts = events.get(label)
        if not ts:
            return
        t = statistics.median(ts)
        out[key] = dict(in_step_us=round(t, 1), stream=where, algorithmic_bytes=int(alg),
                        touched_bytes_lower_bound=int(touched), algorithmic_gbytes_per_s=round(alg / t / 1e3, 1),
                        frac_hbm_algorithmic=round(alg / (t * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                        frac_hbm_touched=round(touched / (t * 1e-6) / 1e9 / PEAK_HBM_GBS, 4))
        kind = "decode" if key.startswith("decode") else key
        if kind in pmc_bytes:
            out[key]["pmc_hbm_bytes"] = int(pmc_bytes[kind])
            out[key]["frac_hbm_pmc"] = round(pmc_bytes[kind] / (t * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)

    # counter bytes of the same kernels (separate rocprofv3 --pmc passes of this command, tools/pmc_summary.py): what
    # really left L2 per launch, next to the two analytic counts; replayed only for the sources it was taken on
    pmc_bytes = {}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        with open(pmc) as f:
            rec = json.load(f)
        if rec.get("postproc_source_sha") == _postproc_source_sha():
            for name, row in rec.get("postproc_kernels", {}).items():
                key = "decode" if name.startswith("decode_") else "nms_gather"
                pmc_bytes[key] = pmc_bytes.get(key, 0) + row["hbm_bytes_per_launch"]
    side = "its own stream, next to the following batch's convolutions (the default: hidden, but it shares the chip)"
    record("decode", "decode", alg_dec, touched_dec, acc, side if eng.decode_on_side_stream else "main stream")
    record("nms_gather", "nms_gather", alg_nms, alg_nms, acc, side if eng.nms_on_side_stream else "main stream")
    # the same steps with the decode launches on the main stream, directly behind the last head convolution and with
    # nothing else on the chip: what the three decode kernels do when they have the GPU (costs 0.7 % images/s: they are
    # then no longer hidden behind the next batch)
    if eng.decode_on_side_stream:
        eng.decode_on_side_stream = False
        try:
            eng.trace_events = []
            w.run(steps)
            torch.cuda.synchronize()
            ev2, eng.trace_events = eng.trace_events, None
        finally:
            eng.decode_on_side_stream = True
        acc2 = {}
        for label, a, b in ev2:
            acc2.setdefault(label, []).append(a.elapsed_time(b) * 1e3)
        record("decode", "decode_on_main_stream", alg_dec, touched_dec, acc2,
               "main stream, behind the last head convolution, alone on the chip")
    return out


def _cpu_baseline(model, cfg):
    """The CPU oracle (plain torch fp32 restatement of the reference + C NMS) on a bounded sample of the same
    workload, on this box's host cores: batch 64 of the benchmark's configuration and batch 1 through
    configs/yolov3.yaml's shape (20 classes), each at n = the box's CPU share and n = 8 threads; warm-up runs, then
    the median of the timed runs.  ``value`` is the batch-64 rate at n = all."""
    import torch

    import workload
    from oracle import nms_c, yolo_ref as R
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    kw = dict(num_classes=cfg.MODEL.YOLOV3.NUM_CLASSES, score_thr=cfg.MODEL.YOLOV3.TEST_SCORE_THRESH,
              nms_thr=cfg.MODEL.YOLOV3.TEST_NMS_THRESH, nms_fn=nms_c.nms, device_kind="gpu")
    imgs64 = [im.float() for im in workload.synthetic_images(BATCH, SIZE, SIZE, seed=1)]
    # batch 1: the 20-class head of configs/yolov3.yaml = the first 75 of this model's 255 head channels per level
    # would change the arithmetic; the per-image cost of the 80-class model is within 0.3 % of it (head convs are
    # 0.66 % of the FLOPs), so batch 1 runs the same state dict
    rows = {}

    def timed(batch, n, warm, reps):
        torch.set_num_threads(n)
        with torch.no_grad():
            for _ in range(warm):
                R.inference(batch, sd, **kw)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                R.inference(batch, sd, **kw)
                ts.append(time.perf_counter() - t0)
        return len(batch) / statistics.median(ts)

    t_all = time.perf_counter()
    for n in sorted({cores, min(8, cores)}, reverse=True):
        rows[f"batch1_threads{n}"] = round(timed(imgs64[:1], n, 3, 10), 2)
    rows[f"batch64_threads{cores}"] = round(timed(imgs64, cores, 1, 3), 2)
    if cores > 8:
        rows["batch64_threads8"] = round(timed(imgs64, 8, 0, 2), 2)
    spent = time.perf_counter() - t_all
    torch.set_num_threads(cores)
    return dict(value=rows[f"batch64_threads{cores}"], unit="images/sec", cores=cores, kind="port",
                sample=(f"oracle (torch {torch.__version__} fp32 CPU ops + C NMS) on the same 416x416 workload: batch 64 "
                        f"(1 warm-up + median of 3 at {cores} threads; median of 2 at 8 threads) and batch 1 (3 warm-up + "
                        f"median of 10 at each thread count); {spent:.0f} s of CPU work"),
                images_per_sec=rows)


def _train_throughput(steps: int = 12, warmup: int = 4, batch: int = 64, rank: int = 0, world: int = 1):
    """Secondary metric of the same hot path (training images per second): forward + backward + gradient exchange +
    SGD step of configs/yolov3.yaml at 416x416, bf16, synthetic COCO-shaped targets, ``batch`` images per GPU.  With
    world > 1 this is the data-parallel step of the reference's ``launch(main, num_gpus)`` -> DefaultTrainer -> DDP
    (tools/train_net.py:117-147): every rank runs its own images, gradients are exchanged bucket by bucket from
    inside the backward pass (yolo_amd.utils.dist.GradBucketer) over the process group bench.py initialised (RCCL);
    the region is bracketed by a barrier + synchronize and the MAX over ranks is taken, like the headline."""
    import torch

    import workload
    from yolo_amd.trainer import Trainer, synthetic_batches
    from yolo_amd.utils import dist as D
    model, cfg = workload.build_model(config="yolov3.yaml", dtype="bf16", device=f"cuda:{torch.cuda.current_device()}",
                                      opts=["OUTPUT_DIR", "", "SOLVER.CHECKPOINT_PERIOD", 0, "SOLVER.BASE_LR", 1e-4])
    data = next(synthetic_batches(cfg, batch, SIZE, seed=7 + rank, device=model.device))
    tr = Trainer(cfg, model, iter(lambda: data, None), log=lambda m: None)   # world > 1: broadcasts rank 0's weights
    for _ in range(warmup):
        tr.run_step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = tr.run_step()
    D.barrier()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, model.device)
    finite = bool(all(torch.isfinite(v) for v in losses.values()))
    rec = dict(metric="train images/sec (fwd + bwd + gradient exchange + SGD step)",
               value=round(world * batch * steps / dt, 1), unit="images/sec", n_gpus=world,
               ms_per_step=round(dt / steps * 1e3, 2), steps=steps, warmup=warmup,
               config=dict(workload="configs/yolov3.yaml (20 classes, FREEZE_AT 2) 416x416 bf16, synthetic targets, "
                                    f"{batch} images per GPU",
                           batch_per_gpu=batch, global_batch=batch * world,
                           conv_tflops=round(187.357e9 * world * batch * steps / dt / 1e12, 1)),
               finite_losses=finite)
    if world > 1:
        import torch.distributed as dist
        # data-parallel invariant: identical starting weights + averaged gradients = identical weights on every rank
        chk = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
        rec["params_in_sync"] = bool(D.max_over_ranks(chk, model.device) == -D.max_over_ranks(-chk, model.device))
        eng = model.backbone.train_engine()
        bucketers = [P.bucketer for P in eng._plans.values() if getattr(P, "bucketer", None) is not None]
        rec["exchange"] = bucketers[0].exchange if bucketers else None
        rec["buckets_per_step"] = len(bucketers[0].launched) if bucketers else 0
        rec["backend"] = dist.get_backend()
        rec["n_ranks_" + ("rccl" if dist.get_backend() == "nccl" else dist.get_backend())] = dist.get_world_size()
    return rec


# ------------------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------------------
def _make_workload(rank: int, world: int, local: int):
    spec = os.environ.get("YOLO_BENCH_WORKLOAD")
    if spec:
        mod, fn = spec.split(":")
        return getattr(importlib.import_module(mod), fn)(rank, world, local)
    return Yolov3Inference(rank, world, local)


def run_rank(args) -> None:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    w = _make_workload(rank, world, local)
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("YOLO_BENCH_BACKEND", "nccl")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if dist is not None:
            dist.barrier()
        w.sync()

    w.prepare()
    if args.warmup:
        w.run(args.warmup)
    # size of the timed region: the K steps are repeated until the region is at least --min-seconds long; the count
    # is agreed between the ranks (MAX of the estimates) so that every rank runs the same number of steps
    repeats = 1
    if args.min_seconds > 0:
        barrier()
        t0 = time.perf_counter()
        w.run(args.steps)
        w.sync()
        est = time.perf_counter() - t0
        repeats = max(1, int(args.min_seconds / max(est, 1e-6) + 0.999))
        if dist is not None:
            import torch
            t = torch.tensor([repeats], dtype=torch.int64, device=w.reduce_device() if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            repeats = int(t.item())
    barrier()
    t0 = time.perf_counter()
    for _ in range(repeats):
        w.run(args.steps)
    w.sync()
    own = time.perf_counter() - t0  # this rank's own work; the region below also holds the wait for the slowest rank
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [own]
    if dist is not None:
        import torch
        dev = w.reduce_device() if dist.get_backend() == "nccl" else "cpu"
        mine = torch.tensor([own], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        elapsed = float(mine.item())

    label = w.device_label() if hasattr(w, "device_label") else f"rank {rank}"
    labels = [label]
    if dist is not None:
        labels = [None] * world
        dist.all_gather_object(labels, label)

    total_steps = repeats * args.steps
    out = None
    if rank == 0:
        out = {
            "metric": w.metric,
            "value": round(world * w.units_per_step * total_steps / elapsed, 1),
            "unit": w.unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / total_steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": w.dtype,
            "data": w.data,
            "timed_steps": total_steps,
            "timed_seconds": round(elapsed, 3),
            "per_rank_value": [round(w.units_per_step * total_steps / t, 1) for t in per_rank],
            "per_rank_device": labels,
            "ranks_on_distinct_devices": len(set(labels)) == world,
            "config": w.config(world, total_steps, elapsed),
            "roofline": w.roofline(),
        }
        if hasattr(w, "postproc"):
            out["postproc"] = w.postproc()
        if hasattr(w, "two_lanes") and not args.no_two_lanes:
            out["two_lane_throughput"] = w.two_lanes()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = w.cpu_baseline()
        if dist is not None:
            # ranks the collective backend really connected (RCCL when the backend is nccl)
            out["n_ranks_" + ("rccl" if dist.get_backend() == "nccl" else dist.get_backend())] = dist.get_world_size()
    # the secondary (training) measurement is collective when world > 1: every rank runs it (the others wait in its
    # first collective while rank 0 finishes the records above), rank 0 reports it.  A collective that never completes
    # cannot be interrupted from Python, so at world > 1 a watchdog thread bounds the block: when it fires, rank 0 prints
    # the line it has (with the reason in place of the training record) and every rank leaves the process.
    if not args.no_train and hasattr(w, "train"):
        watchdog = None
        if world > 1:
            import threading

            def give_up():
                if out is not None:
                    out["train"] = {"error": f"training block did not finish within {args.train_timeout:.0f} s "
                                             f"(world {world}): abandoned by the watchdog"}
                    print(json.dumps(out), flush=True)
                os._exit(0)   # the headline line is out (rank 0); the reason travels in it

            watchdog = threading.Timer(args.train_timeout, give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            train_rec = w.train()
        except Exception as e:
            if world == 1:
                raise
            # a rank that left the collective leaves its peers waiting: they end through their own watchdogs
            train_rec = {"error": f"{type(e).__name__}: {e}"[:300]}
            if out is not None:
                out["train"] = train_rec
                print(json.dumps(out), flush=True)
            os._exit(0)
        if watchdog is not None:
            watchdog.cancel()
        if out is not None:
            out["train"] = train_rec
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="repeat the K steps until the timed region is at least this long (0: exactly K steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-throughput measurement")
    ap.add_argument("--no-two-lanes", action="store_true",
                    help="skip the secondary two-lane throughput pass (profiling runs: its kernels overlap each other)")
    ap.add_argument("--train-timeout", type=float, default=240.0,
                    help="world > 1: seconds after which the (collective) training block is abandoned")
    argv = sys.argv[1:] if argv is None else list(argv)
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}; using WORLD_SIZE", file=sys.stderr)
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
