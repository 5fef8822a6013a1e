"""Throughput of the device-side augmentation against the HBM roofline.

Workload = the training input of configs/yolov3-aug.yaml: every output is a 1280 x 1280 mosaic of four source images
(480 x 640 BGR uint8 here), each resized so that its shorter side is 640 (Pillow's bilinear, 640 x 853), then warped
(random affine), colour-jittered, flipped and pasted -- 64 outputs = 256 source images per batch.  Images are already
in HBM when the timed region starts (the upload is reported separately).

Per stage: time per batch, algorithmic bytes (each input and output byte once), GB/s and the fraction of 8 TB/s.
CPU side: Pillow's resize (the real thing) and the numpy oracle for warp + jitter, one core, on a few images.
usage: python tools/aug_bench.py [batch_outputs] [reps]
"""
import json
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from yolo_amd import ops  # noqa: E402
from yolo_amd.config import get_cfg  # noqa: E402
from yolo_amd.data import YOLOV3DatasetMapper  # noqa: E402
from yolo_amd.data import transforms as T  # noqa: E402
from yolo_amd.data.augmentation_impl import RandomAffine, RandomColorJitter  # noqa: E402

HBM_PEAK = 8.0e12


def events(fn, reps):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def row(name, t, by, n_img):
    return dict(stage=name, ms=round(t * 1e3, 3), algorithmic_MB=round(by / 1e6, 1), GBps=round(by / t / 1e9, 1),
                hbm_frac=round(by / t / HBM_PEAK, 4), source_images_per_s=round(n_img / t))


def main():
    n_out = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda")
    n_src = 4 * n_out
    sh, sw, rh, rw, mos = 480, 640, 640, 853, 640
    rng = np.random.default_rng(0)
    host = [rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8) for _ in range(8)]
    src = [torch.from_numpy(host[i % 8]).to(dev) for i in range(n_src)]
    resized = [torch.empty((rh, rw, 3), dtype=torch.uint8, device=dev) for _ in range(n_src)]
    ws = torch.empty(sh * rw * 3, dtype=torch.uint8, device=dev)
    canvas = torch.empty((n_out, 3, 2 * mos, 2 * mos), dtype=torch.uint8, device=dev)
    ops.resample_tables(sw, rw, dev), ops.resample_tables(sh, rh, dev)

    def do_resize():
        ops.aug_resample_batch(src, [(rh, rw)] * n_src, resized)

    np.random.seed(0)
    random.seed(0)
    aff, jit = RandomAffine(0.1, 0.1, 0, 10.0, 0.0), RandomColorJitter("RGB", 0.1, 1.5, 1.5)
    jobs, pix = [], 0
    shape = np.empty((rh, rw, 3), np.uint8)
    for o in range(n_out):
        yc, xc = [int(np.random.uniform(-x, s * 2 + x)) for x, s in zip((-mos // 2, -mos // 2), [mos, mos])]
        for i in range(4):
            tail = [aff.get_transform(shape), jit.get_transform(shape)] + ([T.HFlipTransform(rw)] if np.random.uniform() < 0.5 else [])
            h, w = rh, rw
            if i == 0:
                x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
                x1b, y1b = w - (x2a - x1a), h - (y2a - y1a)
            elif i == 1:
                x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, mos * 2), yc
                x1b, y1b = 0, h - (y2a - y1a)
            elif i == 2:
                x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(mos * 2, yc + h)
                x1b, y1b = w - (x2a - x1a), 0
            else:
                x1a, y1a, x2a, y2a = xc, yc, min(xc + w, mos * 2), min(mos * 2, yc + h)
                x1b, y1b = 0, 0
            cw, ch = x2a - x1a, y2a - y1a
            jobs.append(T.fused_job(tail, resized[4 * o + i], canvas[o], dst_xy=(x1a, y1a), crop=(x1b, y1b, cw, ch), planar=True))
            pix += cw * ch

    def do_fill():
        canvas.fill_(114)

    def do_apply():
        ops.aug_apply(jobs)

    def do_all():
        do_resize()
        do_fill()
        do_apply()

    rows = []
    t_r = events(do_resize, reps)
    rows.append(row("resize 480x640 -> 640x853 (two passes)", t_r, n_src * 3 * (sh * sw + 2 * sh * rw + rh * rw), n_src))
    t_f = events(do_fill, reps)
    rows.append(row("canvas fill 114", t_f, canvas.numel(), n_src))
    t_a = events(do_apply, reps)
    rows.append(row("fused warp + jitter + flip + paste", t_a, 2 * 3 * pix, n_src))
    t_all = events(do_all, reps)
    by_all = n_src * 3 * (sh * sw + 2 * sh * rw + rh * rw) + canvas.numel() + 2 * 3 * pix
    r = row("whole batch (images resident in HBM)", t_all, by_all, n_src)
    r["mosaic_outputs_per_s"] = round(n_out / t_all)
    rows.append(r)
    pinned = [torch.from_numpy(h).pin_memory() for h in host]
    t0 = time.perf_counter()
    for _ in range(3):
        up = [pinned[i % 8].to(dev, non_blocking=True) for i in range(n_src)]
        torch.cuda.synchronize()
    t_up = (time.perf_counter() - t0) / 3
    rows.append(dict(stage="upload of the source images (pinned host -> HBM)", ms=round(t_up * 1e3, 3),
                     GBps=round(n_src * sh * sw * 3 / t_up / 1e9, 1)))
    # the mapper itself (host planning + launches), synthetic in-memory dataset
    cfg = get_cfg()
    cfg.merge_from_file("configs/yolov3-aug.yaml")
    mapper = YOLOV3DatasetMapper(cfg, True)
    data = [dict(image=host[i % 8], annotations=[dict(bbox=[50.0, 60.0, 300.0, 400.0], bbox_mode=0, category_id=1)]) for i in range(n_out + 8)]
    mapper.map_batch(data[:8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mapper.map_batch(data[8:])
    torch.cuda.synchronize()
    t_m = time.perf_counter() - t0
    rows.append(dict(stage="YOLOV3DatasetMapper.map_batch incl. host planning and upload", ms=round(t_m * 1e3, 1),
                     mosaic_outputs_per_s=round(n_out / t_m)))
    cached = YOLOV3DatasetMapper(cfg, True, cache_bytes=1 << 30)
    for i, d in enumerate(data):
        d["image_id"] = i % 8
    cached.map_batch(data[:8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cached.map_batch(data[8:])
    torch.cuda.synchronize()
    t_c = time.perf_counter() - t0
    rows.append(dict(stage="YOLOV3DatasetMapper.map_batch, sources resident in HBM (cache_bytes)", ms=round(t_c * 1e3, 1),
                     mosaic_outputs_per_s=round(n_out / t_c)))
    # CPU: one core
    from PIL import Image
    from oracle import aug_ref as A
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    small = [np.asarray(Image.fromarray(host[i]).resize((rw, rh), Image.BILINEAR)) for i in range(8)]
    t_pil = (time.perf_counter() - t0) / 8
    t0 = time.perf_counter()
    n_cpu = 3
    for i in range(n_cpu):
        tf = aff.get_transform(shape)
        A.color_jitter(A.warp_affine(small[i], tf.M), 0.95, 1.3, 0.8, True)
    t_or = (time.perf_counter() - t0) / n_cpu
    rows.append(dict(stage="CPU, 1 core: Pillow resize (library) + numpy oracle warp + jitter, per source image",
                     ms=round((t_pil + t_or) * 1e3, 1), pillow_resize_ms=round(t_pil * 1e3, 2),
                     source_images_per_s=round(1.0 / (t_pil + t_or), 1)))
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
