"""Decode and NMS throughput against the HBM roofline.

  * NMS stress (YOLOv3-608: 32 images x 22743 candidates, 80 classes, IoU 0.5; uniform and clustered boxes):
    time of yolo_batched_nms (+ top-100), algorithmic bytes M*(16+4+4) + 8*n_keep + 2*M*8, pair-IoU rate;
  * decode of the 416 batch-64 head tensors (10647 cells x 255 channels per image).
usage: python tools/postproc_bench.py [reps]
"""
import json
import math
import sys

import torch

sys.path.insert(0, ".")
from yolo_amd import ops  # noqa: E402

HBM_PEAK = 8.0e12


def stress(n, seed, side=608.0, num_classes=80):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n, 2, generator=g) * side
    wh = torch.exp(torch.rand(n, 2, generator=g) * (math.log(400.0) - math.log(8.0)) + math.log(8.0))
    return (torch.cat([c - wh / 2, c + wh / 2], 1).float(), (0.001 + torch.rand(n, generator=g) * 0.999).float(),
            torch.randint(0, num_classes, (n,), generator=g))


def clustered(n_clusters, per, seed, side=608.0, num_classes=80):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n_clusters, 2, generator=g) * side
    wh = 20 + torch.rand(n_clusters, 2, generator=g) * 200
    cls = torch.randint(0, num_classes, (n_clusters,), generator=g)
    cc = c.repeat_interleave(per, 0) + torch.randn(n_clusters * per, 2, generator=g) * 6
    ww = wh.repeat_interleave(per, 0) * (1 + 0.1 * torch.randn(n_clusters * per, 2, generator=g))
    return torch.cat([cc - ww / 2, cc + ww / 2], 1).float(), torch.rand(n_clusters * per, generator=g).float(), cls.repeat_interleave(per, 0)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def nms_case(name, per_image, reps, top_k):
    dev = torch.device("cuda")
    boxes = torch.cat([p[0] for p in per_image]).to(dev)
    scores = torch.cat([p[1] for p in per_image]).to(dev)
    classes = torch.cat([p[2] for p in per_image]).to(torch.int32).to(dev)
    counts = [p[0].shape[0] for p in per_image]
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=dev)
    N, M = len(per_image), int(sum(counts))
    ws = torch.empty(ops.nms_workspace_bytes(N, M), dtype=torch.uint8, device=dev)
    stride = top_k if top_k else max(counts)
    keep = torch.empty((N, stride), dtype=torch.int32, device=dev)
    nk = torch.empty((N,), dtype=torch.int32, device=dev)
    run = lambda: ops.batched_nms(boxes, scores, classes, off, M, 0.5, num_classes=80, mode=ops.NMS_PER_CLASS,
                                  top_k=top_k, keep_stride=stride, workspace=ws, keep=keep, n_keep=nk)
    t = timeit(run, reps)
    kept = int(nk.sum())
    by = M * 24 + 8 * kept + 2 * M * 8
    pairs = 0
    for p in per_image:
        h = torch.bincount(p[2], minlength=80).double()
        pairs += float((h * (h - 1) / 2).sum())
    return dict(case=name, images=N, candidates=M, kept=kept, top_k=top_k, us=round(t * 1e6, 1),
                algorithmic_MB=round(by / 1e6, 2), GBps=round(by / t / 1e9, 1), hbm_frac=round(by / t / HBM_PEAK, 4),
                segment_pairs=int(pairs), Gpairs_per_s=round(pairs / t / 1e9, 2), boxes_per_s=round(M / t / 1e6, 1))


def decode_case(reps, dtype):
    dev = torch.device("cuda")
    N, C, A = 64, 80, 3
    g = torch.Generator().manual_seed(0)
    heads = []
    for G in (52, 26, 13):
        v = ops.alloc_view(N, G, G, 256, 0, dtype, dev)
        t = torch.randn(N, G, G, 256, generator=g)
        t[..., 4::85] -= 4.5
        v.interior().copy_(t.to(dev))
        heads.append(ops.View(v.buf, G, G, 0, 255, 0))
    anchors = [[(10, 13), (16, 30), (33, 23)], [(30, 61), (62, 45), (59, 119)], [(116, 90), (156, 198), (373, 326)]]
    cap = N * 4096
    out = ops.Candidates(torch.empty((cap, 4), device=dev), torch.empty(cap, device=dev),
                         torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(N + 1, dtype=torch.int32, device=dev))
    ws = torch.empty(ops.decode_workspace_bytes(heads), dtype=torch.uint8, device=dev)
    run = lambda: ops.decode(heads, (8, 16, 32), anchors, C, 0.25, cap, out=out, workspace=ws)
    t = timeit(run, reps)
    M = int(out.img_offsets[N])
    es = 2 if dtype == torch.bfloat16 else 4
    # algorithmic bytes: every head element once (10647 anchors x (5 + C) values per image) + 28 bytes per candidate.
    # The kernel reads less than that: class scores of anchors whose objectness fails the threshold are skipped.
    by = N * 10647 * (5 + C) * es + M * 28
    return dict(case=f"decode 416 batch 64 heads {dtype}".replace("torch.", ""), candidates=M, us=round(t * 1e6, 1),
                algorithmic_MB=round(by / 1e6, 1), GBps=round(by / t / 1e9, 1), hbm_frac=round(by / t / HBM_PEAK, 4))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rows = [nms_case("stress uniform 32 x 22743", [stress(22743, 100 + i) for i in range(32)], reps, 0),
            nms_case("stress uniform 32 x 22743, top-100", [stress(22743, 100 + i) for i in range(32)], reps, 100),
            nms_case("stress clustered 32 x (300 x 76)", [clustered(300, 76, 200 + i) for i in range(32)], reps, 0),
            nms_case("benchmark-like 64 x 200, top-100", [stress(200, 300 + i, side=416.0) for i in range(64)], reps, 100),
            decode_case(reps, torch.float32), decode_case(reps, torch.bfloat16)]
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
