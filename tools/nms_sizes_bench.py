"""Time of the batched NMS with top-100 over candidates per image (uniform stress boxes, 80 classes), batches of 1 and 32:
where grouping by class with per-segment ordering and a list merge pays against the two-order sort.
usage: [YOLO_HIP_LIB=variant.so] python tools/nms_sizes_bench.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import postproc_bench as B  # noqa: E402

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    for n_img in (1, 32):
        for m in (600, 1000, 2000, 4000, 8000, 22743):
            r = B.nms_case(f"{n_img} x {m}, top-100", [B.stress(m, 100 + i) for i in range(n_img)], reps, 100)
            print(json.dumps({k: r[k] for k in ("case", "us", "kept")}))
