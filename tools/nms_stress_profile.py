"""The NMS stress configuration alone (32 images x 22 743 candidates, 80 classes, top-100), for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 tools/nms_stress_profile.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import postproc_bench as B  # noqa: E402

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    print(B.nms_case("stress uniform 32 x 22743, top-100", [B.stress(22743, 100 + i) for i in range(32)], reps, 100))
