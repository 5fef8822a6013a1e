# Interleaved small-batch latency of several library builds on one box.
# usage: [BATCHES=1,4] bash tools/ab_latency_libs.sh <lib.so>...
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  for lib in "$@"; do
    echo -n "$lib: "
    YOLO_HIP_LIB=$lib timeout -k 10 200 python3 tools/latency_bench.py 100 ${BATCHES:-1,4} 2>/dev/null | cut -c1-60 | tr '\n' ' ' || exit 1
    echo
  done
done
