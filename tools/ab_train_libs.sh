# Interleaved training bench of two library builds on one box.  usage: bash tools/ab_train_libs.sh <libA.so> <libB.so>
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do
  for lib in "$1" "$2"; do
    echo -n "$lib: "
    YOLO_HIP_LIB=$lib timeout -k 10 200 python3 tools/train_bench.py --batch 64 --size 416 --steps 12 --warmup 3 | cut -c60-160 || exit 1
  done
done
