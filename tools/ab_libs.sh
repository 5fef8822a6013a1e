# Interleaved bench of two library builds on one box.  usage: bash tools/ab_libs.sh <libA.so> <libB.so>
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  YOLO_HIP_LIB=$1 timeout -k 10 120 python3 bench.py --no-train --no-cpu-baseline > gpurun_out/ab_A_$i.json 2>/dev/null || exit 1
  YOLO_HIP_LIB=$2 timeout -k 10 120 python3 bench.py --no-train --no-cpu-baseline > gpurun_out/ab_B_$i.json 2>/dev/null || exit 1
done
