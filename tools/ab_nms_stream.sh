cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  YOLO_NMS_STREAM=1 timeout -k 10 120 python3 bench.py --no-train --no-cpu-baseline > gpurun_out/r2_ab_side_$i.json 2>/dev/null || exit 1
  YOLO_NMS_STREAM=0 timeout -k 10 120 python3 bench.py --no-train --no-cpu-baseline > gpurun_out/r2_ab_main_$i.json 2>/dev/null || exit 1
done
timeout -k 10 200 python3 tools/postproc_bench.py > gpurun_out/r2_postproc_final.txt 2>&1 || exit 1
