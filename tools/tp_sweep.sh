cd "$GRAFT_REPO_ROOT" || exit 1
export YOLO_HIP_LIB=yolo_amd/libyolo_hip_trace.so
: > gpurun_out/r2_tp_sweep.txt
timeout -k 10 100 python3 tools/tp_sweep.py 2>/dev/null | tail -1 >> gpurun_out/r2_tp_sweep.txt || exit 1
for tp in 64 96 128 192; do
  YOLO_TP=$tp timeout -k 10 100 python3 tools/tp_sweep.py 2>/dev/null | tail -1 >> gpurun_out/r2_tp_sweep.txt || exit 1
done
