# A/B on one box: head decode on its own stream (next to the following batch's convolutions) or on the main stream
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do
  for v in 1 0; do
    YOLO_DECODE_STREAM=$v timeout -k 10 120 python3 bench.py --no-train --no-cpu-baseline > gpurun_out/ab_dec${v}_$i.json 2>/dev/null || exit 1
  done
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_dec*.json')):
    d=json.load(open(f)); print(f, d['value'], d['postproc']['decode'], d['postproc']['nms_gather']['in_step_us'])
P
