# Interleaved A/B of the training step on one GPU box: an environment switch off / on, three rounds each.
# usage (GPU box, repo root): bash tools/train_ab.sh YOLO_FUSE_BN_BWD_SUMS [out]
var=${1:?environment variable to toggle (0 / 1)}
out=${2:-gpurun_out/train_ab_$var.txt}
cd "$GRAFT_REPO_ROOT" || exit 1
: > "$out"
for round in 1 2 3; do
  for v in 0 1; do
    line=$(env $var=$v timeout -k 10 150 python3 tools/train_bench.py --batch 64 --size 416 --steps 20 --warmup 4 2>/dev/null | tail -1) || exit 1
    echo "$var=$v round $round: $line" >> "$out"
  done
done
python3 - "$out" "$var" <<'PY'
import json, re, statistics, sys
rows = {0: [], 1: []}
for l in open(sys.argv[1]):
    m = re.match(r"\S+=(\d) round \d: (\{.*\})", l)
    if m:
        rows[int(m.group(1))].append(json.loads(m.group(2)))
with open(sys.argv[1], "a") as f:
    for v in (0, 1):
        ms = [r["ms_per_step"] for r in rows[v]]
        f.write(f"{sys.argv[2]}={v}: ms per step {ms}, median {statistics.median(ms):.2f}, images/s {[r['value'] for r in rows[v]]}\n")
print(open(sys.argv[1]).read())
PY
