# Tile sweep: every distinct convolution shape of the benchmark with the library's own pixel-tile choice and with the
# tile pinned (trace build), one JSON line per setting, then a table.  usage (on the GPU box): bash tools/tp_sweep.sh [out]
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=${1:-gpurun_out/r3_tp_sweep.txt}
export YOLO_HIP_LIB=yolo_amd/libyolo_hip_trace.so
: > "$OUT.json"
timeout -k 10 100 python3 tools/tp_sweep.py 2>/dev/null | tail -1 >> "$OUT.json" || exit 1
for tp in 64 96 128 176 192; do
  YOLO_TP=$tp timeout -k 10 100 python3 tools/tp_sweep.py 2>/dev/null | tail -1 >> "$OUT.json" || exit 1
done
python3 - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1] + ".json")]
tps = [r["tp"] for r in rows]
with open(sys.argv[1], "w") as f:
    f.write("tools/tp_sweep.sh: time per launch (us, batch 64, bf16, trace build) with the library's own tile choice (auto) "
            "and with the pixel tile pinned (a pinned 176 only applies where that tile is offered; strip layers ignore the pin)\n")
    f.write(f"{'shape':28s}" + "".join(f"{t:>9s}" for t in tps) + "   best pinned\n")
    for k in rows[0]["us"]:
        v = [r["us"][k] for r in rows]
        b = min(range(1, len(v)), key=lambda i: v[i])
        f.write(f"{k:28s}" + "".join(f"{x:9.1f}" for x in v) + f"   {tps[b]} ({(v[0] / v[b] - 1) * 100:+.1f}% vs auto)\n")
print(open(sys.argv[1]).read())
PY
