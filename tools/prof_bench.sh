# Measurements of the inference bench on one GPU box: the default bench line, per-kernel statistics of the same
# workload under rocprofv3, and the two HBM-traffic counter passes (FETCH_SIZE and WRITE_SIZE apart).
# usage (on the GPU box, from the repo root): bash tools/prof_bench.sh <tag>     -> gpurun_out/<tag>_*
tag=${1:-r3}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
timeout -k 10 400 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_${tag} -o bench -- python3 bench.py --no-train --no-cpu-baseline --no-two-lanes > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err || exit 1
python3 tools/rocpd_top.py $out/prof_${tag}/bench_results.db 40 > $out/${tag}_bench_kernel_stats.csv || exit 1
rm -rf $out/prof_${tag}
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_${tag}_f -o f -- python3 bench.py --no-train --no-cpu-baseline --no-two-lanes --steps 3 --warmup 1 --min-seconds 0 > $out/${tag}_pmc_f.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_${tag}_w -o w -- python3 bench.py --no-train --no-cpu-baseline --no-two-lanes --steps 3 --warmup 1 --min-seconds 0 > $out/${tag}_pmc_w.log 2>&1 || exit 1
f=$(find $out/pmc_${tag}_f -name '*counter_collection.csv' | head -1)
w=$(find $out/pmc_${tag}_w -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py "$f" "$w" $out/${tag}_pmc_traffic.json > $out/${tag}_pmc_hbm_traffic.txt || exit 1
rm -rf $out/pmc_${tag}_f $out/pmc_${tag}_w
