# Kernel statistics of the training step (rocprofv3 --kernel-trace --stats of tools/train_bench.py), with the weight
# gradients on their side stream and on the main stream.  usage (GPU box, repo root): bash tools/prof_train.sh <tag>
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_train -o t -- python3 tools/train_bench.py --batch 64 --size 416 --steps 8 --warmup 2 > $out/${tag}_train_bench.json 2> $out/${tag}_train.err || exit 1
python3 tools/rocpd_top.py $out/prof_${tag}_train/t_results.db 45 > $out/${tag}_train_kernel_stats.csv || exit 1
rm -rf $out/prof_${tag}_train
export YOLO_WGRAD_STREAM=0
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_train1 -o t -- python3 tools/train_bench.py --batch 64 --size 416 --steps 8 --warmup 2 > $out/${tag}_train_bench_1stream.json 2> $out/${tag}_train1.err || exit 1
python3 tools/rocpd_top.py $out/prof_${tag}_train1/t_results.db 45 > $out/${tag}_train_kernel_stats_1stream.csv || exit 1
rm -rf $out/prof_${tag}_train1
