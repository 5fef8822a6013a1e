cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_decode_nms.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r2_nms.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_nms -o nms -- python3 tools/postproc_bench.py 10 > gpurun_out/r2_postproc_prof.txt 2>&1
