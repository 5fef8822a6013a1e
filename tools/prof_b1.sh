cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1 -o b1 -- python3 tools/batch1_loop.py 1 > gpurun_out/r2_b1_prof.log 2>&1 || exit 1
python3 tools/rocpd_top.py gpurun_out/prof_b1/b1_results.db 40 > gpurun_out/r2_b1_kernels.csv
