# MFMA utilisation counters of the bench's kernels (one rocprofv3 --pmc pass).  usage: bash tools/prof_mfma.sh <tag>
tag=${1:-r3}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_${tag}_m -o m -- python3 bench.py --no-train --no-cpu-baseline --no-two-lanes --steps 3 --warmup 1 --min-seconds 0 > $out/${tag}_pmc_m.log 2>&1 || exit 1
f=$(find $out/pmc_${tag}_m -name '*counter_collection.csv' | head -1)
python3 tools/pmc_mfma.py "$f" > $out/${tag}_pmc_mfma.txt || exit 1
rm -rf $out/pmc_${tag}_m
