cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
head -4 tools/pmc_groups_sq.txt | tools/pmc_pass.sh pmcN "rk_near_kernel" dist 10000 4 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmcN_*
bash tools/kernel_trace.sh ktn dist 10000 20 1 0 0 > /dev/null 2>&1; f=$(find gpurun_out/ktn -name "*kernel_stats.csv" | head -1); grep "rk_" $f | cut -c1-160
