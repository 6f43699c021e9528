cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
head -4 tools/pmc_groups_sq.txt | tools/pmc_pass.sh pmcN "rk_near_kernel" dist 50000 4 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmcN_*
