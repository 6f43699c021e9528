cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
head -6 tools/pmc_groups_sq.txt | tools/pmc_pass.sh pmcI "k_bucket_emit|k_part_scatter|k_part_hist" index 10000 3 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmcI_*
