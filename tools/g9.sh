cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tools/pmc_pass.sh pmcS2 rk_scan2_kernel sketch 128 5000000 < tools/pmc_groups_sketch.txt || exit 1
python3 tools/pmc_summary.py gpurun_out/pmcS2_*
