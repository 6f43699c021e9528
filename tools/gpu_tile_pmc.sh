#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export RK_DIST_TILES=1
head -5 tools/pmc_groups_sq.txt | tools/pmc_pass.sh pmcT rk_tile_kernel dist ${1:-50000} 3 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmcT_*
