cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2 3; do for u in 1 0; do echo "u2 $u: $(RK_SCAN2_U2=$u timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -1)"; done; done
tools/pmc_pass.sh pmcS2 rk_scan2_kernel sketch 128 5000000 < tools/pmc_groups_sketch.txt || exit 1
python3 tools/pmc_summary.py gpurun_out/pmcS2_*
