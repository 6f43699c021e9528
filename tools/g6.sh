cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for uw in 1 2 4; do
echo "UW $uw: $(RK_DIST_NEAR_UW=$uw timeout -k 10 200 python3 tools/prof_driver.py dist 50000 50 8 16 0 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-30) | $(RK_DIST_NEAR_UW=$uw timeout -k 10 200 python3 tools/prof_driver.py dist 10000 100 8 16 0 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-30) | $(RK_DIST_NEAR_UW=$uw timeout -k 10 200 python3 tools/prof_driver.py dist 10000 100 1 0 0 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-30) | $(RK_DIST_NEAR_UW=$uw timeout -k 10 200 python3 tools/prof_driver.py dist 50000 50 2 16 0 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-30)"
done
