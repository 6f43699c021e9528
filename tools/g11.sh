cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/prof_driver.py sketch 1000 5000000 > gpurun_out/big.log 2>&1; echo rc=$?; grep -v amdgpu.ids gpurun_out/big.log | tail -12
