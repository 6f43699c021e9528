set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for kb in 0 100 80 64 48; do
RK_DIST_LDS_KB=$kb timeout -k 10 300 python3 tools/prof_driver.py dist 50000 20 > gpurun_out/d.log 2>&1 || { tail -20 gpurun_out/d.log; exit 1; }
echo lds_kb $kb; tail -1 gpurun_out/d.log
done
