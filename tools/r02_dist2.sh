set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for t in 512 768 1024 256; do
RK_DIST_THREADS=$t timeout -k 10 300 python3 tools/prof_driver.py dist 10000 200 > gpurun_out/d.log 2>&1 || { tail -20 gpurun_out/d.log; exit 1; }
echo threads $t; tail -1 gpurun_out/d.log
done
