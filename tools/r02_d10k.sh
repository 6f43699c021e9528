set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3; do
timeout -k 10 300 python3 tools/prof_driver.py dist 10000 200 > gpurun_out/d10k.log 2>&1 || { tail -20 gpurun_out/d10k.log; exit 1; }
tail -1 gpurun_out/d10k.log
done
