set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for s in 2 4 8; do
for mr in 1024 2048 3072 4096 6144; do
RK_DIST_BAND_MIN_ROWS=$mr timeout -k 10 300 python3 tools/prof_driver.py dist 50000 60 $s 16 > gpurun_out/d.log 2>&1 || { tail -20 gpurun_out/d.log; exit 1; }
echo shards $s minrows $mr $(tail -1 gpurun_out/d.log | cut -c1-22)
done
done
