set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for n in 50000 28284 20000 14142; do
for mw in 3 2; do
RK_DIST_PAIR_MINWG=$mw timeout -k 10 300 python3 tools/prof_driver.py dist $n 60 1 16 > gpurun_out/d.log 2>&1 || { tail -20 gpurun_out/d.log; exit 1; }
echo n $n minwg $mw $(tail -1 gpurun_out/d.log | cut -c1-22)
done
done
