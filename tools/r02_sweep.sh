cd $GRAFT_REPO_ROOT
for cfg in "" "RK_DIST_PAIR=2" "RK_DIST_PAIR=2 RK_DIST_THREADS=512" "RK_DIST_THREADS=256" "RK_DIST_THREADS=768" "RK_DIST_PERSIST=2" "RK_DIST_PAIR=2 RK_DIST_PERSIST=2" "RK_DIST_PAIR=2 RK_DIST_THREADS=256 RK_DIST_ROWS=1"; do
  for shard in "8 16" "4 16" "2 16"; do
    echo -n "[$cfg] shard $shard: "; env $cfg python3 tools/prof_driver.py dist 10000 100 $shard 2>&1 | tail -1
  done
done
