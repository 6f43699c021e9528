#!/bin/bash
# Developer loop on the GPU box (gpurun -- 'bash tools/gpu_quick.sh <what>'): the GPU tests of one area + its timings.
#   dist    self join: tests, 10k / 50k / 28,284 genomes, a 1/8 shard
#   shards  50k in 1 / 2 / 4 / 8 row shards (with the bands printed)
#   rq      ref-vs-query: tests, 100k x 1,000
#   sketch  sketcher: tests, 128 x 5 Mb, vector / scalar / LDS instruction counts
set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" > gpurun_out/q.log 2>&1 || { tail -20 gpurun_out/q.log; exit 1; }; grep -v amdgpu.ids gpurun_out/q.log | tail -${LINES_SHOWN:-1}; }
tests() { timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "$1" > gpurun_out/q_tests.log 2>&1 || { tail -60 gpurun_out/q_tests.log; exit 1; }; tail -2 gpurun_out/q_tests.log; }
case "$1" in
dist)
  tests "not sketch"
  for n in 10000 50000 28284; do echo "n $n: $(drv dist $n 200)"; done
  echo "1/8 shard of 10000: $(drv dist 10000 200 8 16)" ;;
shards)
  for s in 1 2 4 8; do echo "50000 genomes, $s shard(s):"; RK_DIST_DEBUG=1 LINES_SHOWN=8 drv dist 50000 60 $s 16 | sort -u | cut -c1-140; done ;;
rq)
  tests "not sketch"
  echo "$(drv dist_rq_dev 100000 1000 20)" ;;
sketch)
  tests "sketch or golden or cli or config1"
  LINES_SHOWN=4 drv sketch 128 5000000 6
  printf 'SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES\n' | tools/pmc_pass.sh pmcX rk_sketch_kernel sketch 128 5000000 || exit 1
  python3 tools/pmc_summary.py gpurun_out/pmcX_1 ;;
*) echo "usage: $0 dist|shards|rq|sketch"; exit 2 ;;
esac
