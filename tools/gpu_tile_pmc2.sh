#!/bin/bash
# counters of the tile kernel's two variants at 50,000 genomes (RK_TILE_SROW=1 / 0): where the scalar row masks lose
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export RK_DIST_TILES=1
for v in 1 0; do
  export RK_TILE_SROW=$v
  printf 'SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS\nSQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY\nSQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ\nSQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS\n' | tools/pmc_pass.sh pmcTS$v "rk_tile_kernel" dist 50000 3 | tail -4
  echo "srow $v"; python3 tools/pmc_summary.py gpurun_out/pmcTS${v}_* | grep rk_tile
done
