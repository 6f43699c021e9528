#!/bin/bash
# vector instructions of k_bucket_emit_tiles per phase (developer ablations RK_INDEX_DEBUG: 8 load only, 16 + LDS counting sort, 32 + rank, 4 + numbering, 1 all but the tile records, 0 all)
cd $GRAFT_REPO_ROOT
for d in 8 16 32 4 1 0; do
  export RK_INDEX_DEBUG=$d
  printf "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU\n" | bash tools/pmc_pass.sh pmcE${d} "k_bucket_emit" index_only 10000 2 > /dev/null
  echo "debug $d"; python3 tools/pmc_summary.py gpurun_out/pmcE${d}_* | grep k_bucket_emit
done
