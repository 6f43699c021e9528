# instruction counts of k_bucket_emit per phase (developer ablations RK_INDEX_DEBUG: 8 load only, 16 + LDS counting sort, 32 + rank, 4 + heads, 1 all but the slice scatter, 0 all)
cd $GRAFT_REPO_ROOT
for d in 8 16 32 4 1 0; do
  export RK_INDEX_DEBUG=$d
  printf "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD\nSQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS\n" | bash tools/pmc_pass.sh pmcE${d} "k_bucket_emit" index 10000 2 > /dev/null
  echo "debug $d"; python3 tools/pmc_summary.py gpurun_out/pmcE${d}_* | grep k_bucket_emit
done
