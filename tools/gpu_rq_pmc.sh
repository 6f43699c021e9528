#!/bin/bash
# counters of the sliced membership pass and its counting kernel (RK_DISTQ_SLICED=1) on configs[4]'s shape
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export RK_DISTQ_SLICED=1
printf 'SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD\nSQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES\n' | tools/pmc_pass.sh pmcQ2 "k_member_sliced" dist_rq_dev 100000 1000 3
python3 tools/pmc_summary.py gpurun_out/pmcQ2_*
