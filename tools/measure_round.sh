# Round measurement on the GPU box (gpurun -- 'bash tools/measure_round.sh r03'): FETCH_SIZE calibration at the access
# widths the kernels use, the counter passes (one group per pass, kernel trace only) for the three hot kernels, so that
# bench.py reports the HBM traffic measured in this very call; the kernel trace of the index build; then bench.py and the
# rocprofv3 kernel trace of the same command.  Outputs under gpurun_out/; tools/collect_profiles.py <tag> turns the
# summaries into profiles/<tag>_*.
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in 4 8 16; do
  printf 'FETCH_SIZE\n' | tools/pmc_pass.sh pmcC$w k_calib_read calib 1024 $w || exit 1
done
# (the near-window kernel: what a first join, a row shard and the command-line tool run; RK_DIST_TILES_AFTER keeps a repeatedly
# joined index on it)
RK_DIST_TILES_AFTER=1000000 tools/pmc_pass.sh pmcD rk_near_kernel dist 10000 4 < tools/pmc_groups_dist.txt || exit 1
printf 'FETCH_SIZE\nWRITE_SIZE\nSQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY\nSQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA\nSQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD\n' > /tmp/groups_short.txt
RK_DIST_TILES_AFTER=1000000 tools/pmc_pass.sh pmcD50 rk_near_kernel dist 50000 3 < /tmp/groups_short.txt || exit 1
# (the tile kernel: a resident index from its second join on -- the headline and config3 --, and wide species)
tools/pmc_pass.sh pmcT10 rk_tile_kernel dist 10000 4 < /tmp/groups_short.txt || exit 1
tools/pmc_pass.sh pmcT50 rk_tile_kernel dist 50000 3 < /tmp/groups_short.txt || exit 1
tools/pmc_pass.sh pmcT100 rk_tile_kernel dist 10000 4 1 0 0 100 < /tmp/groups_short.txt || exit 1
tools/pmc_pass.sh pmcT1000 rk_tile_kernel dist 10000 4 1 0 0 1000 < /tmp/groups_short.txt || exit 1
printf 'FETCH_SIZE\nWRITE_SIZE\nSQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES\n' | tools/pmc_pass.sh pmcSk1000 rk_scan2_kernel sketch 1000 5000000 2 || exit 1
tools/pmc_pass.sh pmcQ rk_distq_kernel dist_rq_dev 100000 1000 3 < tools/pmc_groups_rq.txt || exit 1
printf 'FETCH_SIZE\nWRITE_SIZE\nTCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum\n' | tools/pmc_pass.sh pmcSk rk_scan2_kernel sketch 128 5000000 || exit 1
tools/pmc_pass.sh pmcS rk_scan2_kernel sketch 128 5000000 < tools/pmc_groups_sketch.txt || exit 1
RK_SKETCH_IMG=1 tools/pmc_pass.sh pmcS1 rk_sketch_kernel sketch 128 5000000 < tools/pmc_groups_sketch.txt || exit 1
head -3 tools/pmc_groups_sq.txt | tools/pmc_pass.sh pmcI "k_bucket_emit|k_part_scatter|k_part_hist" index 10000 3 || exit 1
echo "counter passes done"
python3 tools/collect_profiles.py $tag --traffic-only || exit 1
bash tools/kernel_trace.sh prof_index index 10000 6 > gpurun_out/index_kernels.txt 2>&1 || { tail -5 gpurun_out/index_kernels.txt; exit 1; }
python3 bench.py > gpurun_out/bench_round.json 2> gpurun_out/bench_round.err || { tail -20 gpurun_out/bench_round.err; exit 1; }
tail -c 300 gpurun_out/bench_round.json; echo
rm -rf gpurun_out/prof_final
( cd /tmp && timeout -k 10 700 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_final.err ) || { tail -5 gpurun_out/prof_final.err; exit 1; }
echo "kernel trace done"
