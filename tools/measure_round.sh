# Round measurement on the GPU box (gpurun -- 'bash tools/measure_round.sh r03'): FETCH_SIZE calibration at the access
# widths the kernels use, the counter passes (one group per pass, kernel trace only) for the three hot kernels, so that
# bench.py reports the HBM traffic measured in this very call; the kernel trace of the index build; then bench.py and the
# rocprofv3 kernel trace of the same command.  Outputs under gpurun_out/; tools/collect_profiles.py <tag> turns the
# summaries into profiles/<tag>_*.
tag=${1:-r05}
phase=${2:-all}   # pmc: the counter passes + kernel trace of the index build; bench: bench.py and its kernel trace; all: both (two gpurun calls fit their limits better)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ "$phase" != bench ]; then
for w in 4 8 16; do
  printf 'FETCH_SIZE\n' | tools/pmc_pass.sh pmcC$w k_calib_read calib 1024 $w || exit 1
done
# (the near-window kernel: collections below 4,000 genomes, small row shards of an index with slice records; RK_INDEX_TILES=0
# makes the build emit slice records for the 10,000- and 50,000-genome collections)
RK_INDEX_TILES=0 tools/pmc_pass.sh pmcD rk_near_kernel dist 10000 4 < tools/pmc_groups_dist.txt || exit 1
printf 'FETCH_SIZE\nWRITE_SIZE\nSQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY\nSQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA\nSQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD\n' > /tmp/groups_short.txt
RK_INDEX_TILES=0 tools/pmc_pass.sh pmcD50 rk_near_kernel dist 50000 3 < /tmp/groups_short.txt || exit 1
# (the tile kernel: every self join over 4,000 genomes and more, from the first one on -- the headline and config3 --, and wide species)
tools/pmc_pass.sh pmcT10 rk_tile_kernel dist 10000 4 < /tmp/groups_short.txt || exit 1
tools/pmc_pass.sh pmcT50 rk_tile_kernel dist 50000 3 < /tmp/groups_short.txt || exit 1
tools/pmc_pass.sh pmcT100 rk_tile_kernel dist 10000 4 1 0 0 100 < /tmp/groups_short.txt || exit 1
tools/pmc_pass.sh pmcT1000 rk_tile_kernel dist 10000 4 1 0 0 1000 < /tmp/groups_short.txt || exit 1
printf 'FETCH_SIZE\nWRITE_SIZE\nSQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES\n' | tools/pmc_pass.sh pmcSk1000 rk_scan2_kernel sketch 1000 5000000 2 || exit 1
tools/pmc_pass.sh pmcQ rk_distq_kernel dist_rq_dev 100000 1000 3 < tools/pmc_groups_rq.txt || exit 1
printf 'FETCH_SIZE\nWRITE_SIZE\nTCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum\n' | tools/pmc_pass.sh pmcSk rk_scan2_kernel sketch 128 5000000 || exit 1
tools/pmc_pass.sh pmcS rk_scan2_kernel sketch 128 5000000 < tools/pmc_groups_sketch.txt || exit 1
RK_SKETCH_IMG=1 tools/pmc_pass.sh pmcS1 rk_sketch_kernel sketch 128 5000000 < tools/pmc_groups_sketch.txt || exit 1
sed -n '1,3p;5p' tools/pmc_groups_sq.txt | tools/pmc_pass.sh pmcI "k_bucket_emit|k_part_coarse|k_part_fine|k_part_hist|k_trec|k_tdir" index_only 10000 3 || exit 1
echo "counter passes done"
bash tools/kernel_trace.sh prof_index index 10000 6 > gpurun_out/index_kernels.txt 2>&1 || { tail -5 gpurun_out/index_kernels.txt; exit 1; }
python3 tools/index_timeline.py gpurun_out/prof_index 4 >> gpurun_out/index_kernels.txt 2>&1
fi
[ "$phase" = pmc ] && exit 0
# (phase all: the counter records of this very call feed bench.py; phase bench: run `python3 tools/collect_profiles.py <tag>
# --traffic-only` in the build container after the pmc phase -- gpurun_out/ does not travel, profiles/ does)
[ "$phase" = all ] && { python3 tools/collect_profiles.py $tag --traffic-only || exit 1; }
python3 bench.py > gpurun_out/bench_round.json 2> gpurun_out/bench_round.err || { tail -20 gpurun_out/bench_round.err; exit 1; }
tail -c 300 gpurun_out/bench_round.json; echo
rm -rf gpurun_out/prof_final
( cd /tmp && timeout -k 10 700 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_final.err ) || { tail -5 gpurun_out/prof_final.err; exit 1; }
echo "kernel trace done"
