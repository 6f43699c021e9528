# Round measurement on the GPU box: the counter passes (one group per pass, kernel trace only) for the three hot
# kernels first, so that bench.py reports the HBM traffic measured in this very call; then bench.py and the rocprofv3
# kernel trace of the same command.  Outputs under gpurun_out/; the summaries are turned into profiles/<tag>_* by
# tools/collect_profiles.py.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tools/pmc_pass.sh pmcD rk_dist_kernel dist 10000 4 < tools/pmc_groups_dist.txt || exit 1
tools/pmc_pass.sh pmcQ rk_distq_kernel dist_rq_dev 100000 1000 3 < tools/pmc_groups_rq.txt || exit 1
printf 'FETCH_SIZE\nWRITE_SIZE\n' | tools/pmc_pass.sh pmcSk rk_sketch_kernel sketch 128 5000000 || exit 1
tools/pmc_pass.sh pmcS rk_sketch_kernel sketch 128 5000000 < tools/pmc_groups_sketch.txt || exit 1
RK_SKETCH_IMG=0 tools/pmc_pass.sh pmcS0 rk_sketch_kernel sketch 128 5000000 < tools/pmc_groups_sketch.txt || exit 1
echo "counter passes done"
python3 tools/collect_profiles.py r02 --traffic-only || exit 1
python3 bench.py > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err || { tail -20 gpurun_out/bench_r2.err; exit 1; }
tail -c 300 gpurun_out/bench_r2.json; echo
rm -rf gpurun_out/prof_final
( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_final.err ) || { tail -5 gpurun_out/prof_final.err; exit 1; }
echo "kernel trace done"
