set -e
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err
tail -c 600 gpurun_out/bench_r1.json; echo
export TMPDIR=/tmp
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_final.err )
tools/pmc_pass.sh pmcD rk_dist_kernel dist 10000 4 < tools/pmc_groups_dist.txt
printf 'FETCH_SIZE\nWRITE_SIZE\n' | tools/pmc_pass.sh pmcSk rk_sketch_kernel sketch 128 5000000
tools/pmc_pass.sh pmcS rk_sketch_kernel sketch 128 5000000 < tools/pmc_groups_sketch.txt
