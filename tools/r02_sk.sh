set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "sketch or golden or cli or config1" > gpurun_out/gpu_sk_tests.log 2>&1 || { tail -60 gpurun_out/gpu_sk_tests.log; exit 1; }
tail -3 gpurun_out/gpu_sk_tests.log
timeout -k 10 200 python3 tools/prof_driver.py sketch 128 5000000 6 > gpurun_out/sk.log 2>&1 || { tail -20 gpurun_out/sk.log; exit 1; }
cat gpurun_out/sk.log
printf 'SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES\n' | tools/pmc_pass.sh pmcX rk_sketch_kernel sketch 128 5000000 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmcX_1
