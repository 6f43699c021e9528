cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_cli.py -m gpu -x -q -k "big_fasta or streaming_pipeline" > gpurun_out/t7.log 2>&1 || { tail -40 gpurun_out/t7.log; exit 1; }
tail -3 gpurun_out/t7.log
timeout -k 10 600 python3 tools/cli_e2e.py big 3000000000 1 16 2>&1 | grep -v amdgpu.ids | tail -8
