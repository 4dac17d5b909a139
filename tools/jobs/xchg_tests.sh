# the sharded paths after a change to the exchange: multi-process GPU tests (two ranks on one card) + the bench contract's two-rank lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multiprocess.py tests/test_bench_contract.py -m gpu -x -q > gpurun_out/xchg_tests.txt 2>&1; rc=$?; tail -4 gpurun_out/xchg_tests.txt; exit $rc
