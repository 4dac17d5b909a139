cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multiprocess.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -8
