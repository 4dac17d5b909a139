cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multiprocess.py -m gpu -x -q -s -k cfg4 2>&1 | grep -E "cfg4 rehearsal|passed|failed|assert" | head
