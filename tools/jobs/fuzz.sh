cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 420 python tools/fuzz_parity.py 150 60000 2026 > gpurun_out/r03_fuzz.txt 2>&1; tail -3 gpurun_out/r03_fuzz.txt
timeout -k 10 200 python tools/soak.py 150 > gpurun_out/r03_soak.txt 2>&1; tail -3 gpurun_out/r03_soak.txt
timeout -k 10 300 python -m pytest tests/test_gpu_multiprocess.py tests/test_bench_contract.py -x -q 2>&1 | tail -5
