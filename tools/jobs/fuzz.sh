# fuzz + soak beyond what the suite affords (tools/fuzz_parity.py, tools/soak.py), then the multi-process tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 420 python tools/fuzz_parity.py 200 60000 2027 > gpurun_out/r04_fuzz.txt 2>&1; tail -3 gpurun_out/r04_fuzz.txt
timeout -k 10 420 python tools/fuzz_parity.py 300 60000 4 > gpurun_out/r04_fuzz2.txt 2>&1; tail -2 gpurun_out/r04_fuzz2.txt
timeout -k 10 200 python tools/soak.py 150 > gpurun_out/r04_soak.txt 2>&1; tail -3 gpurun_out/r04_soak.txt
timeout -k 10 300 python -m pytest tests/test_gpu_multiprocess.py -x -q 2>&1 | tail -3
