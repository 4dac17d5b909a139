cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python tools/fuzz_parity.py 400 60000 777 > gpurun_out/r03_fuzz_long.txt 2>&1; tail -2 gpurun_out/r03_fuzz_long.txt
timeout -k 10 700 python tools/soak.py 600 > gpurun_out/r03_soak_long.txt 2>&1; tail -2 gpurun_out/r03_soak_long.txt
