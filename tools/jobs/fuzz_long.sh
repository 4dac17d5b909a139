# a longer fuzz + soak than tools/jobs/fuzz.sh (end of round 4, numerics v9): 2 x 800 random worlds, 10 minutes of soak, the probe sweep at 64 rounds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python tools/fuzz_parity.py 800 60000 31415 > gpurun_out/r04_fuzz_long1.txt 2>&1; tail -1 gpurun_out/r04_fuzz_long1.txt
timeout -k 10 900 python tools/fuzz_parity.py 800 60000 27182 > gpurun_out/r04_fuzz_long2.txt 2>&1; tail -1 gpurun_out/r04_fuzz_long2.txt
timeout -k 10 500 python tools/soak.py 400 > gpurun_out/r04_soak_long.txt 2>&1; tail -1 gpurun_out/r04_soak_long.txt
timeout -k 10 600 python tools/probe_sweep.py 64 > gpurun_out/r04_probe_sweep_long.txt 2>&1; tail -1 gpurun_out/r04_probe_sweep_long.txt
