cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
POCS_LIB=ab_build/libpocs_calltimes.so POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r04_calltimes.err > /dev/null
grep "\[call\]" gpurun_out/r04_calltimes.err | tail -120 | head -60
