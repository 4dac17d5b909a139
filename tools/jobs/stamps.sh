# Per-block phase stamps of k_gmm_step and the footprint test's statistics (tools/ablate.sh stamps builds
# ab_build/libpocs_stamps.so ON THE BOX: -DPOCS_TUNING -DPOCS_STAMPS) at one, 20 and 64 runs per launch and for cfg3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/ablate.sh stamps || exit 1
for args in "--batch 1 --steps 16 --warmup 4" "--steps 20 --warmup 5" "--steps 64 --warmup 64" "--workload cfg3 --steps 16 --warmup 16"; do
  echo "== $args"
  POCS_LIB=ab_build/libpocs_stamps.so POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.2 timeout -k 10 300 python bench.py $args --no-cpu-baseline 2>&1 >/dev/null | grep stamps | tail -3
done > gpurun_out/r04_stamps_v9.txt 2>&1
cat gpurun_out/r04_stamps_v9.txt
