# Per-block phase stamps of k_gmm_step (tools/ablate.sh stamps builds ab_build/libpocs_stamps.so: -DPOCS_TUNING
# -DPOCS_STAMPS) at one, 20 and 64 runs per launch: where a block's time goes.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for args in "--batch 1 --steps 16 --warmup 4" "--steps 20 --warmup 5" "--steps 64 --warmup 64"; do
  echo "== $args"
  POCS_LIB=ab_build/libpocs_stamps.so POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.2 python bench.py $args --no-cpu-baseline 2>&1 >/dev/null | grep stamps | tail -2
done
