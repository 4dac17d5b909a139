cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for lib in ablate_build/libpocs_stamps.so; do
for args in "--steps 20 --warmup 5" "--steps 64 --warmup 64"; do
  echo "== $lib $args"
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py $args --no-cpu-baseline 2>&1 >/dev/null | grep "WITHIN\|per-block" | tail -2 | cut -c1-260
done
done
