cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for args in "--steps 20 --warmup 5" "--steps 64 --warmup 64" "--batch 8 --steps 16 --warmup 8"; do
  echo "== $args"
  POCS_LIB=ablate_build/libpocs_stamps.so POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py $args --no-cpu-baseline 2>&1 >/dev/null | grep stamps | grep -v "second time" | tail -1
done
