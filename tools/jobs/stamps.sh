cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for l in 1 0; do
  echo "== lone $l"
  POCS_LONE=$l POCS_LIB=ablate_build/libpocs_stamps.so POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py --batch 1 --steps 16 --warmup 4 --no-cpu-baseline 2>&1 >/dev/null | grep stamps | tail -2
done
