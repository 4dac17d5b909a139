cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for store in 1 0; do
for v in "" RNG BOXMULLER PHILOX COLLIDE MOMENTS RNG_COLLIDE RNG_COLLIDEDPOCS_ABLATE_MOMENTS; do
  lib=probability-of-collision-for-safe-planning_amd/libpocs.so; [ -n "$v" ] && lib=ablate_build/libpocs_$v.so
  POCS_NO_STORE=$((1-store)) POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py --steps 64 --warmup 64 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('store $store %-40s value %.4g kernel %.1f us' % ('${v:-full}', d['value'], d['roofline']['avg_kernel_us']))"
done
done
