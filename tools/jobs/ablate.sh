# The ablation table of DESIGN.md section 5 (tools/ablate.sh ablate first): every timing-only build with and without the
# sample stores, 64 runs per launch, ONE box.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for store in 1 0; do
for v in "" RNG BOXMULLER PHILOX COLLIDE MOMENTS RNG_COLLIDE RNG_COLLIDE_MOMENTS; do
  lib=probability-of-collision-for-safe-planning_amd/libpocs.so; [ -n "$v" ] && lib=ab_build/libpocs_$v.so
  [ -f $lib ] || { echo "store $store ${v:-full}: $lib missing (tools/ablate.sh ablate builds it): skipped"; continue; }
  POCS_NO_STORE=$((1-store)) POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.3 python bench.py --steps 64 --warmup 64 --no-cpu-baseline 2>gpurun_out/ablate.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('store $store %-40s value %.4g kernel %.1f us' % ('${v:-full}', d['value'], d['roofline']['avg_kernel_us']))" || tail -3 gpurun_out/ablate.err
done
done
