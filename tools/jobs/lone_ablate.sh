# where the lone form's sampling phase goes: timing-only ablation builds (tools/ablate.sh ablate, on the box) at ONE run per call, with and without the sample stores
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/ablate.sh ablate > /dev/null || exit 1
for store in 1 0; do
for v in "" RNG COLLIDE MOMENTS RNG_COLLIDE RNG_COLLIDE_MOMENTS; do
  lib=probability-of-collision-for-safe-planning_amd/libpocs.so; [ -n "$v" ] && lib=ab_build/libpocs_$v.so
  POCS_NO_STORE=$((1-store)) POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.3 timeout -k 10 200 python bench.py --batch 1 --steps 16 --warmup 4 --no-cpu-baseline 2>gpurun_out/ablate.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('store $store %-40s value %.4g period %.2f us' % ('${v:-full}', d['value'], d['roofline']['avg_kernel_us']))" || tail -3 gpurun_out/ablate.err
done
done > gpurun_out/lone_ablate.txt 2>&1
cat gpurun_out/lone_ablate.txt
