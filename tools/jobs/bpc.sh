cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for lib in probability-of-collision-for-safe-planning_amd/libpocs.so ablate_build/libpocs_se1.so ablate_build/libpocs_se2.so; do
  for args in "--steps 20 --warmup 5" "--steps 64 --warmup 64"; do
    POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $args: value %.4g kernel %.1f us P %s' % (d['value'], d['roofline']['avg_kernel_us'], d['config']['probability']))"
  done
done
done
