# MC kernels: ab_build/libpocs_base.so ("base") against the tree ("new"): MC parity tests, then mc (in the Infinity Cache), mc past it, cfg5, alternating, ONE box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "mc or MC or particle or simulation or cfg5" > gpurun_out/abm_tests.txt 2>&1; rc=$?; tail -3 gpurun_out/abm_tests.txt; [ $rc = 0 ] || exit $rc
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g ms/step %.4f period %.2f us' % (d['value'], d['ms_per_step'], r['avg_kernel_us']))"; }
{
for i in 1 2 3; do
for v in new base; do
  lib=ab_build/libpocs_$v.so; [ $v = new ] && lib=""
  POCS_LIB=$lib POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --workload mc --no-cpu-baseline 2>/dev/null | line "$v mc"
  POCS_LIB=$lib POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --workload mc --batch 16 --steps 32 --warmup 16 --no-cpu-baseline 2>/dev/null | line "$v mc_nt"
  POCS_LIB=$lib POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | line "$v cfg5"
done
done
} > gpurun_out/abm_ab.txt 2>&1; cat gpurun_out/abm_ab.txt
