# the lone form (one run per call): the named ab_build library ("base") against the tree ("new"), alternating, ONE box;
# first the parity tests of the single call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lone or single or launch or runs_per" > gpurun_out/abl_tests.txt 2>&1; rc=$?; tail -3 gpurun_out/abl_tests.txt; [ $rc = 0 ] || exit $rc
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g ms/step %.4f period %.2f us frac %.3f' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['frac']))"; }
{
for i in 1 2 3; do
for v in new base; do
  lib=ab_build/libpocs_$v.so; [ $v = new ] && lib=""
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --batch 1 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | line "$v 1 run"
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --workload cfg3 --batch 1 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line "$v cfg3 1 run"
done
done
} > gpurun_out/abl_ab.txt 2>&1; cat gpurun_out/abl_ab.txt
