# calls of 2 .. 7 runs: ab_build/libpocs_base.so ("base") against the tree ("new"), alternating, ONE box; first the parity tests of launch shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lone or single or launch or runs_per or batch" > gpurun_out/abs_tests.txt 2>&1; rc=$?; tail -3 gpurun_out/abs_tests.txt; [ $rc = 0 ] || exit $rc
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g ms/step %.4f period %.2f us' % (d['value'], d['ms_per_step'], r.get('waypoint_us') or r['avg_kernel_us']))"; }
{
for i in 1 2; do
for b in 2 3 4 5 6 7; do
for v in new base; do
  lib=ab_build/libpocs_$v.so; [ $v = new ] && lib=""
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.5 python bench.py --batch $b --steps $((b*4)) --warmup $b --no-cpu-baseline 2>/dev/null | line "$v runs $b"
done
done
done
} > gpurun_out/abs_ab.txt 2>&1; cat gpurun_out/abs_ab.txt
