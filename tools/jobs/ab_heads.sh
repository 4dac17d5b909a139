# the heads' and the closer's requests in one round trip: GPU suite on the tree, then ab_build/libpocs_base.so ("base") against
# the tree ("new") at 1 / 4 / 20 / 64 runs per call and cfg3 at one run, alternating, ONE box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/abh_gputests.txt 2>&1; rc=$?; tail -3 gpurun_out/abh_gputests.txt; [ $rc = 0 ] || exit $rc
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g ms/step %.4f period %.2f us frac %.3f' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['frac']))"; }
{
for i in 1 2 3; do
for v in new base; do
  lib=ab_build/libpocs_$v.so; [ $v = new ] && lib=""
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --batch 1 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | line "$v 1 run"
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --batch 4 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | line "$v 4 runs"
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "$v 20 runs"
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --steps 64 --warmup 64 --no-cpu-baseline 2>/dev/null | line "$v 64 runs"
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --workload cfg3 --batch 1 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line "$v cfg3 1 run"
done
done
} > gpurun_out/abh_ab.txt 2>&1; cat gpurun_out/abh_ab.txt
