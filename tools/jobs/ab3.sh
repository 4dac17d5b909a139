# usage: ab3.sh [libs...]  -- tree check of the in-tree library, then the named ab_build/libpocs_<name>.so and the
# in-tree one ("new") at 20 / 64 / 1 runs per call, alternating, twice: ONE box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g ms/step %.4f kernel %.1f us frac %.3f' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['frac']))"; }
timeout -k 10 300 python tools/tree_check.py 200000 2>&1 | tail -1
libs=${@:-base}
for i in 1 2; do
for n in $libs new; do
  lib=ab_build/libpocs_$n.so; [ $n = new ] && lib=""
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "$n 20"
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py --steps 64 --warmup 64 --no-cpu-baseline 2>/dev/null | line "$n 64"
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py --batch 1 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | line "$n 1"
done
done
