# usage: ab_pair.sh NAME [rounds]: ab_build/libpocs_NAME.so against the in-tree library, alternating, 20 and 64 runs per call, ONE box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
n=$1; rounds=${2:-4}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: ms/step %.4f period %.1f us frac %.3f' % (d['ms_per_step'], r['avg_kernel_us'], r['frac']))"; }
for i in $(seq $rounds); do
for v in new $n; do
  lib=ab_build/libpocs_$v.so; [ $v = new ] && lib=""
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=1.0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "$v 20"
  POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=1.0 python bench.py --steps 64 --warmup 64 --no-cpu-baseline 2>/dev/null | line "$v 64"
done
done
