# one environment variable of the HIP runtime on / off, same library, alternating, ONE box.   usage: ab_env.sh NAME=VALUE
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g ms/step %.4f period %.2f us frac %.3f' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['frac']))"; }
{
for i in 1 2 3; do
for v in on off; do
  E="POCS_DUMMY=1"; [ $v = on ] && E="$1"
  env $E POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --batch 1 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | line "$v 1 run"
  env $E POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --batch 4 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | line "$v 4 runs"
  env $E POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "$v 20 runs"
done
done
} > gpurun_out/abe_ab.txt 2>&1; cat gpurun_out/abe_ab.txt
