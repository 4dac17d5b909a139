cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g ms/step %.4f period %.1f us groups %d' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['concurrent_launches']))"; }
for b in 2 3 4 6; do for g in 1 2; do
  POCS_SUB_BATCHES=$g POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.5 python bench.py --batch $b --steps $((4*b)) --warmup $b --no-cpu-baseline 2>/dev/null | line "runs $b G=$g"
done; done
for g in 1 2; do
  POCS_SUB_BATCHES=$g POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.5 python bench.py --workload cfg3 --batch 2 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line "cfg3 2 runs G=$g"
  POCS_SUB_BATCHES=$g POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.5 python bench.py --samples 10000 --batch 64 --steps 64 --warmup 64 --no-cpu-baseline 2>/dev/null | line "N=1e4 64 runs G=$g"
  POCS_SUB_BATCHES=$g POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.5 python bench.py --samples 100000 --batch 16 --steps 16 --warmup 16 --no-cpu-baseline 2>/dev/null | line "N=1e5 16 runs G=$g"
done
