# calls of fewer than 8 runs: one launch per waypoint (G = 1, what the library picks there) against two sub-batches (G = 2, forced), ONE box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g ms/step %.4f period %.2f us' % (d['value'], d['ms_per_step'], r.get('waypoint_us') or r['avg_kernel_us']))"; }
{
for i in 1 2; do
for b in 2 3 4 6 7; do
for g in 1 2; do
  POCS_SUB_BATCHES=$g POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.5 python bench.py --batch $b --steps $((b*4)) --warmup $b --no-cpu-baseline 2>/dev/null | line "runs $b G=$g"
done
done
done
} > gpurun_out/small_batches.txt 2>&1; cat gpurun_out/small_batches.txt
