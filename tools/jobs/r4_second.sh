# round 4, second job: phase stamps of the v8 kernel, the driver tests (run-ahead default), MC past the Infinity Cache
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/jobs/stamps.sh 2>&1 | tee gpurun_out/r04_stamps_a.txt
timeout -k 10 600 python -m pytest tests/test_driver.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -5
for b in 8 16; do
  POCS_SKIP_SINGLE=1 python bench.py --workload mc --batch $b --steps $((2*b)) --warmup $b --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('mc batch $b: value %.4g kernel %.1f us achieved %.0f GB/s frac %s resident %s bound %s copy %.0f' % (d['value'], r['avg_kernel_us'], r['achieved'], r['frac'], r['resident'], r['bound'], r['copy_GBps']))"
done
