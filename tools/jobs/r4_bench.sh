# round 4: the bench contract tests (new fields), then the driver's invocation and the default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_bench_contract.py -x -q > gpurun_out/r04_benchtests.txt 2>&1
rc=$?
tail -30 gpurun_out/r04_benchtests.txt
[ $rc = 0 ] || exit $rc
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_steps20.json 2>gpurun_out/r04_bench_steps20.err && python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_bench_steps20.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'numerics', 'timing', 'single_call_evals_per_s')})
r = d['roofline']
print({k: r[k] for k in ('bound', 'frac', 'avg_kernel_us', 'traffic')}, r['limiter'])
print(d['cpu_baseline'])
PY
