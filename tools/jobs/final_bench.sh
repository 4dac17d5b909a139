cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_steps20warmup5.json 2>gpurun_out/r03_bench_steps20.err
python bench.py > gpurun_out/r03_bench_default.json 2>/dev/null
python bench.py --workload cfg3 --steps 16 --warmup 16 --no-cpu-baseline > gpurun_out/r03_bench_cfg3.json 2>/dev/null
python bench.py --workload cfg5 --no-cpu-baseline > gpurun_out/r03_bench_cfg5.json 2>/dev/null
python bench.py --workload cfg5 --mc-fused --no-cpu-baseline > gpurun_out/r03_bench_cfg5_fused.json 2>/dev/null
python bench.py --workload mc --no-cpu-baseline > gpurun_out/r03_bench_mc.json 2>/dev/null
for b in 1 8 20 64; do POCS_SKIP_SINGLE=1 python bench.py --batch $b --steps $((b*2 > 16 ? b*2 : 16)) --warmup $b --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b: value %.4g ms/step %.4f kernel %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us']))"; done > gpurun_out/r03_sweep.txt
cat gpurun_out/r03_sweep.txt
for f in steps20warmup5 default cfg3 cfg5 cfg5_fused mc; do python -c "
import json; d=json.load(open('gpurun_out/r03_bench_$f.json')); r=d['roofline']
print('$f: value %.4g ms/step %.4f single %s kernel %.1f us frac %.3f traffic_frac %s board %s' % (d['value'], d['ms_per_step'], d.get('single_call_evals_per_s'), r['avg_kernel_us'], r['frac'], r.get('traffic_frac'), {k: r['board'][k] for k in ('sclk_MHz','power_W')} if r.get('board') else None))
print('   limiter', {k: v for k, v in r['limiter'].items() if k != 'kind'})
if 'cpu_baseline' in d: print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('all_cores'))
"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
