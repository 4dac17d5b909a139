# The bench lines of DESIGN.md section 7, ONE box: the driver's invocation, the default, cfg3, cfg5 (streaming and fused), mc
# (in the Infinity Cache and past it), the sweep over runs per call, smoke.   usage: final_bench.sh [TAG]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${1:-r04}
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_steps20warmup5.json 2>gpurun_out/${T}_bench_steps20.err
python bench.py > gpurun_out/${T}_bench_default.json 2>/dev/null
python bench.py --workload cfg3 --steps 16 --warmup 16 --no-cpu-baseline > gpurun_out/${T}_bench_cfg3.json 2>/dev/null
python bench.py --workload cfg5 --no-cpu-baseline > gpurun_out/${T}_bench_cfg5.json 2>/dev/null
python bench.py --workload cfg5 --mc-fused --no-cpu-baseline > gpurun_out/${T}_bench_cfg5_fused.json 2>/dev/null
python bench.py --workload mc --no-cpu-baseline > gpurun_out/${T}_bench_mc.json 2>/dev/null
python bench.py --workload mc --batch 16 --steps 32 --warmup 16 --no-cpu-baseline > gpurun_out/${T}_bench_mc_nt.json 2>/dev/null
for b in 1 8 20 64; do POCS_SKIP_SINGLE=1 POCS_BENCH_TARGET_S=0.5 python bench.py --batch $b --steps $((b*2 > 16 ? b*2 : 16)) --warmup $b --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b: value %.4g ms/step %.4f period %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us']))"; done > gpurun_out/${T}_sweep.txt
cat gpurun_out/${T}_sweep.txt
for f in steps20warmup5 default cfg3 cfg5 cfg5_fused mc mc_nt; do python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_$f.json')); r=d['roofline']; t=d['timing']
print('$f: value %.4g ms/step %.4f [%.4f %.4f] x%d single %s period %.1f us (bracketed %.1f) bound %s frac %s traffic_frac %s board %s' % (d['value'], d['ms_per_step'], t['ms_per_step_min'], t['ms_per_step_max'], t['repeats'], d.get('single_call_evals_per_s'), r['avg_kernel_us'], r['bracketed_kernel_us'], r['bound'], r['frac'], r.get('traffic_frac'), {k: r['board'][k] for k in ('sclk_MHz','power_W')} if r.get('board') else None))
print('   limiter', {k: v for k, v in r['limiter'].items() if k != 'kind'})
if 'cpu_baseline' in d: c=d['cpu_baseline']; print('   cpu', c['value'], c.get('all_cores', {}).get('value'), c.get('host_cores', {}).get('value'), c.get('budget'))
"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
