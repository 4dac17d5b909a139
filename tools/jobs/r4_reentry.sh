# Round 4, re-entry: the GPU suite, smoke and the driver's bench line on the rebuilt library (the container was re-created).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_g_gputests.txt 2>&1
rc=$?
tail -5 gpurun_out/r04_g_gputests.txt
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 &&
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_g_bench_steps20.json 2>gpurun_out/r04_g_bench_steps20.err &&
python -c "
import json; d=json.load(open('gpurun_out/r04_g_bench_steps20.json')); r=d['roofline']
print('value %.4g ms/step %.4f period %.1f frac %s single %s' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['frac'], d.get('single_call_evals_per_s')))"
