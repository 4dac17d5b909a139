# the round's last tree: GPU suite + smoke, the lone form under the profiler again (its head changed after r04_h), the driver's line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_i_gputests.txt 2>&1; rc=$?; tail -3 gpurun_out/r04_i_gputests.txt; [ $rc = 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
export POCS_SKIP_SINGLE=1 POCS_BENCH_TARGET_S=0.3
sed -n '/^one()/,/^}/p' tools/jobs/r4_prof.sh > /tmp/one_fn.sh
source /tmp/one_fn.sh
one r4i_lone r04_i_lone k_gmm_step "k_gmm_step<3, true, 512, true>" gmm 1000000 "python3 bench.py --batch 1 --steps 16 --warmup 4 (one run per call: the lone launch form at the end of round 4 -- requests in one round trip, early cull, advancing waves prioritised; numerics v9)" --batch 1 --steps 16 --warmup 4
cp profiles/traffic.json gpurun_out/traffic_r04i.json
unset POCS_SKIP_SINGLE POCS_BENCH_TARGET_S
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_i_bench_steps20.json 2>/dev/null &&
python -c "
import json; d=json.load(open('gpurun_out/r04_i_bench_steps20.json')); r=d['roofline']
print('value %.4g ms/step %.4f period %.1f frac %s single %s' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['frac'], d.get('single_call_evals_per_s')))"
