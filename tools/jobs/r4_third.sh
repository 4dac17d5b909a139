# round 4, third job: which change cures the lost graph replay; the bench with span timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in graphcopies pageable both; do
  POCS_LIB=ab_build/libpocs_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "graph_replays_survive_readbacks" -p no:cacheprovider > gpurun_out/r04_graphdiag_$v.txt 2>&1
  echo "graph diagnosis, $v: rc $? : $(tail -1 gpurun_out/r04_graphdiag_$v.txt)"
  grep -h "a pointer baked" gpurun_out/r04_graphdiag_$v.txt | head -2
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "graph_replays_survive_readbacks" -p no:cacheprovider 2>&1 | tail -1
for args in "--steps 20 --warmup 5" "" "--batch 1 --steps 16 --warmup 4"; do
  POCS_SKIP_SINGLE=1 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; t=d['timing']
print('$args: value %.4g ms/step %.4f [%.4f, %.4f] x%d  period %.1f us bracketed %.1f frac %.3f nostore %.1f' % (d['value'], d['ms_per_step'], t['ms_per_step_min'], t['ms_per_step_max'], t['repeats'], r['avg_kernel_us'], r['bracketed_kernel_us'], r['frac'], r['limiter']['kernel_us_without_sample_stores'] or 0))"
done
