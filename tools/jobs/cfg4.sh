cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -8
for a in "--steps 20 --warmup 5" ""; do python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['avg_kernel_us'], {k: r['limiter'][k] for k in ('kernel_us_without_sample_stores','stream_alone_us_at_fill_rate','valu_issue_frac','clock_MHz')}, r['frac'], r['frac_of_fill'])"; done
