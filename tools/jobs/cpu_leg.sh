cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 300 python -m pytest tests/test_oracle_vs_ref_loop.py tests/test_oracle_vs_ref_ekf.py -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_steps20warmup5_b.json 2>gpurun_out/r03_bench_steps20_b.err
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_steps20warmup5_b.json')); print(d['value'], d['roofline']['frac']); print(json.dumps(d['cpu_baseline'], indent=1))"
