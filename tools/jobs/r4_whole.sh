# the sharded estimation as ONE library call (graph replay, two sub-batches, exchange in the launches' tails): tests, then
# world size 1 through it against the unsharded call, and round 3's eager form (POCS_ONEHOP=3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_multiprocess.py tests/test_bench_contract.py -x -q 2>&1 | tail -4 || exit 1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g ms/step %.4f groups %s period %.1f us frac %.3f | %s' % (d['value'], d['ms_per_step'], r['concurrent_launches'], r['avg_kernel_us'], r['frac'], d['config']['exchange'][:60]))"; }
export POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.7
for i in 1 2; do
  python bench.py --steps 20 --warmup 20 --no-cpu-baseline 2>/dev/null | line "unsharded, 20 runs"
  POCS_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2971$i RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 20 --warmup 20 --no-cpu-baseline 2>/dev/null | line "world 1 whole call, 20 runs"
  POCS_ONEHOP=3 POCS_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2972$i RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 20 --warmup 20 --no-cpu-baseline 2>/dev/null | line "world 1 eager in-tail, 20 runs"
  python bench.py --steps 64 --warmup 64 --no-cpu-baseline 2>/dev/null | line "unsharded, 64 runs"
  POCS_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2973$i RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 64 --warmup 64 --no-cpu-baseline 2>/dev/null | line "world 1 whole call, 64 runs"
done
