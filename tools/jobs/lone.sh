cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_driver.py tests/test_env_scenes.py -m gpu -x -q > gpurun_out/lone_tests.txt 2>&1 || { tail -30 gpurun_out/lone_tests.txt; exit 1; }
tail -3 gpurun_out/lone_tests.txt
for l in 1 0 1 0; do
  POCS_LONE=$l POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py --batch 1 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lone $l: value %.4g ms/step %.4f kernel %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us']))"
done
