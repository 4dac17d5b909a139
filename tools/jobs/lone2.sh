cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for l in 1 0 1; do
  POCS_LONE=$l POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py --batch 1 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lone $l: value %.4g ms/step %.4f kernel %.1f us P %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['config'].get('probability')))"
done
POCS_LONE=1 POCS_LIB=ablate_build/libpocs_stamps.so POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py --batch 1 --steps 16 --warmup 4 --no-cpu-baseline 2>&1 >/dev/null | grep stamps | tail -1
