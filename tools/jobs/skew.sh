cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for sk in 500 530 550 570 600; do
  for args in "--steps 20 --warmup 5" "--steps 64 --warmup 64"; do
    POCS_GMM_SKEW=$sk POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('skew $sk $args: value %.4g kernel %.1f us P %s' % (d['value'], d['roofline']['avg_kernel_us'], d['config'].get('probability')))"
  done
done
done
