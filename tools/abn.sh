#!/bin/bash
# A/B/C... on ONE box: tools/abn.sh "bench args" lib1.so lib2.so ... ("" = the in-tree library); two rounds
args=$1; shift
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s value %.4g frac %.3f %.1f us' % ('$1', d['value'], r['frac'], r['avg_kernel_us']))"; }
for i in 1 2; do for l in "$@"; do
  POCS_LIB=$l POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py $args --no-cpu-baseline 2>/dev/null | line "${l:-in-tree}"
done; done
