#!/bin/bash
# A/B on ONE box (boxes differ by +-4 %): bench lines of ab_build/libpocs_base.so (build it from the
# commit to compare with: tools/ab.sh --build-base) and of the in-tree library, alternating.
#   usage on the GPU box: tools/ab.sh [bench args...]      default: --steps 64 --warmup 64
set -e
cd "$(dirname "$0")/.."
S=probability-of-collision-for-safe-planning_amd/csrc
if [ "$1" = "--build-base" ]; then
  rev=${2:-HEAD}
  tmp=$(mktemp -d)
  git archive $rev $S include | tar -x -C $tmp
  mkdir -p ab_build
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value \
    $tmp/$S/pocs_kernels.hip $tmp/$S/pocs_host.hip -o ab_build/libpocs_base.so
  rm -rf $tmp
  echo "built ab_build/libpocs_base.so from $rev"
  exit 0
fi
args=${@:---steps 64 --warmup 64}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g frac %.3f %.1f us' % (d['value'], r['frac'], r['avg_kernel_us']))"; }
for i in 1 2 3; do
  POCS_LIB=ab_build/libpocs_base.so POCS_SKIP_SINGLE=1 python bench.py $args --no-cpu-baseline 2>/dev/null | line base
  POCS_SKIP_SINGLE=1 python bench.py $args --no-cpu-baseline 2>/dev/null | line new
done
