#!/bin/bash
# Timing-only ablation builds of libpocs.so (wrong outputs, same launch structure): which part of
# k_gmm_step the time goes to.  Builds into ablate_build/, run on the GPU box with
#   POCS_LIB=ablate_build/libpocs_<variant>.so python bench.py --samples 16000000 ...
set -e
cd "$(dirname "$0")/.."
S=probability-of-collision-for-safe-planning_amd/csrc
mkdir -p ablate_build
for v in RNG BOXMULLER PHILOX COLLIDE MOMENTS "RNG -DPOCS_ABLATE_COLLIDE" "RNG -DPOCS_ABLATE_COLLIDE -DPOCS_ABLATE_MOMENTS"; do
  name=$(echo $v | tr -d ' ' | tr -d '-' | sed 's/DPOCS_ABLATE_/_/')
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value \
    -DPOCS_ABLATE_$v $S/pocs_kernels.hip $S/pocs_host.hip -o ablate_build/libpocs_$name.so 2>/dev/null
  echo built ablate_build/libpocs_$name.so
done
