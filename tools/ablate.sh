#!/bin/bash
# Diagnostic builds of libpocs.so (csrc/pocs_tuning.h, -DPOCS_TUNING: NOT the shipped library) into ab_build/, run on
# the GPU box with POCS_LIB=ab_build/libpocs_<variant>.so python bench.py ...
#   timing-only ablations (wrong outputs, same launch structure: which part of k_gmm_step the time goes to) and the
#   phase-stamp build (pocs_destroy prints per-block phase times).   usage: tools/ablate.sh [stamps|ablate|graphdiag|all]
set -e
cd "$(dirname "$0")/.."
S=probability-of-collision-for-safe-planning_amd/csrc
mkdir -p ab_build
what=${1:-all}
cc() { hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value -DPOCS_TUNING "$@" $S/pocs_kernels.hip $S/pocs_host.hip; }
if [ $what = stamps ] || [ $what = all ]; then
  cc -DPOCS_STAMPS -o ab_build/libpocs_stamps.so 2>/dev/null; echo built ab_build/libpocs_stamps.so
fi
if [ $what = ablate ] || [ $what = all ]; then
  for v in RNG BOXMULLER PHILOX COLLIDE MOMENTS RNG_COLLIDE RNG_COLLIDE_MOMENTS; do
    flags=""; for part in ${v//_/ }; do flags="$flags -DPOCS_ABLATE_$part"; done
    cc $flags -o ab_build/libpocs_$v.so 2>/dev/null; echo built ab_build/libpocs_$v.so
  done
fi
if [ $what = graphdiag ] || [ $what = all ]; then
  # which of round 3's two changes cures the lost graph replay: the graph's shape or the getters' staging
  cc -DPOCS_GRAPH_WITH_COPIES -o ab_build/libpocs_graphcopies.so 2>/dev/null; echo built ab_build/libpocs_graphcopies.so
  cc -DPOCS_PAGEABLE_GETTERS -o ab_build/libpocs_pageable.so 2>/dev/null; echo built ab_build/libpocs_pageable.so
  cc -DPOCS_GRAPH_WITH_COPIES -DPOCS_PAGEABLE_GETTERS -o ab_build/libpocs_both.so 2>/dev/null; echo built ab_build/libpocs_both.so
fi
