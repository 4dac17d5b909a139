#!/bin/bash
# Diagnostic build of libpocs.so (-DPOCS_STEP_STAMPS): every block of every k_gmm_step launch stamps the
# wall clock at entry, body start, body end, after its ticket, after close_sums and after the advance;
# the host prints the per-waypoint means after each whole-run call.  Never timed as a product build.
#   POCS_LIB=ablate_build/libpocs_stepstamps.so POCS_NO_GRAPH=1 POCS_SKIP_SINGLE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
set -e
cd "$(dirname "$0")/.."
S=probability-of-collision-for-safe-planning_amd/csrc
mkdir -p ablate_build
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value \
  -DPOCS_STEP_STAMPS "$@" $S/pocs_kernels.hip $S/pocs_host.hip -o ablate_build/libpocs_stepstamps.so
echo built ablate_build/libpocs_stepstamps.so
