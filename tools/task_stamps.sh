#!/bin/bash
# Diagnostic build of libpocs.so with per-wave phase stamps in k_gmm_run (-DPOCS_TASK_STAMPS): where a
# wave spends its time per task (wait for the staged task, decode + loader attempt, body prologue, sampling
# loop, flush, drain, count-in + closing, blocking staging).  Never timed as a product build.  Run with
#   POCS_LIB=ablate_build/libpocs_stamps.so POCS_NO_GRAPH=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
set -e
cd "$(dirname "$0")/.."
S=probability-of-collision-for-safe-planning_amd/csrc
mkdir -p ablate_build
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value \
  -DPOCS_TASK_STAMPS "$@" $S/pocs_kernels.hip $S/pocs_host.hip -o ablate_build/libpocs_stamps.so
echo built ablate_build/libpocs_stamps.so
