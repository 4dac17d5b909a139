# Per-block phase stamps of k_gmm_step (a -DPOCS_STAMPS build of the library, ablate_build/libpocs_stamps.so:
#   hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -DPOCS_STAMPS csrc/pocs_kernels.hip csrc/pocs_host.hip -o ...)
# at one, 20 and 64 runs per launch: where a block's time goes, how far apart blocks finish within a launch.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for args in "--batch 1 --steps 16 --warmup 4" "--steps 20 --warmup 5" "--steps 64 --warmup 64"; do
  echo "== $args"
  POCS_LIB=ablate_build/libpocs_stamps.so POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 python bench.py $args --no-cpu-baseline 2>&1 >/dev/null | grep stamps | grep -v "second time" | tail -4
done
