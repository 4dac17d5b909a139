# What fewer, longer lane chains would buy and cost (TIMING ONLY: other slice counts are other numerics): copies of the
# sources with POCS_GMM_MAX_VS = 128 / 64 built on the box, against the library as shipped (256), ONE box, alternating.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out ab_build
S=probability-of-collision-for-safe-planning_amd/csrc
for vs in 128 64; do
  rm -rf /tmp/vs$vs; mkdir -p /tmp/vs$vs/$S /tmp/vs$vs/include; cp $S/*.h $S/*.hpp $S/*.hip /tmp/vs$vs/$S/; cp include/*.h /tmp/vs$vs/include/
  sed -i "s/#define POCS_GMM_MAX_VS 256/#define POCS_GMM_MAX_VS $vs/; s/POCS_GMM_MAX_VS == 256/POCS_GMM_MAX_VS == $vs/" /tmp/vs$vs/$S/pocs_kernels.h
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value /tmp/vs$vs/$S/pocs_kernels.hip /tmp/vs$vs/$S/pocs_host.hip -o ab_build/libpocs_vs$vs.so 2>/dev/null || { echo "build vs$vs failed"; exit 1; }
done
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: value %.4g ms/step %.4f period %.1f us frac %.3f' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['frac']))"; }
{
for rep in 1 2; do
for v in shipped vs128 vs64; do
  lib=ab_build/libpocs_$v.so; [ $v = shipped ] && lib=""
  for args in "--steps 20 --warmup 5" "--steps 128 --warmup 64" "--batch 1 --steps 16 --warmup 4" "--batch 8 --steps 32 --warmup 8"; do
    POCS_LIB=$lib POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 POCS_BENCH_TARGET_S=0.8 timeout -k 10 200 python bench.py $args --no-cpu-baseline 2>/dev/null | line "$v [$args]"
  done
done
done
} > gpurun_out/r04_vs_sweep.txt 2>&1
cat gpurun_out/r04_vs_sweep.txt
