# the whole GPU suite, then ab_build/libpocs_base.so (tools/ab.sh --build-base REV) and the working tree's library
# alternating at 20 / 64 / 1 runs per call on ONE box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests_a.txt 2>&1
rc=$?
tail -15 gpurun_out/r04_gputests_a.txt
[ $rc = 0 ] || exit $rc
POCS_BENCH_TARGET_S=0.5 bash tools/jobs/ab3.sh base 2>&1 | tee gpurun_out/r04_ab_a.txt
