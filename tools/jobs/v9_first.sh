# numerics v9, first contact: the whole GPU suite (bit parity with the restated oracle), then v8 (ab_build/libpocs_base.so,
# the previous commit) against v9 on ONE box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/v9_gputests.txt 2>&1; rc=$?; tail -5 gpurun_out/v9_gputests.txt; [ $rc = 0 ] || exit $rc
bash tools/jobs/ab_pair.sh base 3 > gpurun_out/v9_ab.txt 2>&1; cat gpurun_out/v9_ab.txt
