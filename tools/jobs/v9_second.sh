# collision flags kept as lane masks: GPU suite, then the previous commit (ab_build/libpocs_base.so) against the tree, ONE box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/v9b_gputests.txt 2>&1; rc=$?; tail -3 gpurun_out/v9b_gputests.txt; [ $rc = 0 ] || exit $rc
bash tools/jobs/ab_pair.sh base 3 > gpurun_out/v9b_ab.txt 2>&1; cat gpurun_out/v9b_ab.txt
