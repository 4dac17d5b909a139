cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputests.txt 2>&1
rc=$?
tail -25 gpurun_out/r03_gputests.txt
exit $rc
