cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_driver.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -12 || exit 1
timeout -k 10 900 python tools/table1.py > gpurun_out/r04_table1_with_reference_loop.txt 2>gpurun_out/table1.err || { tail -5 gpurun_out/table1.err; exit 1; }
cat gpurun_out/r04_table1_with_reference_loop.txt
