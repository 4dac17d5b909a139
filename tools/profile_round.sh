#!/bin/bash
# Profiles of the default bench on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh TAG [bench args...]
# writes gpurun_out/prof_TAG (kernel trace + stats) and gpurun_out/pmc_TAG_{write,fetch,sq,sq2}
# (separate --pmc passes, never combined with a trace), then prints the per-dispatch means.
# Copy what is to be judged into profiles/.
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 bench.py --no-cpu-baseline "$@" > $O/prof_$TAG.log 2>&1 || { echo "kernel trace failed"; tail -5 $O/prof_$TAG.log; exit 1; }
tail -1 $O/prof_$TAG.log | cut -c1-400
head -4 "$(find $O/prof_$TAG -name '*kernel_stats.csv' | head -1)"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_write -- python3 bench.py --no-cpu-baseline "$@" > $O/pmc_${TAG}_w.log 2>&1 || { echo "WRITE_SIZE pass failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_fetch -- python3 bench.py --no-cpu-baseline "$@" > $O/pmc_${TAG}_f.log 2>&1 || { echo "FETCH_SIZE pass failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_${TAG}_sq -- python3 bench.py --no-cpu-baseline "$@" > $O/pmc_${TAG}_s.log 2>&1 || { echo "SQ pass failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_${TAG}_sq2 -- python3 bench.py --no-cpu-baseline "$@" > $O/pmc_${TAG}_s2.log 2>&1 || { echo "SQ pass 2 failed"; exit 1; }
K=${POCS_PROFILE_KERNEL:-k_gmm_step}
python3 tools/pmc_summary.py $O/pmc_${TAG}_write $O/pmc_${TAG}_fetch $O/pmc_${TAG}_sq $O/pmc_${TAG}_sq2 --kernel $K | tee $O/pmc_${TAG}_summary.txt
