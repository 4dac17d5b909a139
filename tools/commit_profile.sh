#!/bin/bash
# After tools/profile_round.sh TAG ... on the GPU box: compose the two files that get committed,
#   profiles/NAME_kernel_stats.csv  (rocprofv3 --stats of the kernel-trace pass)
#   profiles/NAME_pmc.txt           (per-dispatch means by launch shape + the bench line of the trace pass)
# usage (build container, after gpurun merged gpurun_out/): tools/commit_profile.sh TAG NAME "description"
set -e
cd "$(dirname "$0")/.."
TAG=$1; NAME=$2; DESC=$3
O=gpurun_out
cp "$(find $O/prof_$TAG -name '*kernel_stats.csv' | head -1)" profiles/${NAME}_kernel_stats.csv
{
  echo "# rocprofv3 on MI355X, $DESC"
  echo "# one pass per counter group (tools/profile_round.sh); means per dispatch by launch shape (tools/pmc_summary.py)"
  python3 tools/pmc_summary.py $O/prof_$TAG $O/pmc_${TAG}_write $O/pmc_${TAG}_fetch $O/pmc_${TAG}_sq $O/pmc_${TAG}_sq2 --kernel ${POCS_PROFILE_KERNEL:-k_gmm_step}
  echo "# the bench line of the kernel-trace pass (hipEvents inside bench.py, same process as the trace):"
  grep -h '"metric"' $O/prof_$TAG.log | tail -1
} > profiles/${NAME}_pmc.txt
echo "wrote profiles/${NAME}_kernel_stats.csv profiles/${NAME}_pmc.txt"
