# round 4 profiles (tools/profile_round.sh: kernel trace + stats, then every --pmc group in a pass of its own).  The raw
# csv files (tens of MB) stay on the box: what comes back is, per shape, NAME_kernel_stats.csv, NAME_pmc.txt (per-dispatch
# means by launch shape + the bench line of the trace pass + launch gaps) and one traffic record -- the files under profiles/.
#   usage: r4_prof.sh [gmm|all]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export POCS_SKIP_SINGLE=1 POCS_BENCH_TARGET_S=0.3
one() {   # TAG NAME KERNEL_KEY TRAFFIC_KEY PATH EVALS "description" bench args...
  TAG=$1; NAME=$2; KK=$3; TK=$4; PTH=$5; EV=$6; DESC=$7; shift 7
  POCS_PROFILE_KERNEL=$KK bash tools/profile_round.sh $TAG "$@" > gpurun_out/${TAG}_round.txt 2>&1 || { tail -5 gpurun_out/${TAG}_round.txt; return 1; }
  O=gpurun_out
  cp "$(find $O/prof_$TAG -name '*kernel_stats.csv' | head -1)" $O/${NAME}_kernel_stats.csv
  {
    echo "# rocprofv3 on MI355X, $DESC"
    echo "# one pass per counter group (tools/profile_round.sh); means per dispatch by launch shape (tools/pmc_summary.py)"
    python3 tools/pmc_summary.py $O/prof_$TAG $O/pmc_${TAG}_write $O/pmc_${TAG}_fetch $O/pmc_${TAG}_sq $O/pmc_${TAG}_sq2 --kernel $KK
    echo "# launch durations and gaps inside the trace pass (tools/launch_gaps.py; sub-batches run side by side: tools/overlap.py):"
    python3 tools/launch_gaps.py $O/prof_$TAG $KK
    python3 tools/overlap.py $O/prof_$TAG $KK 2>/dev/null | tail -3
    echo "# the bench line of the kernel-trace pass (same process as the trace):"
    grep -h '"metric"' $O/prof_$TAG.log | tail -1
  } > $O/${NAME}_pmc.txt
  python3 tools/traffic_from_profile.py $TAG "$TK" $NAME $PTH $EV "$DESC" 2>&1 | tail -1
  rm -rf $O/prof_$TAG $O/pmc_${TAG}_write $O/pmc_${TAG}_fetch $O/pmc_${TAG}_sq $O/pmc_${TAG}_sq2
  head -3 $O/${NAME}_kernel_stats.csv | cut -c1-150
}
one r4d_d20 r04_d_driver20 k_gmm_step "k_gmm_step<3, true" gmm 10000000 "python3 bench.py --steps 20 --warmup 5 (the driver's invocation: 20 runs x 10^6 samples per waypoint, K=3, issued as TWO launches of 10 runs side by side; numerics v9, end of round 4)" --steps 20 --warmup 5
one r4d_b64 r04_d_batch64 k_gmm_step "k_gmm_step<3, true" gmm 32000000 "python3 bench.py --steps 256 --warmup 64 (the default: 64 runs x 10^6 samples per waypoint, K=3, TWO launches of 32 runs side by side; numerics v9)" --steps 256 --warmup 64
one r4d_cfg3 r04_d_cfg3 k_gmm_step "k_gmm_step<8, true" gmm 80000000 "python3 bench.py --workload cfg3 --steps 16 --warmup 16 (16 runs x 10^7 samples per waypoint, K=8, 500 waypoints, TWO launches of 8 runs side by side; numerics v9)" --workload cfg3 --steps 16 --warmup 16
# the same kernel with ONE launch of 20 runs per waypoint (POCS_SUB_BATCHES=1): the form in which a kernel trace's duration and the bench line's
# span-timed period are the same quantity (with two sub-batches side by side the profiler keeps the twins from overlapping as they do alone)
POCS_SUB_BATCHES=1 one r4e_d20 r04_e_driver20_single k_gmm_step "k_gmm_step<3, true" gmm 20000000 "POCS_SUB_BATCHES=1 python3 bench.py --steps 20 --warmup 5 (20 runs x 10^6 samples per waypoint, K=3, ONE launch per waypoint; numerics v9, the round's final kernel)" --steps 20 --warmup 5
# one run per call: the lone form (every block closes the previous waypoint in its head and draws the first iterations' normals ahead)
one r4f_lone r04_f_lone k_gmm_step "k_gmm_step<3, true, 512, true" gmm 1000000 "python3 bench.py --batch 1 --steps 16 --warmup 4 (one run per call, the lone launch form with the normals drawn ahead; numerics v9, end of round 4)" --batch 1 --steps 16 --warmup 4
cp profiles/traffic.json gpurun_out/traffic_r04.json
