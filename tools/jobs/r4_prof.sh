# round 4 profiles (tools/profile_round.sh: kernel trace + stats, then every --pmc group in a pass of its own).  The raw
# csv files (tens of MB) stay on the box: what comes back is, per shape, NAME_kernel_stats.csv, NAME_pmc.txt (per-dispatch
# means by launch shape + the bench line of the trace pass + launch gaps) and one traffic record -- the files under profiles/.
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
    echo "# launch durations and gaps inside the trace pass (tools/launch_gaps.py):"
    python3 tools/launch_gaps.py $O/prof_$TAG $KK
    echo "# the bench line of the kernel-trace pass (same process as the trace):"
    grep -h '"metric"' $O/prof_$TAG.log | tail -1
  } > $O/${NAME}_pmc.txt
  python3 tools/traffic_from_profile.py $TAG "$TK" $NAME $PTH $EV "$DESC" 2>&1 | tail -1
  rm -rf $O/prof_$TAG $O/pmc_${TAG}_write $O/pmc_${TAG}_fetch $O/pmc_${TAG}_sq $O/pmc_${TAG}_sq2
  head -3 $O/${NAME}_kernel_stats.csv | cut -c1-150
}
cp profiles/traffic.json gpurun_out/traffic_before.json
one r4b_d20 r04_b_driver20 k_gmm_step "k_gmm_step<3, true" gmm 20000000 "python3 bench.py --steps 20 --warmup 5 (the driver's invocation: 20 runs x 10^6 samples per launch, K=3; numerics v8, end of round 4)" --steps 20 --warmup 5
one r4b_b64 r04_b_batch64 k_gmm_step "k_gmm_step<3, true" gmm 64000000 "python3 bench.py --steps 256 --warmup 64 (the default: 64 runs x 10^6 samples per launch, K=3; numerics v8)" --steps 256 --warmup 64
one r4b_lone r04_b_lone k_gmm_step "k_gmm_step<3, true" gmm 1000000 "python3 bench.py --batch 1 --steps 16 --warmup 4 (one run per call, the lone launch form; numerics v8)" --batch 1 --steps 16 --warmup 4
one r4b_cfg3 r04_b_cfg3 k_gmm_step "k_gmm_step<8, true" gmm 160000000 "python3 bench.py --workload cfg3 --steps 16 --warmup 16 (16 runs x 10^7 samples per launch, K=8, 500 waypoints; numerics v8)" --workload cfg3 --steps 16 --warmup 16
one r4b_mc_nt r04_b_mc_nt k_mc_step "k_mc_step<true>" mc 16000000 "python3 bench.py --workload mc --batch 16 --steps 32 --warmup 16 (k_mc_step<NT>: 16 roll-out batches x 10^6 particles = 448 MB of state per launch, past the Infinity Cache)" --workload mc --batch 16 --steps 32 --warmup 16
one r4b_mc r04_b_mc k_mc_step "k_mc_step<false>" mc 8000000 "python3 bench.py --workload mc (8 roll-out batches x 10^6 particles = 224 MB of state per launch, resident in the Infinity Cache)" --workload mc
one r4b_cfg5 r04_b_cfg5_mc k_mc_step "k_mc_step<false>" mc 6400000 "python3 bench.py --workload cfg5 (64 roll-out batches x 10^5 particles, 500 waypoints, resident in the Infinity Cache)" --workload cfg5
cp profiles/traffic.json gpurun_out/traffic_r04.json
