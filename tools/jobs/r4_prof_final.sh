# the round's FINAL binary under the profiler (heads / closers with their requests in one round trip): the driver's invocation, the same as ONE launch
# per waypoint (trace duration = span period there), one run per call, MC past the Infinity Cache; then the bench lines of that box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export POCS_SKIP_SINGLE=1 POCS_BENCH_TARGET_S=0.3
sed -n '/^one()/,/^}/p' tools/jobs/r4_prof.sh > /tmp/one_fn.sh
source /tmp/one_fn.sh
one r4h_d20 r04_h_driver20 k_gmm_step "k_gmm_step<3, true" gmm 10000000 "python3 bench.py --steps 20 --warmup 5 (the driver's invocation: 20 runs x 10^6 samples per waypoint, K=3, issued as TWO launches of 10 runs side by side; numerics v9, the round's final binary)" --steps 20 --warmup 5
POCS_SUB_BATCHES=1 one r4h_d20s r04_h_driver20_single k_gmm_step "k_gmm_step<3, true" gmm 20000000 "POCS_SUB_BATCHES=1 python3 bench.py --steps 20 --warmup 5 (20 runs x 10^6 samples per waypoint, K=3, ONE launch per waypoint; numerics v9, the round's final binary)" --steps 20 --warmup 5
one r4h_lone r04_h_lone k_gmm_step "k_gmm_step<3, true, 512, true" gmm 1000000 "python3 bench.py --batch 1 --steps 16 --warmup 4 (one run per call: the lone launch form, its head's requests in one round trip; numerics v9, the round's final binary)" --batch 1 --steps 16 --warmup 4
one r4h_mcnt r04_h_mc_nt k_mc_step "k_mc_step<true>" mc 16000000 "python3 bench.py --workload mc --batch 16 --steps 32 --warmup 16 (k_mc_step<NT>: 16 roll-out batches x 10^6 particles = 448 MB of state per launch, past the Infinity Cache; the round's final binary)" --workload mc --batch 16 --steps 32 --warmup 16
cp profiles/traffic.json gpurun_out/traffic_r04h.json
unset POCS_SKIP_SINGLE POCS_BENCH_TARGET_S
bash tools/jobs/final_bench.sh r04h 2>&1 | tail -30
