cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
sed -n '/^one()/,/^}/p' tools/jobs/r4_prof.sh > /tmp/one_fn.sh
export POCS_SKIP_SINGLE=1 POCS_BENCH_TARGET_S=0.3
source /tmp/one_fn.sh
POCS_SUB_BATCHES=1 one r4e_d20 r04_e_driver20_single k_gmm_step "k_gmm_step<3, true" gmm 20000000 "POCS_SUB_BATCHES=1 python3 bench.py --steps 20 --warmup 5 (20 runs x 10^6 samples per waypoint, K=3, ONE launch per waypoint; numerics v9, the round's final kernel)" --steps 20 --warmup 5
cp profiles/traffic.json gpurun_out/traffic_r04.json
