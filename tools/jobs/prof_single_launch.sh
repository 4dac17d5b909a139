# two extra shapes of tools/jobs/r4_prof.sh on their own: 20 runs as ONE launch per waypoint, and one run per call (the lone form)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
sed -n '/^one()/,/^}/p' tools/jobs/r4_prof.sh > /tmp/one_fn.sh
export POCS_SKIP_SINGLE=1 POCS_BENCH_TARGET_S=0.3
source /tmp/one_fn.sh
eval "$(grep '^one r4f_lone' tools/jobs/r4_prof.sh)"
[ "$1" = all ] && eval "$(grep '^POCS_SUB_BATCHES=1 one r4e_d20' tools/jobs/r4_prof.sh)"
cp profiles/traffic.json gpurun_out/traffic_r04.json
