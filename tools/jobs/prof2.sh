cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export POCS_SKIP_SINGLE=1
bash tools/profile_round.sh r3c_cfg3 --workload cfg3 --steps 16 --warmup 16 2>&1 | tail -24
POCS_PROFILE_KERNEL=k_mc_step bash tools/profile_round.sh r3c_cfg5 --workload cfg5 --steps 128 --warmup 64 2>&1 | tail -24
POCS_PROFILE_KERNEL=k_mc_step bash tools/profile_round.sh r3c_mc --workload mc --steps 32 --warmup 16 2>&1 | tail -24
