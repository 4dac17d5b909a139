# round 4 profiles (tools/profile_round.sh: kernel trace + stats, then every --pmc group in a pass of its own)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export POCS_SKIP_SINGLE=1 POCS_BENCH_TARGET_S=0.3
bash tools/profile_round.sh r4a_d20 --steps 20 --warmup 5 2>&1 | tail -32
bash tools/profile_round.sh r4a_b64 --steps 256 --warmup 64 2>&1 | tail -32
POCS_PROFILE_KERNEL=k_mc_step bash tools/profile_round.sh r4a_mc_nt --workload mc --batch 16 --steps 32 --warmup 16 2>&1 | tail -24
