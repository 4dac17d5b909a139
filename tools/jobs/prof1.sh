cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export POCS_SKIP_SINGLE=1
bash tools/profile_round.sh r3c_d20 --steps 20 --warmup 5 2>&1 | tail -30
bash tools/profile_round.sh r3c_b64 --steps 256 --warmup 64 2>&1 | tail -30
bash tools/profile_round.sh r3c_lone --batch 1 --steps 16 --warmup 4 2>&1 | tail -30
