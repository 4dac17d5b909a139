# the whole GPU suite, then the N > 1 bench line as two ranks on ONE card produce it (rehearsal: gloo as the host channel,
# both ranks on device 0; sizes at which the two ranks' launches fit the one card side by side: DESIGN.md section 6)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests_b.txt 2>&1; rc=$?; tail -4 gpurun_out/r04_gputests_b.txt; [ $rc = 0 ] || exit $rc
POCS_FORCE_DEVICE=0 POCS_DIST_BACKEND=gloo POCS_SKIP_SINGLE=1 POCS_NO_BOARD_PROBE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 8 --warmup 4 --samples 40000 --no-cpu-baseline > gpurun_out/r04_two_ranks_one_card_line.json 2> gpurun_out/r04_two_ranks.err
echo "rc $? give-ups $(grep -c 'never arrived' gpurun_out/r04_two_ranks.err)"
python -c "
import json; d=json.load(open('gpurun_out/r04_two_ranks_one_card_line.json')); print(d['config']['exchange'][:80], '|', d['strong']['exchange'][:60], d['strong']['exchange_wait_us']['available'], d['exchange_wait_us']['median_us'])"
