#!/bin/bash
# Sweep of the slices per run (POCS_SLICES) at a few batch sizes; POCS_PERSISTENT=0/1 chooses the kernel.
# usage: POCS_PERSISTENT=0 tools/slices_sweep.sh out.txt
out=${1:-gpurun_out/slices.txt}; : > $out
export POCS_PERSISTENT_MIN_RUNS=1
one() { echo "## $1 slices $2" >> $out; POCS_SLICES=$2 python bench.py $1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.4g frac %.3f kernel %s' % (d['value'], d['roofline']['frac'], d['roofline']['kernel']))" >> $out; }
for s in 12 25 26 38 39 51 52 77; do one "--steps 20 --warmup 5" $s; done
for s in 4 8 12 16 24; do one "--steps 64 --warmup 64" $s; done
for s in 32 64 96 128; do one "--batch 8 --steps 32 --warmup 8" $s; done
for s in 128 256 512; do one "--batch 2 --steps 8 --warmup 4" $s; done
for s in 256 512 977; do one "--batch 1 --steps 8 --warmup 4" $s; done
cat $out
