#!/bin/bash
# Shader clock and socket power while bench.py runs (rocm-smi polled from the side): is the GMM kernel
# held back by the power limit?   usage: tools/clock_probe.sh [libpocs variant .so] -- on the GPU box
# (run from the repository root; output on stdout)
lib=$1
[ -n "$lib" ] && export POCS_LIB=$lib
export POCS_SKIP_SINGLE=1
python bench.py --steps ${STEPS:-32768} --warmup 64 --no-cpu-baseline > /tmp/clock_probe_bench.json 2>/dev/null &
pid=$!
sleep 3
for i in $(seq 1 40); do
  kill -0 $pid 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e 's/.*: //' | tr '\n' ' '
  echo
  sleep 0.5
done
wait $pid
python -c "import json; d=json.load(open('/tmp/clock_probe_bench.json')); r=d['roofline']; print('[${lib:-default}] value %.4g frac %.3f %.1f us' % (d['value'], r['frac'], r['avg_kernel_us']))"
