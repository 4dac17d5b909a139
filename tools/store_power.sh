#!/bin/bash
# tools/ubench/store_power.hip under rocm-smi: clock and power of each store flavour (GPU box).
set -e
cd "$(dirname "$0")/.."
hipcc -O2 --offload-arch=gfx950 tools/ubench/store_power.hip -o /tmp/store_power
for v in 0 1 2 3; do
  /tmp/store_power $v 5 &
  pid=$!
  sleep 2
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done
  wait $pid
done
