#!/bin/bash
# Sweep of bench.py on one GPU: the driver's invocation (--steps 20 --warmup 5), the default (256 / 64) and small
# batches, with one and with two sub-batches per call (POCS_GMM_GROUPS).  usage: tools/bench_sweep.sh out.txt
out=${1:-gpurun_out/sweep.txt}; shift
: > "$out"
run() { echo "## $*" >> "$out"; env "$@" >> "$out" 2>> "$out.err" || echo "FAILED: $*" >> "$out"; }
for g in 1 2; do
  run POCS_GMM_GROUPS=$g python bench.py --steps 20 --warmup 5 --no-cpu-baseline
  run POCS_GMM_GROUPS=$g python bench.py --steps 256 --warmup 64 --no-cpu-baseline
  run POCS_GMM_GROUPS=$g python bench.py --batch 1 --steps 16 --warmup 4 --no-cpu-baseline
  run POCS_GMM_GROUPS=$g python bench.py --batch 8 --steps 32 --warmup 8 --no-cpu-baseline
done
python - "$out" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("##"):
        print(ln.strip())
    elif ln.startswith("{"):
        d = json.loads(ln)
        r = d["roofline"]
        print("   value %.4g evals/s  ms/step %.4f  kernel %s %.1f us/launch  frac %.3f  runs/launch %s" % (
            d["value"], d["ms_per_step"], r["kernel"], r["avg_kernel_us"], r["frac"], d["config"]["runs_per_launch"]))
PY
