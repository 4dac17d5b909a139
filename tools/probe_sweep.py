#!/usr/bin/env python3
"""The device's table-driven sampler functions against the oracle on many words (pocs_probe_device_math): N random
radius / angle words and headings per round, plus every radius word of the form 2^e + j 2^(e-12) (all 4096 twelve-bit
prefixes of every octave: each of the 512 cells eight times) -- a one-off sweep beyond tests/test_gpu_parity.py.
usage (GPU box): python3 tools/probe_sweep.py [ROUNDS [N]]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import oracle  # noqa: E402
import pocs_amd  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
orc = oracle.Oracle()
ctx = pocs_amd.Context(0)
rng = np.random.default_rng(20261005)
bad = 0
total = 0
t0 = time.time()


def check(wr, wa, x):
    global bad, total
    z0, z1, sn, cs, r2 = ctx.probe_device_math(wr, wa, x)
    for i in range(len(wr)):
        w = orc.normal_pair_w2(int(wr[i]), int(wa[i]))
        s = orc.sincos_tab(float(x[i]))
        if (z0[i], z1[i]) != w or (sn[i], cs[i]) != s or r2[i] != orc.radius2_unit32(int(wr[i])):
            bad += 1
            if bad <= 10:
                print("MISMATCH", int(wr[i]), int(wa[i]), float(x[i]), (z0[i], z1[i]), w, (sn[i], cs[i]), s)
    total += len(wr)


pre = np.array([(1 << e) + (j << max(e - 12, 0)) for e in range(0, 32) for j in range(4096 if e >= 12 else 1 << e)], dtype=np.uint64)
pre = np.unique(pre & 0xFFFFFFFF).astype(np.uint32)
check(pre, rng.integers(0, 2 ** 32, len(pre), dtype=np.uint64).astype(np.uint32), rng.uniform(-40, 40, len(pre)))
print("prefix words: %d, mismatches %d, %.0f s" % (len(pre), bad, time.time() - t0), flush=True)
for r in range(rounds):
    wr = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    wa = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    x = np.where(rng.random(n) < 0.9, rng.uniform(-12, 12, n), rng.uniform(-1e5, 1e5, n))
    check(wr, wa, x)
    print("round %d: %d triples so far, mismatches %d, %.0f s" % (r, total, bad, time.time() - t0), flush=True)
print("probe sweep: %d triples, %d mismatches" % (total, bad))
sys.exit(1 if bad else 0)
