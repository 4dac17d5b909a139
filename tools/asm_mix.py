#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing (tuning aid).
usage: asm_mix.py file.s mangled-name-substring"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith("_ZN") and key in l and l.rstrip().endswith(tuple([":"])) or (key in l and re.match(r"^\S+:\s*;", l)))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
cnt = collections.Counter()
labels = []
for l in text[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\S+:", t):
            labels.append((t.split(":")[0], sum(cnt.values())))
        continue
    cnt[t.split()[0]] += 1
tot = sum(cnt.values())
print("instructions:", tot)
groups = collections.Counter()
for op, n in cnt.items():
    g = ("f64" if "_f64" in op else "mul32" if re.search(r"mul_(hi|lo)_u32|mad_u64_u32", op) else
         "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
         "salu" if op.startswith("s_") else "valu32")
    groups[g] += n
print(dict(groups))
for op, n in cnt.most_common(40):
    print("%6d %s" % (n, op))
print("labels:", labels[:60])
