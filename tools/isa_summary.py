#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc --save-temps .s file: usage isa_summary.py file.s <mangled-name substring>."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if key in l and re.match(r"^[A-Za-z_][^ \t]*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith((".amdhsa_kernel", ".size")))
body = lines[start:end]
hist = collections.Counter()
for l in body:
    m = re.match(r"\s+([a-z_0-9]+)\s", l + " ")
    if m and not l.strip().startswith((".", ";")):
        hist[m.group(1)] += 1
total = sum(hist.values())
print(lines[start], "instructions:", total)
groups = collections.Counter()
for op, n in hist.items():
    g = ("f64" if "_f64" in op else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_waitcnt", "s_buffer")) else
         "smem" if op.startswith(("s_load", "s_buffer")) else "wait" if op.startswith("s_waitcnt") else "lds" if op.startswith("ds_") else
         "scratch" if op.startswith("scratch_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other")
    groups[g] += n
print(dict(groups))
for op, n in hist.most_common(40):
    print(f"{n:6d} {op}")
for l in lines[end:end + 400]:
    if key in l and (".num_vgpr" in l or "private_seg_size" in l):
        print(l.strip()[:200])
