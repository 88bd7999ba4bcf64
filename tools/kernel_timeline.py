#!/usr/bin/env python3
"""Prints the kernels of the last frame of a rocprofv3 --kernel-trace CSV: start, duration, name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
begins = [i for i, r in enumerate(rows) if "k_chunk_begin" in r["Kernel_Name"]]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sel = rows[begins[-n_last]:] if begins else rows[-40:]
base = int(sel[0]["Start_Timestamp"])
for r in sel:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    s, e = int(r["Start_Timestamp"]) - base, int(r["End_Timestamp"]) - base
    print("%9.1f %8.1f us  %s" % (s / 1e3, (e - s) / 1e3, name))
