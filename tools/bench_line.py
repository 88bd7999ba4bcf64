#!/usr/bin/env python3
"""Condense bench.py's JSON line (stdin) into one readable line."""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        r = d.get("roofline", {})
        o = r.get("other_kernels", {})
        print("frame %.1f ms  trace %.1f  shadow %.1f  shade %.1f  Mrays/s %.0f  frac %.4f" % (
            d["ms_per_step"], r.get("kernel_ms_per_frame", 0), o.get("k_shadow", {}).get("ms_per_frame", 0) or 0,
            o.get("k_shade", {}).get("ms_per_frame", 0), d["value"], r.get("frac", 0)))
