#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel in a hipcc --save-temps .s file (optionally filtered by substring)."""
import re
import sys

t = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in t.split("  - .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    if key in name:
        print("%-90s vgpr %s sgpr %s sgpr_spill %s vgpr_spill %s scratch %s lds %s" % (
            name[:90], g("vgpr_count"), g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"),
            g("private_segment_fixed_size"), g("group_segment_fixed_size")))
