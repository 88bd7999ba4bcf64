#!/usr/bin/env python3
"""Condense gpurun_out/profiles_<tag>/ (written by tools/collect_profiles.sh) into profiles/.

usage: tools/summarise_profiles.py <tag>
Writes profiles/<tag>_kernel_stats_default.csv, profiles/<tag>_kernel_stats_streams1.csv (rocprofv3
--kernel-trace --stats of `bench.py` and `bench.py --streams 1`), profiles/<tag>_pmc.txt (the raw
per-kernel counter sums) and profiles/<tag>_summary.md (one table per kernel, single-stream frame).
"""
import collections
import csv
import glob
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
N_SIMD, CLOCK_HZ = 1024, 2.4e9


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def counters(sub):
    files = glob.glob(os.path.join(src, sub, "*counter_collection.csv"))
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    if not files:
        return out, disp
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return out, disp


for mode in ("default", "streams1"):
    f = os.path.join(src, "kt_" + mode, "kt_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, mode)))

stats = {}
f = os.path.join(src, "kt_streams1", "kt_kernel_stats.csv")
frames = 4  # bench.py default: 1 warm-up + 3 timed frames; --streams 1 adds no extra frame
for r in csv.DictReader(open(f)):
    stats[short(r["Name"])] = (int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]))

sq1, d1 = counters("pmc_sq1")
sq2, _ = counters("pmc_sq2")
fetch, _ = counters("pmc_fetch")
write, _ = counters("pmc_write")
with open(os.path.join(dst, tag + "_pmc.txt"), "w") as out:
    for title, table in (("SQ pass 1", sq1), ("SQ pass 2", sq2), ("FETCH_SIZE pass", fetch), ("WRITE_SIZE pass", write)):
        out.write("== %s (one frame, bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-frames --streams 1)\n" % title)
        for k in sorted(table):
            out.write(k + "\n")
            for c in sorted(table[k]):
                out.write("    %-28s %.6g\n" % (c, table[k][c]))

with open(os.path.join(dst, tag + "_summary.md"), "w") as out:
    out.write("# Profile summary %s: chapter11 10000x5000 depth 30, one MI355X, `--streams 1`\n\n" % tag)
    out.write("Kernel times: rocprofv3 --kernel-trace --stats over %d frames. Counters: one frame each pass.\n" % frames)
    out.write("VALU busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel time x 2.4 GHz). FETCH_SIZE / WRITE_SIZE are the\n")
    out.write("L2's memory-side request bytes as rocprofv3 reports them (KB), uncorrected: gfx950 tallies reads at half their\n")
    out.write("size, see %s_traffic.json for the bytes corrected by the factors measured on a known 8-byte-per-lane stream.\n\n" % tag)
    out.write("| kernel | launches/frame | ms/frame | avg launch us | VALU inst/frame | VALU busy | waves/SIMD (avg) | fetch GB | write GB |\n")
    out.write("|---|---|---|---|---|---|---|---|---|\n")
    for k, (calls, total, avg) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        ms = total / frames / 1e6
        valu = sq1.get(k, {}).get("SQ_INSTS_VALU", 0.0)
        wave_cycles = sq1.get(k, {}).get("SQ_WAVE_CYCLES", 0.0)
        busy = valu * 4 / (N_SIMD * ms * 1e-3 * CLOCK_HZ) if ms > 0 else 0.0
        occ = wave_cycles * 4 / (N_SIMD * ms * 1e-3 * CLOCK_HZ) if ms > 0 else 0.0
        out.write("| %s | %d | %.2f | %.1f | %.3g | %.0f%% | %.1f | %.1f | %.1f |\n" % (
            k, calls // frames, ms, avg / 1e3, valu, 100 * busy, occ,
            fetch.get(k, {}).get("FETCH_SIZE", 0.0) / 1e6, write.get(k, {}).get("WRITE_SIZE", 0.0) / 1e6))
# HBM-side traffic per kernel for bench.py's roofline.traffic: FETCH_SIZE and WRITE_SIZE are reported in KB;
# gfx950 tallies 128-byte read requests at 64 bytes, so reads are doubled when the calibration stream (8 bytes
# per lane, the record columns' width, 2 GiB per launch) confirms that factor for this access pattern.
import json

cal_f, _ = counters("calib_fetch")
cal_w, _ = counters("calib_write")
CAL_BYTES = 3 * (2 << 30)
read_factor, write_factor, cal = 2.0, 1.0, None
if "calib_read8" in {k.split("(")[0] for k in cal_f} or cal_f:
    rd = sum(v.get("FETCH_SIZE", 0.0) for k, v in cal_f.items() if "calib_read8" in k) * 1024
    wr = sum(v.get("WRITE_SIZE", 0.0) for k, v in cal_w.items() if "calib_write8" in k) * 1024
    if rd > 0 and wr > 0:
        read_factor, write_factor = CAL_BYTES / rd, CAL_BYTES / wr
        cal = {"bytes_streamed": CAL_BYTES, "FETCH_SIZE_bytes_reported": rd, "WRITE_SIZE_bytes_reported": wr,
               "read_factor": read_factor, "write_factor": write_factor}
traffic = {"command": "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-frames --streams 1 (one frame per counter pass)",
           "correction": "bytes = FETCH_SIZE KB x 1024 x %.3f (reads) and WRITE_SIZE KB x 1024 x %.3f (writes); factors %s"
                         % (read_factor, write_factor, "measured on a 2 GiB stream read and written at 8 bytes per lane (tools/calib_fetch.hip)"
                            if cal else "from the gfx950 rule for wide coalesced reads (uncalibrated for this width)"),
           "calibration": cal, "kernels": {}}
groups = collections.defaultdict(lambda: {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0})
for k in set(fetch) | set(write):
    base = k.split("<")[0]
    if not base.startswith("k_"):
        continue
    g = groups[base]
    g["launches"] += len(d1.get(k, ())) or stats.get(k, (0,))[0] // frames
    g["fetch_kb"] += fetch.get(k, {}).get("FETCH_SIZE", 0.0)
    g["write_kb"] += write.get(k, {}).get("WRITE_SIZE", 0.0)
for base, g in groups.items():
    traffic["kernels"][base] = {"launches": g["launches"], "fetch_bytes_reported": g["fetch_kb"] * 1024,
                                "fetch_bytes_corrected": g["fetch_kb"] * 1024 * read_factor,
                                "write_bytes": g["write_kb"] * 1024 * write_factor}
with open(os.path.join(dst, tag + "_traffic.json"), "w") as out:
    json.dump(traffic, out, indent=1, sort_keys=True)
print(open(os.path.join(dst, tag + "_summary.md")).read())
print(json.dumps(traffic["kernels"], indent=1))
