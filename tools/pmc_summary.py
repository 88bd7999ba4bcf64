"""Per-kernel totals of a rocprofv3 --pmc counter_collection.csv (development aid)."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float); disp = collections.defaultdict(set); vg = {}
seen = set()
for r in rows:
    m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
    if not m: continue
    k = m.group(0)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"]); vg[k] = (r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in agg.items():
    print(f"{k}: dispatches {len(disp[k])}, {dur[k]:.2f} ms, vgpr/sgpr/scratch {vg[k]}")
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {x:.5g}")
    w = v.get("SQ_WAVES", 0)
    if w:
        print("    per wave:", ", ".join(f"{c} {x / w:.0f}" for c, x in sorted(v.items()) if c != "SQ_WAVES"))
