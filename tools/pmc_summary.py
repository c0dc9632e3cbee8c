#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv: counters summed over the launches of each kernel, plus per-wave
figures (waves = Grid_Size / 64).  python tools/pmc_summary.py <counter_collection.csv> [name filter ...]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
waves = collections.defaultdict(float)
seen = set()
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if flt and not any(f in k for f in flt):
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k))
        waves[k] += float(r["Grid_Size"]) / 64
        agg[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        agg[k]["_launches"] += 1
        agg[k]["_vgpr"] = float(r["VGPR_Count"])
        agg[k]["_lds"] = float(r["LDS_Block_Size"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"]):
    w = max(waves[k], 1)
    print("%-40s launches %3d  %9.3f ms  waves %9d  vgpr %3d lds %6d" % (k[-40:], v["_launches"], v["_ns"] / 1e6, w, v["_vgpr"], v["_lds"]))
    print("    per wave: " + "  ".join("%s %.0f" % (c.replace("SQ_", ""), x / w) for c, x in sorted(v.items()) if not c.startswith("_")))
