#!/usr/bin/env python3
"""K6 (quotient) kernels of one workload from two rocprofv3 passes of the same command - `--kernel-trace --stats` and
`--kernel-trace --pmc FETCH_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes) - into one json: per kernel the average launch time,
HBM bytes read per launch (FETCH_SIZE is in KB and counts a 128-byte request as 64 bytes on gfx950: x 1024 x 2), and the K6 totals
against the algorithmic bytes of the quotient ((W + NC + NR + CH (1 + npp) + 2) x 8 bytes read per LDE point + 16 written).
    python tools/k6_profile_summary.py <kt_kernel_stats.csv> <fetch_counter_collection.csv> <proofs in the fetch pass> <degree_bits> <num_constants> <out.json> [trimmed_fetch.csv]"""
import collections
import csv
import json
import re
import sys

stats_csv, fetch_csv, proofs, bits, num_constants, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("lcp2::", "")


avg_ms = {short(r["Name"]): (float(r["AverageNs"]) / 1e6, int(r["Calls"])) for r in csv.DictReader(open(stats_csv))}
tot, n, vgpr = collections.defaultdict(float), collections.defaultdict(int), {}
rows = []
for r in csv.DictReader(open(fetch_csv)):
    if r["Counter_Name"] != "FETCH_SIZE" or "lcp2::" not in r["Kernel_Name"]:
        continue
    k = short(r["Kernel_Name"])
    tot[k] += float(r["Counter_Value"])
    n[k] += 1
    vgpr[k] = int(r["VGPR_Count"])
    rows.append(r)
N = 1 << (bits + 3)
CH, npp, W, NR = 2, 9, 135, 80
algorithmic = ((W + num_constants + NR + CH * (1 + npp) + 2) * 8 + CH * 8) * N / 1e9
kernels, k6_read, k6_ms = {}, 0.0, 0.0
for k in sorted(tot, key=lambda k: -tot[k]):
    if not (k.startswith("k_q_") and "true" not in k):   # the quotient pass itself (the <.., true> instances are the row check over H)
        continue
    per_launch = 2.0 * tot[k] * 1024 / n[k] / 1e9
    launches_per_proof = n[k] / proofs
    ms = avg_ms.get(k, (None, 0))[0]
    kernels[k] = {"avg_launch_ms": ms, "read_GB_per_launch": round(per_launch, 2), "launches_per_proof": launches_per_proof,
                  "read_TBps": round(per_launch / ms, 2) if ms else None}
    k6_read += per_launch * launches_per_proof
    k6_ms += (ms or 0.0) * launches_per_proof
check = {k: {"avg_launch_ms": avg_ms.get(k, (None, 0))[0], "read_GB_per_launch": round(2.0 * tot[k] * 1024 / n[k] / 1e9, 2)} for k in tot if k.startswith("k_q_") and "true" in k}
json.dump({"_source": "rocprofv3 --kernel-trace --stats and --kernel-trace --pmc FETCH_SIZE (separate passes) of the same command; FETCH_SIZE KB x 1024 x 2 (gfx950 correction)",
           "degree_bits": bits, "k6_ms_per_proof": round(k6_ms, 2), "k6_read_GB_per_proof": round(k6_read, 1), "k6_algorithmic_GB_per_proof": round(algorithmic, 1),
           "fetch_over_algorithmic": round(k6_read / algorithmic, 2), "kernels": kernels, "row_check_kernels": check}, open(out_path, "w"), indent=1)
if len(sys.argv) > 7:
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(sys.argv[7], "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keep)
        w.writeheader()
        for r in rows:
            r = {k: r[k] for k in keep}
            r["Kernel_Name"] = r["Kernel_Name"].split("(")[0]
            w.writerow(r)
print(open(out_path).read())
