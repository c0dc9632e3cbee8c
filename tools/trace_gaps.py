#!/usr/bin/env python3
"""Idle time of the GPU inside a proof, from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`): per proof (delimited by
k_pow_search and the query gathers that follow it) the span from the first to the last kernel, the sum of the kernel durations and the
difference = what the host costs the device (Fiat-Shamir round trips, staging copies, launch latency).
    python tools/trace_gaps.py <kernel_trace.csv> [out.json]"""
import csv
import json
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("lcp2::", "")) for r in rows)
pows = [i for i, e in enumerate(ev) if e[2].startswith("k_pow_search")]
starts = []
for pi in pows:
    j = pi + 1
    while j < len(ev) and (ev[j][2].startswith("k_gather") or ev[j][2].startswith("__amd_rocclr_copyBuffer")):  # the query answers and their copy back end a proof
        j += 1
    starts.append(j)
out = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = ev[a:b]
    busy = sum(e[1] - e[0] for e in seg)
    span = seg[-1][1] - seg[0][0]
    gaps = sorted(((seg[j + 1][0] - seg[j][1]) / 1e3, seg[j][2][:36], seg[j + 1][2][:36]) for j in range(len(seg) - 1))[::-1]
    out.append({"span_ms": span / 1e6, "kernels_ms": busy / 1e6, "idle_ms": (span - busy) / 1e6, "kernels": b - a,
                "gaps_over_20us": sum(1 for g in gaps if g[0] > 20), "largest_gaps_us": [[round(g[0], 1), g[1], g[2]] for g in gaps[:8]]})
res = {"_source": "tools/trace_gaps.py on a rocprofv3 --kernel-trace csv; proofs after the first (warm-up) one", "proofs": out}
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
