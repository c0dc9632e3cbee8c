#!/usr/bin/env python3
"""Derives per-kernel HBM traffic per launch from two rocprofv3 PMC passes of the same command
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, collected separately as MI355X_MICROARCH.md prescribes):
    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <bench log of the pass> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, so it is doubled (guide's
gfx950 correction).  The bench log of the PMC pass supplies the algorithmic bytes per launch of k_hash_leaves for the
same launch population, so that bench.py can scale the traffic to its own population."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("lcp2::", "")
        tot[name] += float(row["Counter_Value"])
        n[name] += 1
    return tot, n


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
bench = [json.loads(l) for l in open(sys.argv[3]) if l.startswith('{"metric"')][-1]
out = {"_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 1 "
                  "--warmup 0 --no-cpu-baseline`; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B), KB -> bytes x1024",
       "kernels": {}}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    if k.startswith("__amd") or nf[k] == 0:
        continue
    rd = 2.0 * fetch[k] * 1024 / nf[k] / 1e9
    wr = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1) / 1e9
    out["kernels"][k] = {"launches": nf[k], "fetch_kb_per_launch_raw": fetch[k] / nf[k], "read_GB_per_launch_corrected": rd,
                         "write_GB_per_launch": wr}
hl = out["kernels"]["k_hash_leaves"]
alg = bench["roofline"]["algorithmic_bytes_per_launch"] / 1e9
hl["algorithmic_GB_per_launch_same_population"] = alg
hl["traffic_over_algorithmic"] = (hl["read_GB_per_launch_corrected"] + hl["write_GB_per_launch"]) / alg
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["kernels"]["k_hash_leaves"], indent=1))
for k in sorted(out["kernels"]):
    if k.startswith("k_q_") or k.startswith("k_ntt"):
        print(k, out["kernels"][k])
