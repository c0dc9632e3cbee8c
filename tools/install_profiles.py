#!/usr/bin/env python3
"""Copies one round's measurement set (tools/collect_profiles.sh on the GPU box -> gpurun_out/<dir>) into profiles/ under a common
tag and derives the summaries:
    python tools/install_profiles.py gpurun_out/r02_final r02
-> profiles/<tag>_bench.json, _bench_under_rocprof.json, _full_proof_kernel_stats.csv, _pmc_fetch/_write counter csv,
   _pmc_traffic.json (tools/pmc_traffic.py), _sq_counters.csv (per kernel, per launch), _lc_step_kernel_stats.csv, _ubench_int_rates.txt"""
import collections
import csv
import os
import shutil
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(root, "profiles")


def json_line(a, b):
    lines = [l for l in open(a) if l.startswith('{"metric"')]
    open(b, "w").write(lines[-1])


json_line(f"{src}/bench.json", f"{p}/{tag}_bench.json")
json_line(f"{src}/bench_under_rocprof.json", f"{p}/{tag}_bench_under_rocprof.json")
json_line(f"{src}/bench_under_pmc_fetch.json", f"{p}/{tag}_bench_under_pmc_fetch.json")
shutil.copy(f"{src}/kt/kt_kernel_stats.csv", f"{p}/{tag}_full_proof_kernel_stats.csv")
shutil.copy(f"{src}/lc/lc_kernel_stats.csv", f"{p}/{tag}_lc_step_kernel_stats.csv")
if os.path.exists(f"{src}/ubench.txt"):
    shutil.copy(f"{src}/ubench.txt", f"{p}/{tag}_ubench_int_rates.txt")
if os.path.exists(f"{src}/ubench_mulchain.txt"):
    shutil.copy(f"{src}/ubench_mulchain.txt", f"{p}/{tag}_ubench_mulchain.txt")
if os.path.exists(f"{src}/sharded_rehearsal.log"):
    shutil.copy(f"{src}/sharded_rehearsal.log", f"{p}/{tag}_sharded_rehearsal.log")
if os.path.exists(f"{src}/sharded_rehearsal_chunked.log"):
    shutil.copy(f"{src}/sharded_rehearsal_chunked.log", f"{p}/{tag}_sharded_rehearsal_chunked.log")
if os.path.exists(f"{src}/bench_force_sharded.json"):
    json_line(f"{src}/bench_force_sharded.json", f"{p}/{tag}_bench_force_sharded.json")
for which in ("fetch", "write"):
    # keep the per-dispatch counter rows of the library's kernels only (the full csv is tens of MB)
    rows = list(csv.DictReader(open(f"{src}/{which}/{which}_counter_collection.csv")))
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value",
            "Start_Timestamp", "End_Timestamp"]
    with open(f"{p}/{tag}_pmc_{which}_size_counter_collection.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keep)
        w.writeheader()
        for r in rows:
            if "lcp2::" in r["Kernel_Name"]:
                r = {k: r[k] for k in keep}
                r["Kernel_Name"] = r["Kernel_Name"].split("(")[0]
                w.writerow(r)
subprocess.run([sys.executable, f"{root}/tools/pmc_traffic.py", f"{p}/{tag}_pmc_fetch_size_counter_collection.csv",
                f"{p}/{tag}_pmc_write_size_counter_collection.csv", f"{p}/{tag}_bench_under_pmc_fetch.json", f"{p}/{tag}_pmc_traffic.json"], check=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
seen = set()
for r in csv.DictReader(open(f"{src}/sq/sq_counter_collection.csv")):
    k = r["Kernel_Name"]
    if "lcp2::" not in k:
        continue
    key = k.split("(")[0].replace("void ", "").replace("lcp2::", "")
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if (r["Dispatch_Id"], key) not in seen:
        seen.add((r["Dispatch_Id"], key))
        acc[key]["dur_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        acc[key]["waves"].append(float(r["Grid_Size"]) / 64)
        acc[key]["vgpr"].append(float(r["VGPR_Count"]))
with open(f"{p}/{tag}_sq_counters.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- python3 tools/prof_prove.py 22 1\n")
    f.write("# one proof at n = 2^22 (plus build()); totals over the launches of each kernel; per_wave = total / waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs:\n")
    f.write("# clock = GRBM_GUI_ACTIVE / 8 / duration; cycles_per_valu_instr_per_simd = (GRBM_GUI_ACTIVE / 8) / (SQ_INSTS_VALU / 1024 SIMDs)\n")
    f.write("kernel,launches,total_ms,waves,vgpr,valu_per_wave,salu_per_wave,lds_per_wave,wait_any_frac_of_wave_cycles,clock_GHz,cycles_per_valu_instr_per_simd\n")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]["dur_ms"])):
        tot = lambda c: sum(v[c])  # noqa: E731
        waves, d = max(tot("waves"), 1), tot("dur_ms")
        gui, iv = tot("GRBM_GUI_ACTIVE"), tot("SQ_INSTS_VALU")
        f.write("%s,%d,%.3f,%d,%d,%.0f,%.0f,%.0f,%.3f,%.3f,%.2f\n" % (
            k, len(v["dur_ms"]), d, waves, max(v["vgpr"]), iv / waves, tot("SQ_INSTS_SALU") / waves, tot("SQ_INSTS_LDS") / waves,
            tot("SQ_WAIT_ANY") / max(tot("SQ_WAVE_CYCLES"), 1), gui / 8 / (d * 1e-3) / 1e9 if d else 0, (gui / 8) / (iv / 1024) if iv else 0))
print(open(f"{p}/{tag}_sq_counters.csv").read())
# ---- round 4 additions: the reference gate set, K6 summaries, idle time, the oracle's scaling sample
def have(path):
    return os.path.exists(path)


if have(f"{src}/mixkt/mixkt_kernel_stats.csv"):
    shutil.copy(f"{src}/mixkt/mixkt_kernel_stats.csv", f"{p}/{tag}_reference_mix_kernel_stats.csv")
    subprocess.run([sys.executable, f"{root}/tools/k6_profile_summary.py", f"{src}/mixkt/mixkt_kernel_stats.csv", f"{src}/mixfetch/mixfetch_counter_collection.csv", "2", "22", "5",
                    f"{p}/{tag}_reference_mix_k6.json", f"{p}/{tag}_reference_mix_fetch_size.csv"], check=True, stdout=subprocess.DEVNULL)
    for name in ("mix_native", "mix_interpreted"):
        if have(f"{src}/{name}.json"):
            open(f"{p}/{tag}_reference_{name}.json", "w").write([l for l in open(f"{src}/{name}.json") if l.startswith("{")][-1])
# K6 of the headline circuit (the own SHA-256 gates) from the bench's own two passes: kernel stats + FETCH_SIZE
subprocess.run([sys.executable, f"{root}/tools/k6_profile_summary.py", f"{src}/kt/kt_kernel_stats.csv", f"{src}/fetch/fetch_counter_collection.csv", "1", "22", "5",
                f"{p}/{tag}_real_gadget_k6.json"], check=True, stdout=subprocess.DEVNULL)
for name, trace in (("headline", f"{src}/kt/kt_kernel_trace.csv"), ("lc_step", f"{src}/lc/lc_kernel_trace.csv"), ("reference_mix", f"{src}/mixkt/mixkt_kernel_trace.csv")):
    if have(trace):
        subprocess.run([sys.executable, f"{root}/tools/trace_gaps.py", trace, f"{p}/{tag}_gaps_{name}.json"], check=True, stdout=subprocess.DEVNULL)
if have(f"{src}/cpu_baseline_scaling.json"):
    open(f"{p}/{tag}_cpu_baseline_scaling.json", "w").write([l for l in open(f"{src}/cpu_baseline_scaling.json") if l.startswith("{")][-1])
if have(f"{src}/lc.log"):
    shutil.copy(f"{src}/lc.log", f"{p}/{tag}_lc_step.log")
