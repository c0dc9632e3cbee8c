#!/usr/bin/env python3
"""Copies one round's measurement set from gpurun_out/ into profiles/ under a common tag and derives the summaries:
    python tools/install_profiles.py r01_v8 8
expects gpurun_out/{r01_vN_bench.log, r01_vN_bench_under_rocprof.log, prof_vN/vN_kernel_stats.csv, pmc_fetchN/, pmc_writeN/, pmc_valuN/}."""
import collections
import csv
import os
import shutil
import subprocess
import sys

tag, n = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g, p = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def json_line(src, dst):
    lines = [l for l in open(src) if l.startswith('{"metric"')]
    open(dst, "w").write(lines[-1])


json_line(f"{g}/{tag}_bench.log", f"{p}/{tag}_bench.json")
json_line(f"{g}/{tag}_bench_under_rocprof.log", f"{p}/{tag}_bench_under_rocprof.json")
json_line(f"{g}/pmc_fetch{n}.log", f"{p}/{tag}_bench_under_pmc_fetch.json")
shutil.copy(f"{g}/prof_v{n}/v{n}_kernel_stats.csv", f"{p}/{tag}_full_proof_kernel_stats.csv")
shutil.copy(f"{g}/pmc_fetch{n}/fetch_counter_collection.csv", f"{p}/{tag}_pmc_fetch_size_counter_collection.csv")
shutil.copy(f"{g}/pmc_write{n}/write_counter_collection.csv", f"{p}/{tag}_pmc_write_size_counter_collection.csv")
subprocess.run([sys.executable, f"{root}/tools/pmc_traffic.py", f"{p}/{tag}_pmc_fetch_size_counter_collection.csv",
                f"{p}/{tag}_pmc_write_size_counter_collection.csv", f"{g}/pmc_fetch{n}.log", f"{p}/{tag}_pmc_traffic.json"], check=True,
               stdout=subprocess.DEVNULL)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{g}/pmc_valu{n}/v_counter_collection.csv")):
    k = r["Kernel_Name"]
    if "lcp2::" in k:
        key = k.split("(")[0].replace("void ", "").replace("lcp2::", "")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[key]["dur_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(f"{p}/{tag}_valu_counters.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- python3 tools/prof_commit.py 32 22 2\n")
    f.write("# averages per launch (commitment of 32 columns at n = 2^22); GRBM_GUI_ACTIVE is summed over the 8 XCDs; clock = GRBM_GUI_ACTIVE / 8 / duration;\n")
    f.write("# cycles_per_valu_instr_per_simd = (GRBM_GUI_ACTIVE / 8) / (SQ_INSTS_VALU / 1024 SIMDs)\n")
    f.write("kernel,launches,avg_duration_ms,SQ_INSTS_VALU,GRBM_GUI_ACTIVE,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,clock_GHz,cycles_per_valu_instr_per_simd\n")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]["dur_ms"])):
        avg = lambda c: sum(v[c]) / len(v[c]) if v[c] else 0.0  # noqa: E731
        d, gui, iv = avg("dur_ms"), avg("GRBM_GUI_ACTIVE"), avg("SQ_INSTS_VALU")
        f.write("%s,%d,%.4f,%.0f,%.0f,%.0f,%.0f,%.3f,%.2f\n" % (k, len(v["SQ_INSTS_VALU"]), d, iv, gui, avg("SQ_WAVE_CYCLES"), avg("SQ_BUSY_CYCLES"),
                                                          gui / 8 / (d * 1e-3) / 1e9 if d else 0, (gui / 8) / (iv / 1024) if iv else 0))
print(open(f"{p}/{tag}_valu_counters.csv").read().split("\n")[4:7])
