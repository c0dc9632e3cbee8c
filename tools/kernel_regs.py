#!/usr/bin/env python3
"""Register and LDS footprint of every kernel in liblcp2.so (code-object metadata): python tools/kernel_regs.py [filter]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_hazards as ch  # noqa: E402

lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eth-lc-plonky2_amd", "liblcp2.so")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, "fat.bin")
    subprocess.run([os.path.join(ch.LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    blob = open(fat, "rb").read()
    offs, at = [], blob.find(ch.MAGIC)
    while at >= 0:
        offs.append(at)
        at = blob.find(ch.MAGIC, at + 1)
    for i, a in enumerate(offs):
        b = offs[i + 1] if i + 1 < len(offs) else len(blob)
        part, co = os.path.join(d, "b%d.bin" % i), os.path.join(d, "b%d.co" % i)
        open(part, "wb").write(blob[a:b])
        subprocess.run([os.path.join(ch.LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part, "--targets=" + ch.TARGET,
                        "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(ch.LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            blk = ".agpr_count:" + blk
            get = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]  # noqa: E731
            name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip().split("(")[0]
            if flt in name:
                print("%-50s vgpr %4s agpr %3s sgpr %3s spill v%s s%s lds %6s scratch %s" % (name[-50:], get("vgpr_count"), get("agpr_count"), get("sgpr_count"),
                      get("vgpr_spill_count"), get("sgpr_spill_count"), get("group_segment_fixed_size"), get("private_segment_fixed_size")))
