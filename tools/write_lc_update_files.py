#!/usr/bin/env python3
"""tests/golden/lc_updates.json -> tmp_fixtures/u633.json, u634.json (one update per file, the layout of the reference's
src/light_client_update_period_63{3,4}.json), the inputs of examples/lc_prover:
    python tools/write_lc_update_files.py && LCP2_PROF=1 ./examples/lc_prover tmp_fixtures/u633.json tmp_fixtures/u634.json --repeat 3"""
import json
import os

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lc = json.load(open(os.path.join(root, "tests", "golden", "lc_updates.json")))
out = os.path.join(root, "tmp_fixtures")
os.makedirs(out, exist_ok=True)
for tag, u in lc.items():
    json.dump(u, open(os.path.join(out, f"u{tag}.json"), "w"))
print("wrote", sorted(os.listdir(out)))
