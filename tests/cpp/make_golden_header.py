#!/usr/bin/env python3
"""tests/golden/sha256_kat.json -> golden_data.hpp (byte arrays for the C++ gadget tests)."""
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
kat = json.load(open(os.path.join(here, "..", "golden", "sha256_kat.json")))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "golden_data.hpp")


def arr(name, data):
    return "static const uint8_t %s[%d] = {%s};\n" % (name, len(data), ",".join(str(b) for b in data))


def arr2(name, rows):
    return "static const uint8_t %s[%d][%d] = {%s};\n" % (name, len(rows), len(rows[0]), ",".join("{" + ",".join(str(b) for b in r) + "}" for r in rows))


s = "// generated from tests/golden/sha256_kat.json (data of the reference's own tests)\n#pragma once\n#include <cstdint>\n"
for n, root in kat["zero_leaf_merkle_roots"].items():
    s += arr("ZERO_ROOT_%s" % n, bytes.fromhex(root))
sc = kat["sync_committee"]
s += arr2("SC_PUBKEYS", [bytes.fromhex(p) for p in sc["pubkeys"]])
s += arr("SC_AGG_PUBKEY", bytes.fromhex(sc["aggregate_pubkey"]))
s += arr("SC_SSZ_ROOT", bytes.fromhex(sc["ssz_root"]))
for test, vals in kat["unit_tests"].items():
    short = test.replace("test_", "").upper()
    short = short.replace("VERIFY_SYNC_COMMITTE_TARGET_WHEN_", "SC_")
    for k, v in vals.items():
        name = "%s__%s" % (short, k.upper())
        if isinstance(v, int):
            s += "static const uint64_t %s = %dull;\n" % (name, v)
        elif isinstance(v[0], list):
            s += arr2(name, v)
        else:
            s += arr(name, v)
open(out, "w").write(s)
