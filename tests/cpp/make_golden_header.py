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

# light-client updates 633 -> 634 (src/light_client_updates/*.json, the inputs of src/main.rs:84-175)
lc = json.load(open(os.path.join(here, "..", "golden", "lc_updates.json")))
hx = lambda v: bytes.fromhex(v[2:] if v.startswith("0x") else v)
for tag, u in lc.items():
    P = "LC%s__" % tag
    for which, h in (("ATTESTED", u["attested_beacon_header"]), ("FINALIZED", u["finality_update"]["header_update"]["beacon_header"])):
        s += "static const uint64_t %s%s_SLOT = %dull;\n" % (P, which, int(h["slot"]))
        s += "static const uint64_t %s%s_PROPOSER_INDEX = %dull;\n" % (P, which, int(h["proposer_index"]))
        for f in ("parent_root", "state_root", "body_root"):
            s += arr("%s%s_%s" % (P, which, f.upper()), hx(h[f]))
    s += arr2(P + "FINALITY_BRANCH", [hx(b) for b in u["finality_update"]["finality_branch"]])
    s += arr2(P + "NEXT_SYNC_COMMITTEE_PUBKEYS", [hx(b) for b in u["sync_committee_update"]["next_sync_committee"]["pubkeys"]])
    s += arr(P + "NEXT_SYNC_COMMITTEE_AGGREGATE", hx(u["sync_committee_update"]["next_sync_committee"]["aggregate_pubkey"]))
    s += arr2(P + "NEXT_SYNC_COMMITTEE_BRANCH", [hx(b) for b in u["sync_committee_update"]["next_sync_committee_branch"]])
    s += arr(P + "SYNC_COMMITTEE_BITS", hx(u["sync_aggregate"]["sync_committee_bits"]))
    s += arr(P + "SYNC_COMMITTEE_SIGNATURE", hx(u["sync_aggregate"]["sync_committee_signature"]))
open(out, "w").write(s)
