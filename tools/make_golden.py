#!/usr/bin/env python3
"""Extract the reference's own known-answer DATA into tests/golden/*.json.

Run in the build container (needs /root/reference, which does not exist on the
GPU box).  Only data is taken -- byte arrays, hex strings and integers that the
reference's tests feed to / expect from its gadgets -- never source text:

  src/merkle_tree_gadget.rs:183-325     zero-leaf SHA-256 Merkle roots (h = 1..4)
  src/sync_committee_pubkeys.rs:107-622 512 pubkeys, aggregate key, SSZ root
  src/unit_tests.rs:37-620              signing root, header root, finality branch,
                                        contract state, sync-committee branches
  src/light_client_update_period_63{3,4}.json   two mainnet LC updates (data files)

Poseidon vectors are NOT in the reference (its prover lives in an un-vendored
crate); tests/golden/poseidon_kat.json is written from the upstream plonky2
permutation test vectors quoted in SURVEY.md App. B.
"""
import json
import os
import re
import sys

REF = "/root/reference/eth-lc-plonky2/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def parse_int_arrays(body):
    """`let NAME[: T] = [ ... ];` with (possibly nested) integer arrays -> {NAME: list}"""
    out = {}
    for m in re.finditer(r"let\s+(?:mut\s+)?(\w+)(?:\s*:\s*[^=]+)?\s*=\s*(\[[\d\s,\[\]]+\])\s*(?:\.to_vec\(\))?;", body):
        txt = re.sub(r",\s*\]", "]", m.group(2))
        try:
            out[m.group(1)] = json.loads(txt)
        except json.JSONDecodeError:
            pass
    for m in re.finditer(r"let\s+(\w+)(?:\s*:\s*u64)?\s*=\s*(\d+)\s*;", body):
        out.setdefault(m.group(1), int(m.group(2)))
    return out


def split_tests(src):
    parts = re.split(r"\n\s*(?:#\[should_panic\]\s*)?\n?\s*fn\s+(test_\w+)\s*\(\)", src)
    tests = {}
    for i in range(1, len(parts), 2):
        tests[parts[i]] = parts[i + 1]
    return tests


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures are committed, nothing to do")
    os.makedirs(OUT, exist_ok=True)
    kat = {"_source": "extracted by tools/make_golden.py from /root/reference/eth-lc-plonky2/src (data only)"}

    # --- zero-leaf Merkle roots
    src = open(os.path.join(REF, "merkle_tree_gadget.rs")).read()
    roots = {}
    for name, body in split_tests(src).items():
        arrs = parse_int_arrays(body)
        m = re.search(r"(\d+)\]?\s*;\s*(\d+)\]", body)  # vec![[0u8; 32]; N]
        n = int(re.search(r"vec!\[\[0u8;\s*32\];\s*(\d+)\]", body).group(1))
        roots[str(n)] = bytes(arrs["root"]).hex()
    kat["zero_leaf_merkle_roots"] = roots

    # --- sync committee
    src = open(os.path.join(REF, "sync_committee_pubkeys.rs")).read()
    hexes = re.findall(r'"0x([0-9a-fA-F]+)"', src)
    pks = [h for h in hexes if len(h) == 96]
    r32 = [h for h in hexes if len(h) == 64]
    assert len(pks) == 513 and len(r32) == 1, (len(pks), len(r32))
    kat["sync_committee"] = {"pubkeys": pks[:512], "aggregate_pubkey": pks[512], "ssz_root": r32[0]}

    # --- unit_tests.rs vectors
    src = open(os.path.join(REF, "unit_tests.rs")).read()
    ut = {}
    for name, body in split_tests(src).items():
        arrs = parse_int_arrays(body)
        if arrs:
            ut[name] = arrs
    kat["unit_tests"] = ut
    json.dump(kat, open(os.path.join(OUT, "sha256_kat.json"), "w"), separators=(",", ":"))

    # --- LC update fixtures (compact copy of the two data files)
    upd = {}
    for period in (633, 634):
        upd[str(period)] = json.load(open(os.path.join(REF, f"light_client_update_period_{period}.json")))
    json.dump(upd, open(os.path.join(OUT, "lc_updates.json"), "w"), separators=(",", ":"))

    # --- Poseidon permutation vectors (upstream plonky2 test vectors)
    pos = {
        "_source": "upstream plonky2 poseidon_goldilocks test vectors as quoted in SURVEY.md App. B (not in the reference repo)",
        "round_constants_first4": ["b585f766f2144405", "7746a55f43921ad7", "b2fb0d31cee799b4", "0f6760a4803427d7"],
        "round_constant_359": "bc8dfb627fe558fc",
        "vectors": [
            {"in": "zeros", "out": "3c18a9786cb0b359 c4055e3364a246c3 7953db0ab48808f4 c71603f33a1144ca d7709673896996dc 46a84e87642f44ed d032648251ee0b3c 1c687363b207df62 df8565563e8045fe 40f5b37ff4254dae d070f637b431067c 1792b1c4342109d7".split()},
            {"in": "range12", "out": "d64e1e3efc5b8e9e 53666633020aaa47 d40285597c6a8825 613a4f81e81231d2 414754bfebd051f0 cb1f8980294a023f 6eb2a9e4d54a9d0f 1902bc3af467e056 f045d5eafdc6021f e4150f77caaa3be5 c9bfd01d39b50cce 5c0a27fcb0e1459b".split()},
            {"in": "neg_one", "out": "be0085cfc57a8357 d95af71847d05c09 cf55a13d33c1c953 95803a74f4530e82 fcd99eb30a135df1 e095905e913a3029 de0392461b42919b 7d3260e24e81d031 10d3d0465d9deaa0 a87571083dfc2a47 e18263681e9958f8 e28e96f1ae5e60d3".split()},
        ],
    }
    json.dump(pos, open(os.path.join(OUT, "poseidon_kat.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
