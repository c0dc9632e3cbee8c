"""Pins the oracle (CPU restatement) to every known-answer vector the reference's own tests hold
for the hot path, plus the upstream Poseidon permutation vectors."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle_lib import P, vp

G = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(G, "sha256_kat.json")))
POS = json.load(open(os.path.join(G, "poseidon_kat.json")))
UPD = json.load(open(os.path.join(G, "lc_updates.json")))


def b32(x):
    return np.frombuffer(bytes(x) if not isinstance(x, str) else bytes.fromhex(x.replace("0x", "")), dtype=np.uint8).copy()


def test_field_reduction_against_python_ints(oracle):
    rng = np.random.default_rng(0)
    edge = [0, 1, 2, P - 1, P - 2, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 63, P >> 1, 0xFFFFFFFF, 0xFFFFFFFF00000000]
    vals = edge + [int(x) for x in rng.integers(0, P, size=200, dtype=np.uint64)]
    for a in vals[:40]:
        for b in vals[:40]:
            assert oracle.orc_gl_mul(a, b) == (a * b) % P
            assert oracle.orc_gl_add(a, b) == (a + b) % P
            assert oracle.orc_gl_sub(a, b) == (a - b) % P
    for a in vals[1:60]:
        assert oracle.orc_gl_mul(a, oracle.orc_gl_inv(a)) == 1
    # non-canonical inputs are accepted
    assert oracle.orc_gl_mul(P + 5, 3) == 15
    # two-adic root of unity
    assert oracle.orc_gl_root_of_unity(32) == 1753635133440165772 == pow(7, (P - 1) >> 32, P)
    assert pow(oracle.orc_gl_root_of_unity(5), 32, P) == 1 and pow(oracle.orc_gl_root_of_unity(5), 16, P) != 1


def test_extension_field(oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        a = rng.integers(0, P, size=2, dtype=np.uint64)
        b = rng.integers(0, P, size=2, dtype=np.uint64)
        out = np.zeros(2, dtype=np.uint64)
        oracle.orc_gl2_mul(vp(a), vp(b), vp(out))
        a0, a1, b0, b1 = map(int, (a[0], a[1], b[0], b[1]))
        assert int(out[0]) == (a0 * b0 + 7 * a1 * b1) % P and int(out[1]) == (a0 * b1 + a1 * b0) % P
        inv = np.zeros(2, dtype=np.uint64)
        oracle.orc_gl2_inv(vp(a), vp(inv))
        one = np.zeros(2, dtype=np.uint64)
        oracle.orc_gl2_mul(vp(a), vp(inv), vp(one))
        assert list(one) == [1, 0]


def test_poseidon_round_constants(oracle):
    rc = oracle.orc_poseidon_round_constants()
    assert ["%016x" % rc[i] for i in range(4)] == POS["round_constants_first4"]
    assert "%016x" % rc[359] == POS["round_constant_359"]
    assert all(rc[i] < P for i in range(360))


@pytest.mark.parametrize("vec", POS["vectors"], ids=lambda v: v["in"])
def test_poseidon_permutation_vectors(oracle, vec):
    init = {"zeros": [0] * 12, "range12": list(range(12)), "neg_one": [P - 1] * 12}[vec["in"]]
    s = np.array(init, dtype=np.uint64)
    oracle.orc_poseidon_permute(vp(s))
    assert ["%016x" % x for x in s] == vec["out"]


def test_hashing_modes(oracle):
    rng = np.random.default_rng(2)
    x = rng.integers(0, P, size=20, dtype=np.uint64)
    # hash_or_noop pads short inputs
    out = np.zeros(4, dtype=np.uint64)
    oracle.orc_hash_or_noop(vp(x), 3, vp(out))
    assert list(out) == [x[0], x[1], x[2], 0]
    # two_to_one = permutation of [l, r, 0,0,0,0]
    st = np.zeros(12, dtype=np.uint64)
    st[:8] = x[:8]
    oracle.orc_poseidon_permute(vp(st))
    oracle.orc_two_to_one(vp(x[:4].copy()), vp(x[4:8].copy()), vp(out))
    assert list(out) == list(st[:4])
    # hash_no_pad of 8 elements = one permutation in overwrite mode; of 9 = two
    oracle.orc_hash_no_pad(vp(x), 8, vp(out))
    assert list(out) == list(st[:4])
    st2 = st.copy()
    st2[0] = x[8]
    oracle.orc_poseidon_permute(vp(st2))
    oracle.orc_hash_no_pad(vp(x), 9, vp(out))
    assert list(out) == list(st2[:4])


def sha_root(oracle, leaves, height):
    leaves = np.ascontiguousarray(leaves, dtype=np.uint8)
    root = np.zeros(32, dtype=np.uint8)
    oracle.orc_sha256_merkle_root(vp(leaves), height, vp(root), None)
    return bytes(root).hex()


@pytest.mark.parametrize("n", [2, 4, 8, 16])
def test_zero_leaf_merkle_roots(oracle, n):  # src/merkle_tree_gadget.rs:183-325
    h = n.bit_length() - 1
    assert sha_root(oracle, np.zeros((n, 32), dtype=np.uint8), h) == KAT["zero_leaf_merkle_roots"][str(n)]


def test_two_to_one_is_sha256_of_concatenation(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        l = rng.integers(0, 256, 32, dtype=np.uint8)
        r = rng.integers(0, 256, 32, dtype=np.uint8)
        out = np.zeros(32, dtype=np.uint8)
        oracle.orc_sha256_two_to_one(vp(l), vp(r), vp(out))
        assert bytes(out) == hashlib.sha256(bytes(l) + bytes(r)).digest()


def test_ssz_sync_committee_root(oracle):  # src/sync_committee_pubkeys.rs:100-653
    sc = KAT["sync_committee"]
    pks = np.frombuffer(bytes.fromhex("".join(sc["pubkeys"])), dtype=np.uint8).copy()
    agg = b32(sc["aggregate_pubkey"])
    root = np.zeros(32, dtype=np.uint8)
    oracle.orc_ssz_sync_committee_root(vp(pks), vp(agg), vp(root))
    assert bytes(root).hex() == sc["ssz_root"]


def test_signing_root(oracle):  # src/unit_tests.rs:37-65
    t = KAT["unit_tests"]["test_signing_root"]
    out = np.zeros(32, dtype=np.uint8)
    oracle.orc_sha256_two_to_one(vp(b32(t["attested_header_root"])), vp(b32(t["domain"])), vp(out))
    assert list(out) == t["signing_root"]


def test_beacon_block_header_root(oracle):  # src/unit_tests.rs:67-106
    t = KAT["unit_tests"]["test_beacon_block_header"]
    out = np.zeros(32, dtype=np.uint8)
    oracle.orc_beacon_header_root(t["slot"], t["proposer_index"], vp(b32(t["parent_root"])), vp(b32(t["state_root"])),
                                  vp(b32(t["body_root"])), vp(out))
    assert list(out) == t["header_root"]


def branch_root(oracle, leaf, branch, height, index):
    br = np.ascontiguousarray(np.array(branch, dtype=np.uint8))
    out = np.zeros(32, dtype=np.uint8)
    oracle.orc_sha256_merkle_branch_root(vp(b32(leaf)), vp(br), height, index, vp(out))
    return list(out)


def test_verify_finality_branch(oracle):  # src/unit_tests.rs:108-167, index 105 height 6 (targets.rs:25-26)
    t = KAT["unit_tests"]["test_verify_finality_branch"]
    assert branch_root(oracle, t["finalized_header_root"], t["finality_branch"], 6, 105) == t["attested_state_root"]


def test_contract_state(oracle):  # src/unit_tests.rs:169-246 (BASELINE config 1)
    t = KAT["unit_tests"]["test_contract_state"]
    for which in ("cur", "new"):
        out = np.zeros(32, dtype=np.uint8)
        oracle.orc_contract_state_root(t[f"{which}_slot"], vp(b32(t[f"{which}_header"])), vp(b32(t[f"{which}_sync_committee_i"])),
                                       vp(b32(t[f"{which}_sync_committee_ii"])), vp(out))
        assert list(out) == t[f"{which}_state"]


@pytest.mark.parametrize("name,ok", [
    ("test_verify_sync_committe_target_when_attested_from_next_period1", True),
    ("test_verify_sync_committe_target_when_attested_from_next_period2", False),  # #[should_panic] in the reference
    ("test_verify_sync_committe_target_when_not_attested_from_next_period1", True),
])
def test_sync_committee_branches(oracle, name, ok):  # src/unit_tests.rs:288-620, index 55 height 5
    t = KAT["unit_tests"][name]
    got = branch_root(oracle, t["new_sync_committee_ii"], t["new_sync_committee_ii_branch"], 5, 55)
    assert (got == t["finalized_state_root"]) == ok


@pytest.mark.parametrize("period", ["633", "634"])
def test_lc_update_fixtures_are_self_consistent(oracle, period):  # src/light_client_update_period_63{3,4}.json
    u = UPD[period]
    att = u["attested_beacon_header"]
    fin = u["finality_update"]["header_update"]["beacon_header"]
    fin_root = np.zeros(32, dtype=np.uint8)
    oracle.orc_beacon_header_root(int(fin["slot"]), int(fin["proposer_index"]), vp(b32(fin["parent_root"])), vp(b32(fin["state_root"])),
                                  vp(b32(fin["body_root"])), vp(fin_root))
    br = [list(b32(x)) for x in u["finality_update"]["finality_branch"]]
    assert bytes(branch_root(oracle, bytes(fin_root), br, 6, 105)).hex() == att["state_root"][2:]
    nsc = u["sync_committee_update"]["next_sync_committee"]
    pks = np.frombuffer(bytes.fromhex("".join(p[2:] for p in nsc["pubkeys"])), dtype=np.uint8).copy()
    sc_root = np.zeros(32, dtype=np.uint8)
    oracle.orc_ssz_sync_committee_root(vp(pks), vp(b32(nsc["aggregate_pubkey"])), vp(sc_root))
    br = [list(b32(x)) for x in u["sync_committee_update"]["next_sync_committee_branch"]]
    assert bytes(branch_root(oracle, bytes(sc_root), br, 5, 55)).hex() == att["state_root"][2:]
    if period == "634":
        assert bytes(sc_root).hex() == KAT["sync_committee"]["ssz_root"]
        prev = UPD["633"]["sync_committee_update"]["next_sync_committee"]
        ppks = np.frombuffer(bytes.fromhex("".join(p[2:] for p in prev["pubkeys"])), dtype=np.uint8).copy()
        prev_root = np.zeros(32, dtype=np.uint8)
        oracle.orc_ssz_sync_committee_root(vp(ppks), vp(b32(prev["aggregate_pubkey"])), vp(prev_root))
        assert "0x" + bytes(prev_root).hex() == u["sync_committee_update"]["next_sync_committee_branch"][0]


def test_fft_roundtrip_and_definition(oracle):
    rng = np.random.default_rng(4)
    n = 32
    x = rng.integers(0, P, size=n, dtype=np.uint64)
    v = x.copy()
    oracle.orc_fft(vp(v), n)
    w = oracle.orc_gl_root_of_unity(5)
    for i in (0, 1, 5, 31):
        assert int(v[i]) == sum(int(x[j]) * pow(w, i * j, P) for j in range(n)) % P
    oracle.orc_ifft(vp(v), n)
    assert (v == x).all()
    c = x.copy()
    oracle.orc_coset_fft(vp(c), n, 7)
    assert int(c[3]) == sum(int(x[j]) * pow(7 * pow(w, 3, P), j, P) for j in range(n)) % P
    oracle.orc_coset_ifft(vp(c), n, 7)
    assert (c == x).all()
