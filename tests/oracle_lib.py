"""Loads oracle/liboracle.so (the CPU checker) for tests, smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
P = 0xFFFFFFFF00000001
c = ctypes
u64p = c.POINTER(c.c_uint64)


def build(force=False):
    lib = os.path.join(ODIR, "liboracle.so")
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        subprocess.run(["make", "-s", "-C", ODIR], check=True)
    return lib


_lib = None


def vp(a):
    return a.ctypes.data_as(c.c_void_p)


def load():
    global _lib
    if _lib is not None:
        return _lib
    # LCP2_ORACLE_LIB: another build of the same sources (tests/checks/oracle_sanitize.sh: AddressSanitizer + UBSan)
    L = c.CDLL(os.environ.get("LCP2_ORACLE_LIB") or build())
    V = c.c_void_p
    sig = {
        "orc_poseidon_round_constants": (u64p, []),
        "orc_poseidon_permute": (None, [V]),
        "orc_poseidon_permute_batch": (None, [V, V, c.c_size_t]),
        "orc_hash_no_pad": (None, [V, c.c_size_t, V]),
        "orc_hash_or_noop": (None, [V, c.c_size_t, V]),
        "orc_two_to_one": (None, [V, V, V]),
        "orc_merkle_cap": (c.c_int, [V, c.c_size_t, c.c_size_t, c.c_uint, V]),
        "orc_merkle_verify": (c.c_int, [V, c.c_size_t, c.c_size_t, V, c.c_uint, V]),
        "orc_sha256_two_to_one": (None, [V, V, V]),
        "orc_sha256_merkle_root": (None, [V, c.c_uint, V, V]),
        "orc_sha256_merkle_branch_root": (None, [V, V, c.c_uint, c.c_size_t, V]),
        "orc_sha256_compress": (None, [V, V, V]),
        "orc_ssz_sync_committee_root": (None, [V, V, V]),
        "orc_ssz_sync_committee_leaves": (None, [V, V]),
        "orc_contract_state_root": (None, [c.c_uint64, V, V, V, V]),
        "orc_beacon_header_root": (None, [c.c_uint64, c.c_uint64, V, V, V, V]),
        "orc_fft": (None, [V, c.c_size_t]),
        "orc_ifft": (None, [V, c.c_size_t]),
        "orc_coset_fft": (None, [V, c.c_size_t, c.c_uint64]),
        "orc_coset_ifft": (None, [V, c.c_size_t, c.c_uint64]),
        "orc_ifft_batch": (None, [V, c.c_size_t, c.c_size_t]),
        "orc_fft_batch": (None, [V, c.c_size_t, c.c_size_t]),
        "orc_lde_batch": (None, [V, c.c_size_t, c.c_size_t, c.c_uint, c.c_uint64, V]),
        "orc_circuit_new": (V, [V, V, V, c.c_uint32, V, c.c_uint32, V, c.c_size_t, V, c.c_size_t, c.c_uint32]),
        "orc_verifier_new": (V, [V, V, c.c_uint32, V, c.c_uint32, V, c.c_size_t, V, c.c_size_t, c.c_uint32, V, V]),
        "orc_circuit_free": (None, [V]),
        "orc_circuit_digest": (None, [V, V, V]),
        "orc_proof_words": (c.c_size_t, [V]),
        "orc_prove": (c.c_int, [V, V, V, V]),
        "orc_verify": (c.c_int, [V, V, V]),
        "orc_check_witness": (c.c_size_t, [V, V, V, V]),
        "orc_last_challenges": (None, [V, V]),
        "orc_gl_mul": (c.c_uint64, [c.c_uint64, c.c_uint64]),
        "orc_gl_add": (c.c_uint64, [c.c_uint64, c.c_uint64]),
        "orc_gl_sub": (c.c_uint64, [c.c_uint64, c.c_uint64]),
        "orc_gl_inv": (c.c_uint64, [c.c_uint64]),
        "orc_gl_pow": (c.c_uint64, [c.c_uint64, c.c_uint64]),
        "orc_gl_root_of_unity": (c.c_uint64, [c.c_uint]),
        "orc_gl2_mul": (None, [V, V, V]),
        "orc_gl2_inv": (None, [V, V]),
    }
    for name, (res, args) in sig.items():
        if hasattr(L, name):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
    _lib = L
    return L


# ---- numpy helpers shared by the tests
def rand_field(rng, shape, canonical=True):
    """uniform field elements; canonical=False mixes in non-canonical encodings (>= p)"""
    a = rng.integers(0, P, size=shape, dtype=np.uint64, endpoint=False)
    if not canonical:
        flat = a.reshape(-1)
        k = max(1, flat.size // 16)
        idx = rng.choice(flat.size, size=k, replace=False)
        small = rng.integers(0, 2 ** 32 - 1, size=k, dtype=np.uint64)
        flat[idx] = small + np.uint64(P)  # values in [p, 2^64)
    return a


def bitrev_perm(bits):
    n = 1 << bits
    idx = np.arange(n, dtype=np.uint64)
    r = np.zeros(n, dtype=np.uint64)
    for b in range(bits):
        r |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(bits - 1 - b)
    return r.astype(np.int64)


def merkle_cap(L, leaves, cap_height):
    leaves = np.ascontiguousarray(leaves, dtype=np.uint64)
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    rc = L.orc_merkle_cap(vp(leaves), leaves.shape[0], leaves.shape[1], cap_height, vp(cap))
    assert rc == 0
    return cap


def lde_leaf_order(L, coeffs, rate_bits=3, shift=7):
    """oracle LDE (natural order) re-indexed to plonky2's Merkle leaf order (bit-reversed rows)"""
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    ncols, n = coeffs.shape
    out = np.zeros((ncols, n << rate_bits), dtype=np.uint64)
    L.orc_lde_batch(vp(coeffs), ncols, n, rate_bits, shift, vp(out))
    perm = bitrev_perm(int(n).bit_length() - 1 + rate_bits)
    return out[:, perm]


def commit_reference(L, values, rate_bits=3, cap_height=4):
    """PolynomialBatch::from_values restated with the oracle primitives.
    Returns (coeffs [ncols][n], lde in leaf order [ncols][8n], cap [2^cap_height][4])."""
    values = np.ascontiguousarray(values, dtype=np.uint64) % np.uint64(P)
    ncols, n = values.shape
    coeffs = values.copy()
    L.orc_ifft_batch(vp(coeffs), ncols, n)
    lde = lde_leaf_order(L, coeffs, rate_bits)
    leaves = np.ascontiguousarray(lde.T)
    return coeffs, lde, merkle_cap(L, leaves, cap_height)


def merkle_verify(L, leaf, index, siblings, cap):
    leaf = np.ascontiguousarray(leaf, dtype=np.uint64)
    siblings = np.ascontiguousarray(siblings, dtype=np.uint64)
    cap = np.ascontiguousarray(cap, dtype=np.uint64)
    return bool(L.orc_merkle_verify(vp(leaf), leaf.size, int(index), vp(siblings), siblings.shape[0], vp(cap)))


class OracleChallenges(c.Structure):
    _fields_ = [("betas", c.c_uint64 * 4), ("gammas", c.c_uint64 * 4), ("alphas", c.c_uint64 * 4), ("zeta", c.c_uint64 * 2),
                ("fri_alpha", c.c_uint64 * 2), ("fri_betas", (c.c_uint64 * 2) * 8), ("pow_witness", c.c_uint64),
                ("query_indices", c.c_uint64 * 64)]


class OracleCircuit:
    """orc_circuit built from an eth_lc_plonky2_amd.circuit.Circuit (the same description the GPU prover consumes)."""

    def __init__(self, L, circ):
        self.L, self.circ = L, circ
        gs = circ.gateset
        self.params = circ.params
        self.h = L.orc_circuit_new(c.byref(circ.params), vp(circ.constants_sigmas), vp(circ.k_is), gs.num_selectors,
                                   c.cast(circ.gates_array, c.c_void_p), len(gs.gates), vp(gs.code), gs.code_len, vp(gs.imm),
                                   gs.imm.size, circ.num_public_inputs)
        assert self.h, "orc_circuit_new rejected the description"
        self.proof_words = L.orc_proof_words(c.byref(circ.params))

    @classmethod
    def verifier_only(cls, L, circ, digest, cap):
        """the oracle's verifier from digest + constants/sigmas cap alone (plonky2's VerifierCircuitData): no preprocessed
        values are read and nothing is committed, so it runs at any circuit size (oracle/plonk.c orc_verifier_new)"""
        self = cls.__new__(cls)
        self.L, self.params = L, circ.params
        gs = circ.gateset
        d = np.ascontiguousarray(digest, dtype=np.uint64)
        cp = np.ascontiguousarray(cap, dtype=np.uint64)
        assert d.size == 4 and cp.size == 4 << circ.params.cap_height
        self.h = L.orc_verifier_new(c.byref(circ.params), vp(circ.k_is), gs.num_selectors, c.cast(circ.gates_array, c.c_void_p),
                                    len(gs.gates), vp(gs.code), gs.code_len, vp(gs.imm), gs.imm.size, circ.num_public_inputs, vp(d), vp(cp))
        assert self.h, "orc_verifier_new rejected the description"
        self.proof_words = L.orc_proof_words(c.byref(circ.params))
        return self

    def check_witness(self, wires, pis):
        bad = np.zeros(2, dtype=np.uint64)
        return self.L.orc_check_witness(self.h, vp(np.ascontiguousarray(wires)), vp(np.ascontiguousarray(pis)), vp(bad)), bad

    def prove(self, wires, pis):
        proof = np.zeros(self.proof_words, dtype=np.uint64)
        rc = self.L.orc_prove(self.h, vp(np.ascontiguousarray(wires, dtype=np.uint64)), vp(np.ascontiguousarray(pis, dtype=np.uint64)), vp(proof))
        assert rc == 0
        return proof

    def verify(self, proof, pis):
        return self.L.orc_verify(self.h, vp(np.ascontiguousarray(proof, dtype=np.uint64)), vp(np.ascontiguousarray(pis, dtype=np.uint64)))

    def digest(self):
        d = np.zeros(4, dtype=np.uint64)
        cap = np.zeros((1 << self.params.cap_height, 4), dtype=np.uint64)
        self.L.orc_circuit_digest(self.h, vp(d), vp(cap))
        return d, cap

    def challenges(self):
        ch = OracleChallenges()
        self.L.orc_last_challenges(self.h, c.byref(ch))
        return ch

    def close(self):
        if self.h:
            self.L.orc_circuit_free(self.h)
            self.h = None
