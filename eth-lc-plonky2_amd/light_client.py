"""The reference's main() on this backend, in-process: ctypes binding of host/lc_capi.h (liblcp2_host.so = the C++ host layer:
CircuitBuilder, the gadgets of src/merkle_tree_gadget.rs / sync_committee_pubkeys.rs / targets.rs, light-client update ingestion,
witness generation with the SHA-256 rows filled on the device).  `LightClientStep.prove()` is `data.prove(pw)` of
eth-lc-plonky2/src/main.rs:229-232 - witness generation inside, as the reference times it - on the caller's Context (device and
stream).  examples/lc_prover.cpp is the same flow as a program."""
import ctypes
import json
import os

import numpy as np

from . import binding as _b
from . import build as _build

BLS_PROOF_STAND_IN, SYNC_COMMITTEE_ONLY = 1, 2


class Info(ctypes.Structure):
    _fields_ = [("degree_bits", ctypes.c_uint32), ("num_public_inputs", ctypes.c_uint32), ("num_gates", ctypes.c_uint64), ("proof_words", ctypes.c_uint64),
                ("build_ms", ctypes.c_double), ("attach_ms", ctypes.c_double), ("inner_prove_ms", ctypes.c_double),
                ("inner_degree_bits", ctypes.c_uint32), ("inner_public_inputs", ctypes.c_uint32)]


_host = None


def load_host_library():
    global _host
    if _host is not None:
        return _host
    _b.load_library()  # liblcp2.so first (and torch's HIP runtime before it, if torch is installed)
    path = _build.HOST_LIB
    if not os.path.exists(path):
        path = _build.build_host()
    lib = ctypes.CDLL(path)
    c = ctypes
    lib.lch_light_client_step_create.restype = c.c_int
    lib.lch_light_client_step_create.argtypes = [c.c_void_p, c.c_char_p, c.c_char_p, c.c_uint32, c.c_uint32, c.POINTER(c.c_void_p)]
    lib.lch_destroy.restype = None
    lib.lch_destroy.argtypes = [c.c_void_p]
    lib.lch_get_info.restype = c.c_int
    lib.lch_get_info.argtypes = [c.c_void_p, c.POINTER(Info)]
    lib.lch_prove.restype = c.c_int
    lib.lch_prove.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t]
    lib.lch_verify.restype = c.c_int
    lib.lch_verify.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t]
    lib.lch_expected_public_inputs.restype = c.c_int
    lib.lch_expected_public_inputs.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t]
    lib.lch_last_error.restype = c.c_char_p
    _host = lib
    return lib


def reference_updates(path=None):
    """the two consecutive updates the reference ships (src/light_client_update_period_633.json / _634.json), from the committed
    fixture tests/golden/lc_updates.json: (prev_json, cur_json) as text"""
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "lc_updates.json")
    lc = json.load(open(path))
    return json.dumps(lc["633"]), json.dumps(lc["634"])


class LightClientStep:
    """build() of the light-client circuit for an update pair + its PartialWitness; prove() = data.prove(pw), verify() = data.verify(proof)"""

    def __init__(self, ctx, prev_json, cur_json, flags=0, extra_committees=0):
        self.lib = load_host_library()
        self.ctx = ctx  # keeps the context alive
        h = ctypes.c_void_p()
        rc = self.lib.lch_light_client_step_create(ctx.handle, prev_json.encode(), cur_json.encode(), flags, extra_committees, ctypes.byref(h))
        if rc:
            raise _b.Lcp2Error(rc, self.lib.lch_last_error().decode())
        self.handle = h
        self.info = Info()
        self.lib.lch_get_info(self.handle, ctypes.byref(self.info))
        n = 8 if flags & SYNC_COMMITTEE_ONLY else 16
        self.expected_public_inputs = np.zeros(n, dtype=np.uint64)
        rc = self.lib.lch_expected_public_inputs(self.handle, self.expected_public_inputs.ctypes.data_as(ctypes.c_void_p), n)
        if rc:
            raise _b.Lcp2Error(rc, self.lib.lch_last_error().decode())

    def prove(self):
        proof = np.zeros(self.info.proof_words, dtype=np.uint64)
        pis = np.zeros(self.info.num_public_inputs, dtype=np.uint64)
        rc = self.lib.lch_prove(self.handle, proof.ctypes.data_as(ctypes.c_void_p), proof.size, pis.ctypes.data_as(ctypes.c_void_p), pis.size)
        if rc:
            raise _b.Lcp2Error(rc, self.lib.lch_last_error().decode())
        return proof, pis

    def verify(self, proof, pis):
        proof = np.ascontiguousarray(proof, dtype=np.uint64)
        pis = np.ascontiguousarray(pis, dtype=np.uint64)
        rc = self.lib.lch_verify(self.handle, proof.ctypes.data_as(ctypes.c_void_p), proof.size, pis.ctypes.data_as(ctypes.c_void_p), pis.size)
        if rc == -7:
            raise _b.ProofRejected(self.lib.lch_last_error().decode())
        if rc:
            raise _b.Lcp2Error(rc, self.lib.lch_last_error().decode())

    def close(self):
        if self.handle:
            self.lib.lch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
