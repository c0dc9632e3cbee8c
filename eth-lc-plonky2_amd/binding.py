"""ctypes binding of the C ABI declared in include/lcp2.h.

This is the same shape of stub a Rust `plonky2` fork would write with `extern "C"`
(see INTEGRATION.md); Python is used here only because the image has no Rust
toolchain.  There is no CPU fallback: if liblcp2.so is missing or no HIP device
is usable, calls raise `Lcp2Error`.
"""
import ctypes
import os

import numpy as np

from . import build as _build

u64p = ctypes.POINTER(ctypes.c_uint64)
u32p = ctypes.POINTER(ctypes.c_uint32)
u8p = ctypes.POINTER(ctypes.c_uint8)
MEM_HOST, MEM_DEVICE = 0, 1
E_UNSAT = -5
SECTION_OPENINGS, SECTION_FRI_CAP0, SECTION_AFTER_CAPS = 0, 1, 2

K_INTT, K_LDE, K_LEAF_HASH, K_MERKLE, K_PERM_Z, K_QUOTIENT, K_OPENINGS, K_FRI, K_POW, K_SHA256, K_OTHER = range(11)
KERNEL_FAMILIES = ["intt", "lde", "leaf_hash", "merkle", "perm_z", "quotient", "openings", "fri", "pow", "sha256", "other"]

GOLDILOCKS_P = 0xFFFFFFFF00000001


class Lcp2Error(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"lcp2 status {status}: {message}")
        self.status = status


class Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in (
        "degree_bits", "num_wires", "num_routed_wires", "num_constants", "rate_bits", "cap_height",
        "num_challenges", "quotient_degree_factor", "proof_of_work_bits", "num_query_rounds", "num_fri_layers")]
    _fields_.append(("fri_arity_bits", ctypes.c_uint32 * 8))


class CircuitDesc(ctypes.Structure):
    _fields_ = [("params", Params), ("constants_sigmas", ctypes.c_void_p), ("constants_sigmas_mem", ctypes.c_int),
                ("k_is", ctypes.c_void_p), ("num_selectors", ctypes.c_uint32), ("num_gates", ctypes.c_uint32),
                ("gates", ctypes.c_void_p), ("code", ctypes.c_void_p), ("code_words", ctypes.c_size_t),
                ("imm", ctypes.c_void_p), ("num_imm", ctypes.c_size_t), ("num_public_inputs", ctypes.c_uint32),
                ("num_regs", ctypes.c_uint32)]


class ProofLayout(ctypes.Structure):
    """lcp2_proof_layout: word offsets of every field of the flat proof"""
    _fields_ = ([(n, ctypes.c_uint64) for n in ("cap_words", "wires_cap", "zs_cap", "quot_cap", "op_constants", "op_sigmas", "op_wires", "op_zs",
                                                "op_zs_next", "op_partial_products", "op_quotient", "fri_caps", "queries", "query_words")]
                + [("q_init_off", ctypes.c_uint64 * 4), ("q_init_cols", ctypes.c_uint64 * 4), ("q_init_sib", ctypes.c_uint64),
                   ("q_step_off", ctypes.c_uint64 * 8), ("q_step_sib", ctypes.c_uint64 * 8)]
                + [(n, ctypes.c_uint64) for n in ("final_poly", "final_len", "pow_witness", "total")])


_lib = None


class ChallengerState(ctypes.Structure):
    """lcp2_challenger: plonky2's Challenger { sponge_state, input_buffer, output_buffer } by value"""
    _fields_ = [("sponge", ctypes.c_uint64 * 12), ("input", ctypes.c_uint64 * 8), ("output", ctypes.c_uint64 * 8),
                ("input_len", ctypes.c_uint32), ("output_len", ctypes.c_uint32)]


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so.  Two HIP runtimes in one
    process cannot both own the GPU, and torch tensors / streams are only meaningful to the runtime that made
    them, so when torch is installed its runtime is loaded first (RTLD_GLOBAL): liblcp2.so's dependency on
    SONAME libamdhip64.so.7 then binds to that same copy.  Without torch the system ROCm runtime is used."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return


def load_library():
    """Loads (building if the sources are newer) eth-lc-plonky2_amd/liblcp2.so."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.path.exists(path):
        path = _build.build_native()
    _share_torch_hip_runtime()
    lib = ctypes.CDLL(path)
    c = ctypes
    sigs = {
        "lcp2_status_str": (c.c_char_p, [c.c_int]),
        "lcp2_abi_version": (c.c_int, []),
        "lcp2_device_count": (c.c_int, []),
        "lcp2_params_standard": (c.c_int, [c.c_uint32, c.c_uint32, c.POINTER(Params)]),
        "lcp2_ctx_create": (c.c_int, [c.c_int, c.c_void_p, c.POINTER(c.c_void_p)]),
        "lcp2_ctx_create_ex": (c.c_int, [c.c_int, c.c_void_p, c.c_uint32, c.POINTER(c.c_void_p)]),
        "lcp2_ctx_destroy": (None, [c.c_void_p]),
        "lcp2_ctx_sync": (c.c_int, [c.c_void_p]),
        "lcp2_ctx_stream": (c.c_void_p, [c.c_void_p]),
        "lcp2_last_error": (c.c_char_p, [c.c_void_p]),
        "lcp2_poseidon_permute_batch": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t, c.c_int]),
        "lcp2_field_op_batch": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t, c.c_uint32, c.c_int]),
        "lcp2_merkle_cap": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_size_t, c.c_uint32, c.c_int, c.c_void_p]),
        "lcp2_ntt_batch": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_uint32, c.c_int, c.c_uint64, c.c_int]),
        "lcp2_lde_batch": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t, c.c_uint32, c.c_uint32, c.c_int]),
        "lcp2_sha256_tree": (c.c_int, [c.c_void_p, c.c_void_p, c.c_uint32, c.c_size_t, c.c_void_p, c.c_void_p, c.c_int]),
        "lcp2_sha256_witness": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_uint32, c.c_void_p, c.c_size_t, c.c_void_p, c.c_uint64, c.c_void_p]),
        "lcp2_scatter_cells": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_uint64]),
        "lcp2_poseidon_gate_rows": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_uint64]),
        "lcp2_commit_wires_rows_begin": (c.c_int, [c.c_void_p, c.c_void_p]),
        "lcp2_commit_wires_chunk": (c.c_int, [c.c_void_p, c.c_void_p, c.c_uint32, c.c_uint32]),
        "lcp2_commit_wires_rows_finish": (c.c_int, [c.c_void_p, c.c_void_p]),
        "lcp2_host_register": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t]),
        "lcp2_host_unregister": (c.c_int, [c.c_void_p, c.c_void_p]),
        "lcp2_witness_stage": (c.c_int, [c.c_void_p, c.c_void_p, c.c_uint32]),
        "lcp2_prove_staged": (c.c_int, [c.c_void_p, c.c_uint32, c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t]),
        "lcp2_buffer_alloc": (c.c_int, [c.c_void_p, c.c_size_t, c.POINTER(c.c_void_p)]),
        "lcp2_buffer_free": (c.c_int, [c.c_void_p, c.c_void_p]),
        "lcp2_buffer_zero": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t]),
        "lcp2_buffer_read": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t]),
        "lcp2_buffer_write": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t]),
        "lcp2_buffer_copy": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t]),
        "lcp2_buffer_copy_2d": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_size_t, c.c_size_t]),
        "lcp2_commit_values": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_uint32, c.c_uint32, c.c_uint32, c.c_int, c.POINTER(c.c_void_p), c.c_void_p]),
        "lcp2_commit_coeffs": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_uint32, c.c_uint32, c.c_uint32, c.c_int, c.POINTER(c.c_void_p), c.c_void_p]),
        "lcp2_commit_cosets": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_uint32, c.c_uint32, c.c_uint32, c.c_uint32, c.c_uint32, c.c_int, c.POINTER(c.c_void_p), c.c_void_p]),
        "lcp2_oracle_destroy": (None, [c.c_void_p]),
        "lcp2_oracle_open": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]),
        "lcp2_oracle_read": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p]),
        "lcp2_circuit_create": (c.c_int, [c.c_void_p, c.POINTER(CircuitDesc), c.POINTER(c.c_void_p)]),
        "lcp2_verifier_create": (c.c_int, [c.POINTER(CircuitDesc), c.c_void_p, c.c_void_p, c.POINTER(c.c_void_p)]),
        "lcp2_circuit_destroy": (None, [c.c_void_p]),
        "lcp2_circuit_digest": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p]),
        "lcp2_proof_words": (c.c_size_t, [c.POINTER(Params)]),
        "lcp2_proof_layout_of": (c.c_int, [c.POINTER(Params), c.POINTER(ProofLayout)]),
        "lcp2_prove": (c.c_int, [c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t]),
        "lcp2_commit_wires": (c.c_int, [c.c_void_p, c.c_void_p, c.c_int, c.c_void_p]),
        "lcp2_commit_wires_coeffs": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]),
        "lcp2_commit_wires_rows": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]),
        "lcp2_perm_zs_rows_begin": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]),
        "lcp2_perm_zs_rows_finish": (c.c_int, [c.c_void_p, c.c_void_p, c.POINTER(c.c_void_p), c.POINTER(c.c_size_t)]),
        "lcp2_perm_zs_commit": (c.c_int, [c.c_void_p, c.c_void_p]),
        "lcp2_perm_zs": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]),
        "lcp2_quotient": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]),
        "lcp2_fri_open": (c.c_int, [c.c_void_p, c.c_void_p, c.POINTER(ChallengerState), c.c_void_p]),
        "lcp2_fri_open_begin": (c.c_int, [c.c_void_p, c.c_void_p, c.POINTER(ChallengerState), c.c_void_p]),
        "lcp2_fri_open_commit": (c.c_int, [c.c_void_p, c.c_void_p]),
        "lcp2_fri_open_finish": (c.c_int, [c.c_void_p, c.POINTER(ChallengerState), c.c_void_p]),
        "lcp2_proof_section": (c.c_int, [c.c_void_p, c.c_int, c.POINTER(c.c_size_t), c.POINTER(c.c_size_t)]),
        "lcp2_challenger_init": (None, [c.POINTER(ChallengerState)]),
        "lcp2_challenger_observe": (c.c_int, [c.POINTER(ChallengerState), c.c_void_p, c.c_size_t]),
        "lcp2_challenger_get": (c.c_int, [c.POINTER(ChallengerState), c.c_void_p, c.c_size_t]),
        "lcp2_hash_no_pad": (c.c_int, [c.c_void_p, c.c_size_t, c.c_void_p]),
        "lcp2_circuit_create_sharded": (c.c_int, [c.c_void_p, c.POINTER(CircuitDesc), c.c_uint32, c.c_uint32, c.POINTER(c.c_void_p)]),
        "lcp2_circuit_set_constants_cap": (c.c_int, [c.c_void_p, c.c_void_p]),
        "lcp2_quotient_values": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p]),
        "lcp2_quotient_buffer": (c.c_int, [c.c_void_p, c.POINTER(c.c_void_p), c.POINTER(c.c_size_t)]),
        "lcp2_quotient_commit": (c.c_int, [c.c_void_p, c.c_void_p]),
        "lcp2_verify": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.POINTER(c.c_int)]),
        "lcp2_last_challenges": (c.c_int, [c.c_void_p, c.c_void_p]),
        "lcp2_proof_bytes": (c.c_size_t, [c.POINTER(Params), c.c_size_t, c.c_uint32]),
        "lcp2_proof_to_bytes": (c.c_int, [c.POINTER(Params), c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_uint32, c.c_void_p, c.c_size_t]),
        "lcp2_proof_from_bytes": (c.c_int, [c.POINTER(Params), c.c_void_p, c.c_size_t, c.c_uint32, c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t]),
        "lcp2_verifier_data_to_bytes": (c.c_int, [c.c_void_p, c.c_void_p, c.c_size_t]),
        "lcp2_prof_enable": (c.c_int, [c.c_void_p, c.c_int]),
        "lcp2_prof_reset": (c.c_int, [c.c_void_p]),
        "lcp2_prof_get": (c.c_int, [c.c_void_p, c.c_int, c.POINTER(c.c_double), c.POINTER(c.c_uint64), c.POINTER(c.c_double)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not export what lcp2.h declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = None  # filled by tests from include/lcp2.h


def _np_u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def standard_params(degree_bits, num_constants=4):
    lib = load_library()
    p = Params()
    rc = lib.lcp2_params_standard(degree_bits, num_constants, ctypes.byref(p))
    if rc:
        raise Lcp2Error(rc, lib.lcp2_status_str(rc).decode())
    return p


class Oracle:
    """PolynomialBatch handle (device resident)."""

    def __init__(self, ctx, handle, ncols, log_n, rate_bits, cap_height, cap):
        self.ctx, self.handle = ctx, handle
        self.ncols, self.log_n, self.rate_bits, self.cap_height = ncols, log_n, rate_bits, cap_height
        self.cap = cap
        self.block_first, self.block_count = 0, 1 << rate_bits  # a coset-sharded oracle holds fewer leaf blocks

    def open(self, indices):
        idx = _np_u64(indices)
        k = idx.size
        nsib = self.log_n + self.rate_bits - self.cap_height  # the same for a coset-sharded oracle: local cap = global cap slice
        leaves = np.zeros((k, self.ncols), dtype=np.uint64)
        sib = np.zeros((k, max(nsib, 0), 4), dtype=np.uint64)
        self.ctx._check(self.ctx.lib.lcp2_oracle_open(self.handle, _ptr(idx), k, _ptr(leaves), _ptr(sib)))
        return leaves, sib

    def read(self, coeffs=True, lde=True):
        n = 1 << self.log_n
        c = np.zeros((self.ncols, n), dtype=np.uint64) if coeffs else None
        l = np.zeros((self.ncols, n * self.block_count), dtype=np.uint64) if lde else None
        self.ctx._check(self.ctx.lib.lcp2_oracle_read(self.handle, _ptr(c), _ptr(l)))
        return c, l

    def close(self):
        if self.handle:
            self.ctx.lib.lcp2_oracle_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One prover context = one GPU + one HIP stream (lcp2_ctx).  stream: a raw hipStream_t, or None / 0 for a private
    non-blocking stream - torch.cuda.current_stream().cuda_stream is 0 for the default stream, so work queued there (a fill, an
    upload) is NOT ordered before the library's kernels: torch.cuda.synchronize() first, ctx.sync() before reading results - or
    pass order_with_default_stream=True (LCP2_CTX_ORDER_WITH_DEFAULT_STREAM): the private stream is then an ordinary blocking
    stream, which the runtime orders against stream 0 in both directions."""

    def __init__(self, device=0, stream=None, order_with_default_stream=False):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.lcp2_ctx_create_ex(device, stream, 1 if order_with_default_stream else 0, ctypes.byref(h))
        if rc:
            raise Lcp2Error(rc, self.lib.lcp2_status_str(rc).decode())
        self.handle = h

    def _check(self, rc):
        if rc:
            msg = self.lib.lcp2_status_str(rc).decode()
            detail = self.lib.lcp2_last_error(self.handle)
            raise Lcp2Error(rc, msg + (": " + detail.decode() if detail else ""))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lcp2_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self.lib.lcp2_ctx_sync(self.handle))

    def stream_ptr(self):
        """the hipStream_t the context runs on (an integer; torch.cuda.ExternalStream(ptr) wraps it)"""
        return self.lib.lcp2_ctx_stream(self.handle) or 0

    # ---- primitives on host numpy arrays
    def poseidon_permute_batch(self, states):
        s = _np_u64(states).reshape(-1, 12)
        out = np.empty_like(s)
        self._check(self.lib.lcp2_poseidon_permute_batch(self.handle, _ptr(s), _ptr(out), s.shape[0], MEM_HOST))
        return out

    FIELD_OPS = {"mul": 0, "pow7": 1, "add": 2, "sub": 3, "canon": 4, "add_lazy": 5, "sub_lazy": 6, "shl32_lazy": 24, "mul_u32": 25,
                 **{"shl%d" % (12 * k): 16 + k for k in range(1, 8)}}

    def field_op_batch(self, op, a, b=None):
        """elementwise field arithmetic with the device functions of csrc/gl64.hpp (LCP2_FIELD_*); operands any u64, results canonical"""
        a = _np_u64(a).reshape(-1)
        out = np.empty_like(a)
        if b is not None:
            b = _np_u64(b).reshape(-1)
            assert a.shape == b.shape
        self._check(self.lib.lcp2_field_op_batch(self.handle, _ptr(a), _ptr(b) if b is not None else None, _ptr(out), a.shape[0],
                                                 self.FIELD_OPS[op], MEM_HOST))
        return out

    def merkle_cap(self, leaves, cap_height):
        l = _np_u64(leaves)
        assert l.ndim == 2
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        self._check(self.lib.lcp2_merkle_cap(self.handle, _ptr(l), l.shape[0], l.shape[1], cap_height, MEM_HOST, _ptr(cap)))
        return cap

    def ntt_batch(self, cols, inverse=False, shift=1):
        c = _np_u64(cols).copy()
        assert c.ndim == 2
        log_n = int(c.shape[1]).bit_length() - 1
        assert 1 << log_n == c.shape[1]
        self._check(self.lib.lcp2_ntt_batch(self.handle, _ptr(c), c.shape[0], log_n, int(inverse), shift, MEM_HOST))
        return c

    def lde_batch(self, coeffs, rate_bits=3):
        c = _np_u64(coeffs)
        log_n = int(c.shape[1]).bit_length() - 1
        out = np.zeros((c.shape[0], c.shape[1] << rate_bits), dtype=np.uint64)
        self._check(self.lib.lcp2_lde_batch(self.handle, _ptr(c), _ptr(out), c.shape[0], log_n, rate_bits, MEM_HOST))
        return out

    def sha256_tree(self, leaves, height, trees=1, trace=False):
        l = np.ascontiguousarray(leaves, dtype=np.uint8).reshape(trees, 1 << height, 32)
        nodes = np.zeros((trees, (2 << height) - 1, 32), dtype=np.uint8)
        tr = np.zeros((trees, (1 << height) - 1, 2, 176), dtype=np.uint32) if trace else None
        self._check(self.lib.lcp2_sha256_tree(self.handle, _ptr(l), height, trees, _ptr(nodes), _ptr(tr), MEM_HOST))
        return (nodes, tr) if trace else nodes

    def _commit(self, fn, cols, rate_bits, cap_height, mem, shape):
        if mem == MEM_HOST:
            c = _np_u64(cols)
            ncols, n = c.shape
            p = _ptr(c)
        else:
            ncols, n = shape
            p = ctypes.c_void_p(cols)
        log_n = int(n).bit_length() - 1
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        h = ctypes.c_void_p()
        self._check(fn(self.handle, p, ncols, log_n, rate_bits, cap_height, mem, ctypes.byref(h), _ptr(cap)))
        return Oracle(self, h, ncols, log_n, rate_bits, cap_height, cap)

    def commit_values(self, cols, rate_bits=3, cap_height=4, mem=MEM_HOST, shape=None):
        """PolynomialBatch::from_values.  cols: numpy [ncols][n] or a device pointer with shape=(ncols, n)."""
        return self._commit(self.lib.lcp2_commit_values, cols, rate_bits, cap_height, mem, shape)

    def commit_coeffs(self, cols, rate_bits=3, cap_height=4, mem=MEM_HOST, shape=None):
        return self._commit(self.lib.lcp2_commit_coeffs, cols, rate_bits, cap_height, mem, shape)

    def commit_cosets(self, coeffs, block_first, block_count, rate_bits=3, cap_height=4, mem=MEM_HOST, shape=None):
        """One rank's share of a coset-sharded commitment: coefficients of all columns in, its leaf blocks + cap part out."""
        if mem == MEM_HOST:
            c = _np_u64(coeffs)
            ncols, n = c.shape
            p = _ptr(c)
        else:
            ncols, n = shape
            p = ctypes.c_void_p(coeffs)
        log_n = int(n).bit_length() - 1
        cap = np.zeros((block_count << (cap_height - rate_bits), 4), dtype=np.uint64)
        h = ctypes.c_void_p()
        self._check(self.lib.lcp2_commit_cosets(self.handle, p, ncols, log_n, rate_bits, cap_height, block_first, block_count, mem,
                                                ctypes.byref(h), _ptr(cap)))
        o = Oracle(self, h, ncols, log_n, rate_bits, cap_height, cap)
        o.block_first, o.block_count = block_first, block_count
        return o

    # ---- device buffers
    def buffer_alloc(self, words):
        p = ctypes.c_void_p()
        self._check(self.lib.lcp2_buffer_alloc(self.handle, int(words) * 8, ctypes.byref(p)))
        return p.value

    def buffer_free(self, dev_ptr):
        self._check(self.lib.lcp2_buffer_free(self.handle, ctypes.c_void_p(dev_ptr)))

    def buffer_read(self, dev_ptr, words):
        out = np.zeros(words, dtype=np.uint64)
        self._check(self.lib.lcp2_buffer_read(self.handle, _ptr(out), ctypes.c_void_p(dev_ptr), words * 8))
        return out

    def buffer_write(self, dev_ptr, arr):
        a = _np_u64(arr)
        self._check(self.lib.lcp2_buffer_write(self.handle, ctypes.c_void_p(dev_ptr), _ptr(a), a.size * 8))

    def buffer_copy(self, dst_ptr, src_ptr, words):
        self._check(self.lib.lcp2_buffer_copy(self.handle, ctypes.c_void_p(dst_ptr), ctypes.c_void_p(src_ptr), words * 8))

    def buffer_copy_2d(self, dst_ptr, dst_pitch_words, src_ptr, src_pitch_words, width_words, height):
        """`height` runs of `width_words`, the pitches apart (device to device)"""
        self._check(self.lib.lcp2_buffer_copy_2d(self.handle, ctypes.c_void_p(dst_ptr), dst_pitch_words * 8, ctypes.c_void_p(src_ptr),
                                                 src_pitch_words * 8, width_words * 8, height))

    # ---- timing
    def prof_enable(self, on=True):
        self._check(self.lib.lcp2_prof_enable(self.handle, int(on)))

    def prof_reset(self):
        self._check(self.lib.lcp2_prof_reset(self.handle))

    def prof_get(self):
        out = {}
        for i, name in enumerate(KERNEL_FAMILIES):
            ms, n, b = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_double()
            self._check(self.lib.lcp2_prof_get(self.handle, i, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(b)))
            out[name] = {"ms": ms.value, "launches": n.value, "bytes": b.value}
        return out


class Challenger:
    """plonky2's Challenger<F, PoseidonHash> on the host (lcp2_challenger_*): the transcript of a caller that drives the
    seams of data.prove() itself (staged or sharded proving)."""

    def __init__(self):
        self.lib = load_library()
        self.state = ChallengerState()
        self.lib.lcp2_challenger_init(ctypes.byref(self.state))

    def observe(self, values):
        v = np.ascontiguousarray(np.asarray(values, dtype=np.uint64).ravel())
        rc = self.lib.lcp2_challenger_observe(ctypes.byref(self.state), _ptr(v), v.size)
        if rc:
            raise Lcp2Error(rc, self.lib.lcp2_status_str(rc).decode())

    def get(self, count=1):
        out = np.zeros(count, dtype=np.uint64)
        rc = self.lib.lcp2_challenger_get(ctypes.byref(self.state), _ptr(out), count)
        if rc:
            raise Lcp2Error(rc, self.lib.lcp2_status_str(rc).decode())
        return out


def hash_no_pad(values):
    """PoseidonHash::hash_no_pad on the host (public-input hash of the transcript)"""
    lib = load_library()
    v = np.ascontiguousarray(np.asarray(values, dtype=np.uint64).ravel())
    out = np.zeros(4, dtype=np.uint64)
    rc = lib.lcp2_hash_no_pad(_ptr(v) if v.size else None, v.size, _ptr(out))
    if rc:
        raise Lcp2Error(rc, lib.lcp2_status_str(rc).decode())
    return out


SER_PUBLIC_INPUT_COUNT = 1


def proof_layout(params):
    """where each field of plonky2's Proof / FriProof sits in the flat proof array (lcp2_proof_layout_of)"""
    lib, out = load_library(), ProofLayout()
    rc = lib.lcp2_proof_layout_of(ctypes.byref(params), ctypes.byref(out))
    if rc:
        raise Lcp2Error(rc, lib.lcp2_status_str(rc).decode())
    return out


def proof_to_bytes(params, proof, public_inputs, flags=SER_PUBLIC_INPUT_COUNT):
    """ProofWithPublicInputs::to_bytes (plonky2 util/serialization.rs layout, see include/lcp2.h; parity unpinned)"""
    lib = load_library()
    pr, pis = _np_u64(proof).ravel(), _np_u64(public_inputs).ravel()
    out = np.zeros(lib.lcp2_proof_bytes(ctypes.byref(params), pis.size, flags), dtype=np.uint8)
    rc = lib.lcp2_proof_to_bytes(ctypes.byref(params), _ptr(pr), pr.size, _ptr(pis), pis.size, flags, _ptr(out), out.size)
    if rc:
        raise Lcp2Error(rc, lib.lcp2_status_str(rc).decode())
    return out.tobytes()


def proof_from_bytes(params, data, num_public_inputs, flags=SER_PUBLIC_INPUT_COUNT):
    """-> (proof words, public inputs); Lcp2Error on a malformed buffer"""
    lib = load_library()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    proof = np.zeros(lib.lcp2_proof_words(ctypes.byref(params)), dtype=np.uint64)
    pis = np.zeros(num_public_inputs, dtype=np.uint64)
    rc = lib.lcp2_proof_from_bytes(ctypes.byref(params), _ptr(buf) if buf.size else None, buf.size, flags, _ptr(proof), proof.size, _ptr(pis), pis.size)
    if rc:
        raise Lcp2Error(rc, lib.lcp2_status_str(rc).decode())
    return proof, pis


class ProofRejected(Lcp2Error):
    """data.verify(proof) failed; .check names the failed step (see lcp2_verify)."""

    def __init__(self, check):
        super().__init__(-7, f"proof rejected (check {check})")
        self.check = check


def _describe(circ, constants_sigmas_ptr=None, mem=MEM_HOST):
    """lcp2_circuit_desc for a circuit.Circuit; returns (desc, keepalive)"""
    gs = circ.gateset
    gates = circ.gates_array
    d = CircuitDesc()
    d.params = circ.params
    d.constants_sigmas = constants_sigmas_ptr if constants_sigmas_ptr is not None else circ.constants_sigmas.ctypes.data
    d.constants_sigmas_mem = mem
    d.k_is = circ.k_is.ctypes.data
    d.num_selectors, d.num_gates = gs.num_selectors, len(gs.gates)
    d.gates = ctypes.cast(gates, ctypes.c_void_p)
    d.code, d.code_words = gs.code.ctypes.data, gs.code_len
    d.imm, d.num_imm = gs.imm.ctypes.data, gs.imm.size
    d.num_public_inputs, d.num_regs = circ.num_public_inputs, gs.max_regs
    return d, (gates, gs, circ)


class CircuitData:
    """builder.build::<C>() result: the handle `prove` and `verify` are called on
    (reference: eth-lc-plonky2/src/main.rs:227-233)."""

    def __init__(self, ctx, circ, handle, keep):
        self.ctx, self.circ, self.handle, self._keep = ctx, circ, handle, keep
        self.lib = load_library()
        self.proof_words = self.lib.lcp2_proof_words(ctypes.byref(circ.params))

    @classmethod
    def build(cls, ctx, circ, constants_sigmas_ptr=None, mem=MEM_HOST):
        d, keep = _describe(circ, constants_sigmas_ptr, mem)
        h = ctypes.c_void_p()
        ctx._check(ctx.lib.lcp2_circuit_create(ctx.handle, ctypes.byref(d), ctypes.byref(h)))
        return cls(ctx, circ, h, keep)

    @classmethod
    def build_sharded(cls, ctx, circ, block_first, block_count, constants_sigmas_ptr=None, mem=MEM_HOST):
        """one rank's part of a coset-sharded circuit: holds the leaf blocks [block_first, block_first + block_count).
        The digest is valid after set_constants_cap(OR of every rank's digest()[1])."""
        d, keep = _describe(circ, constants_sigmas_ptr, mem)
        h = ctypes.c_void_p()
        ctx._check(ctx.lib.lcp2_circuit_create_sharded(ctx.handle, ctypes.byref(d), block_first, block_count, ctypes.byref(h)))
        return cls(ctx, circ, h, keep)

    def set_constants_cap(self, cap):
        cp = _np_u64(cap)
        self._check(self.lib.lcp2_circuit_set_constants_cap(self.handle, _ptr(cp)))

    def quotient_values(self, alphas, public_inputs_hash):
        a, h = _np_u64(alphas), _np_u64(public_inputs_hash)
        assert h.size == 4
        self._check(self.lib.lcp2_quotient_values(self.handle, _ptr(a), _ptr(h)))

    def quotient_buffer(self):
        """(device pointer, uint64 words) of the quotient values [num_challenges][8n]: the one bulk exchange of a sharded proof"""
        ptr, words = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self.lib.lcp2_quotient_buffer(self.handle, ctypes.byref(ptr), ctypes.byref(words)))
        return ptr.value, words.value

    def quotient_commit(self):
        cap = self._cap()
        self._check(self.lib.lcp2_quotient_commit(self.handle, _ptr(cap)))
        return cap

    @classmethod
    def verifier_only(cls, circ, digest, cap):
        lib = load_library()
        d, keep = _describe(circ, 0)
        h = ctypes.c_void_p()
        dg, cp = _np_u64(digest), _np_u64(cap)
        rc = lib.lcp2_verifier_create(ctypes.byref(d), _ptr(dg), _ptr(cp), ctypes.byref(h))
        if rc:
            raise Lcp2Error(rc, lib.lcp2_status_str(rc).decode())
        return cls(None, circ, h, keep)

    def _check(self, rc):
        if rc:
            if self.ctx is not None:
                self.ctx._check(rc)
            raise Lcp2Error(rc, self.lib.lcp2_status_str(rc).decode())

    def digest(self):
        d = np.zeros(4, dtype=np.uint64)
        cap = np.zeros((1 << self.circ.params.cap_height, 4), dtype=np.uint64)
        self._check(self.lib.lcp2_circuit_digest(self.handle, _ptr(d), _ptr(cap)))
        return d, cap

    def prove(self, wires, public_inputs, mem=MEM_HOST):
        """data.prove(pw): wires = full witness [num_wires][n] (numpy, or a device pointer with mem=MEM_DEVICE)"""
        pis = _np_u64(public_inputs).ravel()
        proof = np.zeros(self.proof_words, dtype=np.uint64)
        if mem == MEM_HOST:
            w = _np_u64(wires)
            if w.shape != (self.circ.params.num_wires, 1 << self.circ.params.degree_bits):
                raise Lcp2Error(-1, "witness must be [num_wires][n]")
            wp = _ptr(w)
        else:
            wp = ctypes.c_void_p(wires)
        self._check(self.lib.lcp2_prove(self.handle, wp, mem, _ptr(pis), pis.size, _ptr(proof), proof.size))
        return proof

    # ---- host witnesses with the upload off the critical path (lcp2_witness_stage / lcp2_prove_staged)
    def stage_witness(self, wires, slot):
        """starts the upload of a host witness into staging slot 0 / 1; `wires` must stay alive and unchanged until prove_staged(slot) returned"""
        w = _np_u64(wires)
        if w.shape != (self.circ.params.num_wires, 1 << self.circ.params.degree_bits):
            raise Lcp2Error(-1, "witness must be [num_wires][n]")
        if not hasattr(self, "_staged"):
            self._staged = {}
        self._staged[slot] = w
        self._check(self.lib.lcp2_witness_stage(self.handle, _ptr(w), slot))

    def prove_staged(self, slot, public_inputs):
        pis = _np_u64(public_inputs).ravel()
        proof = np.zeros(self.proof_words, dtype=np.uint64)
        self._check(self.lib.lcp2_prove_staged(self.handle, slot, _ptr(pis), pis.size, _ptr(proof), proof.size))
        self._staged.pop(slot, None)
        return proof

    def host_witness_benchmark(self, wires, public_inputs, steps=4, reference_proof=None):
        """`steps` proofs from HOST witnesses (two pinned copies of `wires`, alternating): the plain way - lcp2_prove(MEM_HOST), the copy
        in front of every proof - and the staged way - the upload of witness i + 1 overlaps proof i.  Returns the per-proof times."""
        import time
        ctx = self.ctx
        bufs = [np.ascontiguousarray(wires, dtype=np.uint64), np.array(wires, dtype=np.uint64, order="C")]
        for b in bufs:
            ctx._check(ctx.lib.lcp2_host_register(ctx.handle, _ptr(b), b.nbytes))
        try:
            ctx.sync()
            t0 = time.perf_counter()
            plain = self.prove(bufs[0], public_inputs, mem=MEM_HOST)
            t_plain = time.perf_counter() - t0
            self.stage_witness(bufs[0], 0)
            times = []
            for i in range(steps):
                t0 = time.perf_counter()
                if i + 1 < steps:
                    self.stage_witness(bufs[(i + 1) % 2], (i + 1) % 2)
                proof = self.prove_staged(i % 2, public_inputs)
                times.append(time.perf_counter() - t0)
            ok = bool((proof == plain).all()) and (reference_proof is None or bool((proof == reference_proof).all()))
        finally:
            ctx.sync()
            for b in bufs:
                ctx.lib.lcp2_host_unregister(ctx.handle, _ptr(b))
        steady = times[1:-1] if len(times) > 2 else times  # the first proof waits for its own upload, the last one has nothing to overlap
        return {"workload": "lcp2_prove from HOST witnesses (%.2f GB each, pinned with lcp2_host_register): upload of witness i + 1 on the copy stream while "
                            "proof i runs (lcp2_witness_stage / lcp2_prove_staged)" % (bufs[0].nbytes / 1e9),
                "ms_per_proof_steady_state": 1e3 * min(steady), "ms_per_proof_all": [round(1e3 * t, 2) for t in times],
                "ms_plain_host_prove": 1e3 * t_plain, "proofs_equal_device_resident_proof": ok}

    # ---- the seams of data.prove() one by one (the caller runs the Fiat-Shamir transcript)
    def _cap(self):
        return np.zeros((1 << self.circ.params.cap_height, 4), dtype=np.uint64)

    def commit_wires(self, wires, mem=MEM_HOST):
        cap = self._cap()
        if mem == MEM_HOST:
            self._staged_wires = _np_u64(wires)  # stays alive until the proof is finished
            wp = _ptr(self._staged_wires)
        else:
            wp = ctypes.c_void_p(wires)
        self._check(self.lib.lcp2_commit_wires(self.handle, wp, mem, _ptr(cap)))
        return cap

    def commit_wires_coeffs(self, wires_ptr, coeffs_ptr):
        """commit_wires with the coefficients supplied (device pointers): the sharded proof's polynomial-parallel iNTT"""
        cap = self._cap()
        self._check(self.lib.lcp2_commit_wires_coeffs(self.handle, ctypes.c_void_p(wires_ptr), ctypes.c_void_p(coeffs_ptr), _ptr(cap)))
        return cap

    # row exchange form of a sharded proof (include/lcp2.h): the rank holds the witness values of its row block only
    def commit_wires_rows(self, rows_ptr, coeffs_ptr):
        cap = self._cap()
        self._check(self.lib.lcp2_commit_wires_rows(self.handle, ctypes.c_void_p(rows_ptr), ctypes.c_void_p(coeffs_ptr), _ptr(cap)))
        return cap

    def commit_wires_rows_begin(self, rows_ptr):
        self._check(self.lib.lcp2_commit_wires_rows_begin(self.handle, ctypes.c_void_p(rows_ptr)))

    def commit_wires_chunk(self, coeffs_ptr, first_col, ncols):
        self._check(self.lib.lcp2_commit_wires_chunk(self.handle, ctypes.c_void_p(coeffs_ptr), first_col, ncols))

    def commit_wires_rows_finish(self):
        cap = self._cap()
        self._check(self.lib.lcp2_commit_wires_rows_finish(self.handle, _ptr(cap)))
        return cap

    def perm_zs_rows_begin(self, betas, gammas, world):
        """this rank's share of the table [world][num_challenges] of row-block products"""
        out, b, g = np.zeros(world * self.circ.params.num_challenges, dtype=np.uint64), _np_u64(betas), _np_u64(gammas)
        self._check(self.lib.lcp2_perm_zs_rows_begin(self.handle, _ptr(b), _ptr(g), _ptr(out)))
        return out

    def perm_zs_rows_finish(self, block_products):
        """-> (device pointer, words) of the exchange buffer [world][columns][rows]; this rank's slot is filled"""
        bp = _np_u64(block_products)
        ptr, words = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self.lib.lcp2_perm_zs_rows_finish(self.handle, _ptr(bp), ctypes.byref(ptr), ctypes.byref(words)))
        return ptr.value, words.value

    def perm_zs_commit(self):
        cap = self._cap()
        self._check(self.lib.lcp2_perm_zs_commit(self.handle, _ptr(cap)))
        return cap

    def perm_zs(self, betas, gammas):
        cap, b, g = self._cap(), _np_u64(betas), _np_u64(gammas)
        self._check(self.lib.lcp2_perm_zs(self.handle, _ptr(b), _ptr(g), _ptr(cap)))
        return cap

    def quotient(self, alphas, public_inputs_hash):
        """compute_quotient_polys + commitment; takes public_inputs_hash (4 elements) as plonky2's function does"""
        cap, a, h = self._cap(), _np_u64(alphas), _np_u64(public_inputs_hash)
        assert h.size == 4
        self._check(self.lib.lcp2_quotient(self.handle, _ptr(a), _ptr(h), _ptr(cap)))
        return cap

    def fri_open(self, zeta, challenger_state, proof):
        """proof: uint64 array of proof_words; words from the openings on are written; challenger_state is updated"""
        z = _np_u64(zeta)
        assert proof.dtype == np.uint64 and proof.size == self.proof_words and proof.flags.c_contiguous
        self._check(self.lib.lcp2_fri_open(self.handle, _ptr(z), ctypes.byref(challenger_state), _ptr(proof)))

    # lcp2_fri_open in its three phases (the exchange points of a coset-sharded proof)
    def fri_open_begin(self, zeta, challenger_state, proof):
        z = _np_u64(zeta)
        assert proof.dtype == np.uint64 and proof.size == self.proof_words and proof.flags.c_contiguous
        self._check(self.lib.lcp2_fri_open_begin(self.handle, _ptr(z), ctypes.byref(challenger_state), _ptr(proof)))

    def fri_open_commit(self, proof):
        assert proof.dtype == np.uint64 and proof.size == self.proof_words and proof.flags.c_contiguous
        self._check(self.lib.lcp2_fri_open_commit(self.handle, _ptr(proof)))

    def fri_open_finish(self, proof, challenger_state=None):
        assert proof.dtype == np.uint64 and proof.size == self.proof_words and proof.flags.c_contiguous
        self._check(self.lib.lcp2_fri_open_finish(self.handle, ctypes.byref(challenger_state) if challenger_state is not None else None, _ptr(proof)))

    def proof_section(self, section):
        """(first word, word count) of SECTION_OPENINGS / SECTION_FRI_CAP0 / SECTION_AFTER_CAPS inside the proof array"""
        first, count = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._check(self.lib.lcp2_proof_section(self.handle, section, ctypes.byref(first), ctypes.byref(count)))
        return first.value, count.value

    def verify(self, proof, public_inputs):
        """data.verify(proof): raises ProofRejected like the reference's unwrap()"""
        failed = ctypes.c_int(0)
        pr, pis = _np_u64(proof).ravel(), _np_u64(public_inputs).ravel()
        # the lengths travel with the buffers: the library refuses a proof that is not exactly lcp2_proof_words() long
        rc = self.lib.lcp2_verify(self.handle, _ptr(pr), pr.size, _ptr(pis), pis.size, ctypes.byref(failed))
        if rc == -7:
            raise ProofRejected(failed.value)
        self._check(rc)

    def verifier_data_bytes(self):
        """VerifierOnlyCircuitData::to_bytes: constants_sigmas_cap then circuit_digest"""
        out = np.zeros(((4 << self.circ.params.cap_height) + 4) * 8, dtype=np.uint8)
        self._check(self.lib.lcp2_verifier_data_to_bytes(self.handle, _ptr(out), out.size))
        return out.tobytes()

    def last_challenges(self):
        out = np.zeros(97, dtype=np.uint64)
        self._check(self.lib.lcp2_last_challenges(self.handle, _ptr(out)))
        return {"betas": out[0:4], "gammas": out[4:8], "alphas": out[8:12], "zeta": out[12:14], "fri_alpha": out[14:16],
                "fri_betas": out[16:32].reshape(8, 2), "pow_witness": int(out[32]), "query_indices": out[33:97]}

    def close(self):
        if self.handle:
            self.lib.lcp2_circuit_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
