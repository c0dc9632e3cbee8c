"""Host-side circuit description: gate programs, selectors, copy-constraint permutation.

This is the data a `CircuitBuilder::build()` hands to the prover (plonky2 0.1.4
plonk/circuit_builder.rs, gates/selectors.rs; reference call site
eth-lc-plonky2/src/main.rs:227).  The real gate set of plonky2 / plonky2_crypto is not
visible from the reference (un-vendored crates), so gates are described by a small
constraint bytecode that the K6 kernel (and the verifier) interpret:

    word0 = op | dst << 8 | kind_a << 16 | kind_b << 20      word1 = idx_a | idx_b << 16
    op:   0 ADD  1 SUB  2 MUL  3 EMIT(a)        kind: 0 REG 1 WIRE 2 CONST 3 IMM 4 PI
    EMIT folds a constraint into the running  acc <- acc * alpha + a , so a gate lists its
    constraints from the last to the first.
"""
import ctypes

import numpy as np

from . import gl_np as gl

OP_ADD, OP_SUB, OP_MUL, OP_EMIT, OP_XOR, OP_DBLADD, OP_EMITBOOL, OP_MULADD = 0, 1, 2, 3, 4, 5, 6, 7
K_REG, K_WIRE, K_CONST, K_IMM, K_PI = 0, 1, 2, 3, 4
UNUSED_SELECTOR = 0xFFFFFFFF
MAX_REGS = 64


class Gate(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("selector_index", "selector_value", "group_start", "group_end",
                                               "code_offset", "code_len", "num_constraints")]


def R(i):
    return (K_REG, i)


def W(i):
    return (K_WIRE, i)


def C(i):
    return (K_CONST, i)


def PI(i):
    return (K_PI, i)


class GateAsm:
    """Assembles one gate's constraint program.  Temporaries come from a free list of registers."""

    def __init__(self, imm_table):
        self.words = []
        self.imm_table = imm_table
        self.free = list(range(MAX_REGS - 1, -1, -1))
        self.max_reg = 0
        self.num_constraints = 0

    def imm(self, value):
        value = int(value) % gl.P
        if value not in self.imm_table:
            self.imm_table[value] = len(self.imm_table)
        return (K_IMM, self.imm_table[value])

    def _alloc(self):
        r = self.free.pop()
        self.max_reg = max(self.max_reg, r + 1)
        return r

    def release(self, *ops):
        for o in ops:
            if o[0] == K_REG and o[1] not in self.free:
                self.free.append(o[1])

    def _op(self, op, a, b, dst=None):
        d = self._alloc() if dst is None else dst
        self.words += [op | d << 8 | a[0] << 16 | b[0] << 20, a[1] | b[1] << 16]
        return (K_REG, d)

    def add(self, a, b):
        return self._op(OP_ADD, a, b)

    def sub(self, a, b):
        return self._op(OP_SUB, a, b)

    def mul(self, a, b):
        return self._op(OP_MUL, a, b)

    def xor(self, a, b):
        """a + b - 2ab (a ^ b on bits)"""
        return self._op(OP_XOR, a, b)

    def dbladd(self, a, b):
        """2a + b (one Horner step of a bit recomposition)"""
        return self._op(OP_DBLADD, a, b)

    def muladd(self, acc, a, b):
        """acc <- acc + a * b; acc must be a register operand"""
        assert acc[0] == K_REG
        return self._op(OP_MULADD, a, b, dst=acc[1])

    def emit(self, a):
        self.words += [OP_EMIT | a[0] << 16, a[1]]
        self.num_constraints += 1

    def emit_bool(self, a):
        """constraint a * a - a"""
        self.words += [OP_EMITBOOL | a[0] << 16, a[1]]
        self.num_constraints += 1


# ---------------------------------------------------------------- gate library (own layout)
def gate_noop(asm):
    """NoopGate: no constraints (padding rows)."""


def gate_public_input(npi):
    def build(asm):
        for i in reversed(range(npi)):  # wire_i - public_input_i
            t = asm.sub(W(i), PI(i))
            asm.emit(t)
            asm.release(t)
    return build


def gate_constant(asm):
    """ConstantGate: wire_i = const_i for the two gate constants."""
    for i in (1, 0):
        t = asm.sub(W(i), C(i))
        asm.emit(t)
        asm.release(t)


ARITH_OPS = 20


def gate_arithmetic(asm):
    """ArithmeticGate (base): c0 * x * y + c1 * z - out over 20 groups of 4 routed wires."""
    for k in reversed(range(ARITH_OPS)):
        xy = asm.mul(W(4 * k), W(4 * k + 1))
        t = asm.mul(xy, C(0))
        u = asm.mul(W(4 * k + 2), C(1))
        s = asm.add(t, u)
        d = asm.sub(s, W(4 * k + 3))
        asm.emit(d)
        asm.release(xy, t, u, s, d)


SBOX_LANES = 12


def gate_sbox7(asm):
    """Degree-7 gate: wire[80+2i+1] = wire[80+2i]^7 for 12 lanes of unrouted wires (the x^7 S-box shape of the
    Poseidon gate; exercises the maximum constraint degree the quotient domain allows)."""
    for i in reversed(range(SBOX_LANES)):
        x = W(80 + 2 * i)
        x2 = asm.mul(x, x)
        x4 = asm.mul(x2, x2)
        x3 = asm.mul(x2, x)
        x7 = asm.mul(x3, x4)
        d = asm.sub(x7, W(80 + 2 * i + 1))
        asm.emit(d)
        asm.release(x2, x4, x3, x7, d)


class GateSet:
    """Sorted gate list + plonky2's greedy selector grouping (gates/selectors.rs::selector_polynomials)."""

    def __init__(self, gates, max_degree=9):
        # gates: list of (name, degree, build_fn), already sorted by (degree, name) as plonky2 sorts its gate set
        self.names = [g[0] for g in gates]
        self.degrees = [g[1] for g in gates]
        self.imm_table = {}
        code, self.gates = [], []
        groups = []
        n = len(gates)
        if self.degrees[-1] + n - 1 <= max_degree:
            groups = [(0, n)]
        else:
            start = 0
            while start < n:
                size = 0
                while start + size < n and size + self.degrees[start + size] < max_degree:
                    size += 1
                assert size > 0, "gate degree too high for the quotient degree"
                groups.append((start, start + size))
                start += size
        self.groups = groups
        self.num_selectors = len(groups)
        self.max_regs = 1
        for gi, (name, deg, fn) in enumerate(gates):
            asm = GateAsm(self.imm_table)
            fn(asm)
            sel = next(i for i, (a, b) in enumerate(groups) if a <= gi < b)
            g = Gate(sel, gi, groups[sel][0], groups[sel][1], len(code) // 2, len(asm.words) // 2, asm.num_constraints)
            code += asm.words
            self.gates.append(g)
            self.max_regs = max(self.max_regs, asm.max_reg)
        self.code = np.array(code if code else [0, 0], dtype=np.uint32)
        self.code_len = len(code)
        imm = [0] * max(len(self.imm_table), 1)
        for v, i in self.imm_table.items():
            imm[i] = v
        self.imm = np.array(imm, dtype=np.uint64)

    def index(self, name):
        return self.names.index(name)

    def selector_columns(self, gate_of_row):
        """selector polynomial values for a row -> gate-index assignment"""
        gate_of_row = np.asarray(gate_of_row, dtype=np.uint64)
        cols = np.full((self.num_selectors, gate_of_row.size), UNUSED_SELECTOR, dtype=np.uint64)
        for s, (a, b) in enumerate(self.groups):
            m = (gate_of_row >= a) & (gate_of_row < b)
            cols[s, m] = gate_of_row[m]
        return cols


class Circuit:
    """Everything `lcp2_circuit_create` consumes (and the oracle's orc_circuit_new)."""

    def __init__(self, params, gateset, constants_sigmas, k_is, num_public_inputs):
        self.params, self.gateset = params, gateset
        self.constants_sigmas = np.ascontiguousarray(constants_sigmas, dtype=np.uint64)
        self.k_is = np.ascontiguousarray(k_is, dtype=np.uint64)
        self.num_public_inputs = num_public_inputs
        self.n = 1 << params.degree_bits

    @property
    def gates_array(self):
        return (Gate * len(self.gateset.gates))(*self.gateset.gates)


def sigma_values(sig_row, sig_col, k_is, degree_bits):
    """sigma_j(w^i) = k_{j'} * w^{i'} for the cell (i', j') that follows (i, j) in its copy cycle"""
    sub = gl.powers(gl.root_of_unity(degree_bits), 1 << degree_bits)
    k_is = np.asarray(k_is, dtype=np.uint64)
    out = np.empty(sig_row.shape, dtype=np.uint64)
    step = 1 << 16  # cache-sized pieces: the temporaries of gl.mul stay in L2
    for j in range(sig_row.shape[0]):
        for a in range(0, sig_row.shape[1], step):
            out[j, a:a + step] = gl.mul(k_is[sig_col[j, a:a + step]], sub[sig_row[j, a:a + step]])
    return out


def standard_gateset(npi):
    return GateSet([
        ("noop", 0, gate_noop),
        ("constant", 1, gate_constant),
        ("public_input", 1, gate_public_input(npi)),
        ("arithmetic", 3, gate_arithmetic),
        ("sbox7", 7, gate_sbox7),
    ])


def synthetic_circuit(params, seed, npi=4, small_values=False):
    """A satisfiable circuit of 2^degree_bits rows over the standard gate set with real copy constraints:
    arithmetic rows are paired (the second row's x inputs are the first row's outputs), the first arithmetic
    row reads the public inputs, and one constant is fanned out to many rows (a long permutation cycle).
    Returns (Circuit, wires [num_wires][n], public_inputs)."""
    rng = np.random.default_rng(seed)
    n = 1 << params.degree_bits
    Wn, NR = params.num_wires, params.num_routed_wires
    gs = standard_gateset(npi)
    assert params.num_constants == gs.num_selectors + 2
    G = {name: gs.index(name) for name in gs.names}
    rows = np.arange(n)
    gate_of_row = np.full(n, G["arithmetic"], dtype=np.int64)
    gate_of_row[0] = G["public_input"]
    nconst = min(3, max(n - 2, 1))
    gate_of_row[1:1 + nconst] = G["constant"]
    gate_of_row[rows % 16 == 5] = G["sbox7"] if n >= 32 else G["arithmetic"]
    gate_of_row[0] = G["public_input"]
    gate_of_row[1:1 + nconst] = G["constant"]
    npad = min(4, n // 4)
    gate_of_row[n - npad:] = G["noop"]
    arith = np.nonzero(gate_of_row == G["arithmetic"])[0]
    if arith.size % 2:
        gate_of_row[arith[-1]] = G["noop"]
        arith = arith[:-1]
    hi = 256 if small_values else gl.P
    wires = rng.integers(0, hi, size=(Wn, n), dtype=np.uint64)
    c0 = rng.integers(0, hi, size=n, dtype=np.uint64)
    c1 = rng.integers(0, hi, size=n, dtype=np.uint64)
    pis = rng.integers(0, gl.P, size=npi, dtype=np.uint64)
    sig_row = np.tile(rows, (NR, 1))
    sig_col = np.tile(np.arange(NR)[:, None], (1, n))

    def link2(ra, ca, rb, cb):  # 2-cycles between equal cells (vectorised)
        sig_row[ca, ra], sig_col[ca, ra] = rb, cb
        sig_row[cb, rb], sig_col[cb, rb] = ra, ca

    # public inputs
    wires[:npi, 0] = pis
    # constants
    crow = np.arange(1, 1 + nconst)
    wires[0, crow], wires[1, crow] = c0[crow], c1[crow]
    if arith.size:
        first, second = arith[0::2], arith[1::2]
        # the first arithmetic row reads the public inputs on its y inputs
        for k in range(min(npi, ARITH_OPS)):
            wires[4 * k + 1, first[0]] = pis[k]
            link2(np.array([first[0]]), 4 * k + 1, np.array([0]), k)
        # fan one constant out to the z input of op 0 of every 8th first-row: one long cycle
        fan = first[::8]
        wires[2, fan] = c0[1]
        cyc_r = np.concatenate([[1], fan])
        cyc_c = np.concatenate([[0], np.full(fan.size, 2)])
        sig_row[cyc_c, cyc_r] = np.roll(cyc_r, -1)
        sig_col[cyc_c, cyc_r] = np.roll(cyc_c, -1)
        for k in range(ARITH_OPS):
            x, y, z = wires[4 * k, first], wires[4 * k + 1, first], wires[4 * k + 2, first]
            out = gl.add(gl.mul(gl.mul(x, y), c0[first]), gl.mul(z, c1[first]))
            wires[4 * k + 3, first] = out
            wires[4 * k, second] = out
            link2(second, 4 * k, first, 4 * k + 3)
            y2, z2 = wires[4 * k + 1, second], wires[4 * k + 2, second]
            wires[4 * k + 3, second] = gl.add(gl.mul(gl.mul(out, y2), c0[second]), gl.mul(z2, c1[second]))
    sb = np.nonzero(gate_of_row == G["sbox7"])[0]
    for i in range(SBOX_LANES):
        x = wires[80 + 2 * i, sb]
        x2 = gl.mul(x, x)
        x4 = gl.mul(x2, x2)
        wires[80 + 2 * i + 1, sb] = gl.mul(gl.mul(x2, x), x4)
    k_is = gl.powers(7, NR)
    sig = sigma_values(sig_row, sig_col, k_is, params.degree_bits)
    consts = np.concatenate([gs.selector_columns(gate_of_row), c0[None, :], c1[None, :]])
    cs = np.concatenate([consts, sig])
    return Circuit(params, gs, cs, k_is, npi), wires, pis
