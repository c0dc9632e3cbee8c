"""Host-side circuit description: gate programs, selectors, copy-constraint permutation.

This is the data a `CircuitBuilder::build()` hands to the prover (plonky2 0.1.4
plonk/circuit_builder.rs, gates/selectors.rs; reference call site
eth-lc-plonky2/src/main.rs:227).  plonky2's gate objects cannot cross a C ABI, so every gate
type is described by a small constraint bytecode that the K6 kernel (and the verifier) interpret
(include/lcp2.h has the instruction format):

    word0 = op | dst << 8 | kind_a << 16 | kind_b << 20      word1 = idx_a | idx_b << 16
    op:   0 ADD  1 SUB  2 MUL  3 EMIT(a)  4 XOR  5 DBLADD  6 EMITBOOL  7 MULADD  8 SBOX (a^7)  9 PMDS (Poseidon MDS layer)
    kind: 0 REG 1 WIRE 2 CONST 3 IMM 4 PI (public_inputs_hash[idx])

The gate library below restates `eval_unfiltered` of the plonky2 gates a circuit built with
`standard_recursion_config` and no recursion contains: NoopGate, ConstantGate, PublicInputGate,
BaseSumGate<2>, ArithmeticGate and PoseidonGate ([RECALL] of the published source: the crate is not in the
reference repository, see DESIGN.md "Oracle").
"""
import ctypes

import numpy as np

from . import gl_np as gl
from . import poseidon_py as pos

OP_ADD, OP_SUB, OP_MUL, OP_EMIT, OP_XOR, OP_DBLADD, OP_EMITBOOL, OP_MULADD, OP_SBOX, OP_PMDS = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
K_REG, K_WIRE, K_CONST, K_IMM, K_PI = 0, 1, 2, 3, 4
UNUSED_SELECTOR = 0xFFFFFFFF
MAX_REGS = 64
GATE_EMIT_FORWARD = 1
# "this program is plonky2's X gate": claims that lcp2_circuit_create checks against its native evaluators (include/lcp2.h)
GATE_NATIVE_POSEIDON, GATE_NATIVE_ARITHMETIC, GATE_NATIVE_BASE_SUM2 = 0x100, 0x200, 0x300
GATE_NATIVE_MASK = 0xFF00
# Straight-line device evaluators generated offline from gate programs (tools/gen/gen_native_gates.cpp -> csrc/generated_gates_*.hpp):
# the index of each program in the generated files, = the order of host/gates.cpp's SHA-256 gates followed by the order of
# tools/gen/reference_gate_programs.txt (tests/test_generated_gates.py holds this table to the generated headers).  A GateSet claims
# LCP2_GATE_NATIVE_GENERATED(k) for a gate of one of these names; lcp2_circuit_create checks the claim against the program, so a gate
# of the same name with other parameters is refused at build(), never mis-proved (pass native=False for such a set).
GENERATED_GATE_NAMES = ("ShaAddGate", "ShaRoundAGate", "ShaRoundEGate", "ShaScheduleGate",
                        "ComparisonGate", "U32AddManyGate", "U32ArithmeticGate", "U32RangeCheckGate", "U32SubtractionGate", "CosetInterpolationGate",
                        "PoseidonMdsGate", "ReducingExtensionGate", "ReducingGate", "ArithmeticExtensionGate", "MulExtensionGate",
                        "ExponentiationGate", "RandomAccessGate")
GENERATED_GATE_INDEX = {name: k for k, name in enumerate(GENERATED_GATE_NAMES)}


def gate_native_generated(k):
    """LCP2_GATE_NATIVE_GENERATED(k) of include/lcp2.h"""
    return 0x8000 | (k << 8)


class Gate(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("selector_index", "selector_value", "group_start", "group_end",
                                               "code_offset", "code_len", "num_constraints", "flags")]


class ImmTable:
    """immediates of a gate set: single values are shared, blocks (the 12 constants of a PMDS) are contiguous"""

    def __init__(self):
        self.values, self.index = [], {}

    def get(self, value):
        value = int(value) % gl.P
        if value not in self.index:
            self.index[value] = len(self.values)
            self.values.append(value)
        return self.index[value]

    def block(self, values):
        base = len(self.values)
        self.values += [int(v) % gl.P for v in values]
        return base


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
        self.flags = 0

    def imm(self, value):
        return (K_IMM, self.imm_table.get(value))

    def _alloc(self):
        r = self.free.pop()
        self.max_reg = max(self.max_reg, r + 1)
        return r

    def reserve(self, first, count):
        """take registers [first, first + count) out of the free list (a PMDS window)"""
        for r in range(first, first + count):
            self.free.remove(r)
        self.max_reg = max(self.max_reg, first + count)

    def release(self, *ops):
        for o in ops:
            if o[0] == K_REG and o[1] not in self.free:
                self.free.append(o[1])

    def _op(self, op, a, b, dst=None):
        d = self._alloc() if dst is None else dst
        self.max_reg = max(self.max_reg, d + 1)
        self.words += [op | d << 8 | a[0] << 16 | b[0] << 20, a[1] | b[1] << 16]
        return (K_REG, d)

    def add(self, a, b, dst=None):
        return self._op(OP_ADD, a, b, dst)

    def sub(self, a, b, dst=None):
        return self._op(OP_SUB, a, b, dst)

    def mul(self, a, b, dst=None):
        return self._op(OP_MUL, a, b, dst)

    def xor(self, a, b):
        """a + b - 2ab (a ^ b on bits)"""
        return self._op(OP_XOR, a, b)

    def dbladd(self, a, b, dst=None):
        """2a + b (one Horner step of a bit recomposition)"""
        return self._op(OP_DBLADD, a, b, dst)

    def muladd(self, acc, a, b):
        """acc <- acc + a * b; acc must be a register operand"""
        assert acc[0] == K_REG
        return self._op(OP_MULADD, a, b, dst=acc[1])

    def sbox(self, a, dst):
        """reg[dst] <- a^7"""
        return self._op(OP_SBOX, a, (K_REG, 0), dst)

    def pmds(self, dst, src, constants):
        """reg[dst .. dst+12) <- MDS * reg[src .. src+12) + constants (12 immediates)"""
        base = self.imm_table.block(constants)
        self.words += [OP_PMDS | dst << 8 | K_REG << 16 | K_IMM << 20, src | base << 16]

    def emit(self, a):
        self.words += [OP_EMIT | a[0] << 16, a[1]]
        self.num_constraints += 1

    def emit_bool(self, a):
        """constraint a * a - a"""
        self.words += [OP_EMITBOOL | a[0] << 16, a[1]]
        self.num_constraints += 1


# ---------------------------------------------------------------- gate library: plonky2's gates as constraint programs
def gate_noop(asm):
    """NoopGate: no constraints (padding rows)."""


def gate_public_input(asm):
    """PublicInputGate (gates/public_input.rs): wires 0..4 = public_inputs_hash.  The public inputs themselves are hashed
    in-circuit by PoseidonGate rows whose output is copy-constrained to these wires (circuit_builder.rs::build)."""
    for i in reversed(range(4)):
        t = asm.sub(W(i), PI(i))
        asm.emit(t)
        asm.release(t)


def gate_constant(asm):
    """ConstantGate { num_consts: 2 } (gates/constant.rs): local_constants[i] - local_wires[i]."""
    for i in (1, 0):
        t = asm.sub(C(i), W(i))
        asm.emit(t)
        asm.release(t)


ARITH_OPS = 20


def gate_arithmetic(asm):
    """ArithmeticGate { num_ops: 20 } (gates/arithmetic_base.rs): output - (c0 * multiplicand_0 * multiplicand_1 + c1 * addend)
    over wires 4i .. 4i+3 = multiplicand_0, multiplicand_1, addend, output."""
    asm.flags |= GATE_NATIVE_ARITHMETIC
    for k in reversed(range(ARITH_OPS)):
        xy = asm.mul(W(4 * k), W(4 * k + 1))
        t = asm.mul(xy, C(0))
        u = asm.mul(W(4 * k + 2), C(1))
        s = asm.add(t, u)
        d = asm.sub(W(4 * k + 3), s)
        asm.emit(d)
        asm.release(xy, t, u, s, d)


BASE_SUM_LIMBS = 63


def gate_base_sum(num_limbs=BASE_SUM_LIMBS):
    """BaseSumGate<2> { num_limbs } (gates/base_sum.rs): wire 0 = sum, wires 1 .. num_limbs = little-endian bits.
    constraints: [reduce_with_powers(limbs, 2) - sum] ++ [limb * (limb - 1) for every limb]"""
    def build(asm):
        asm.flags |= GATE_NATIVE_BASE_SUM2
        for i in reversed(range(num_limbs)):
            asm.emit_bool(W(1 + i))
        acc = asm.dbladd(W(num_limbs), W(num_limbs - 1))
        for i in reversed(range(num_limbs - 2)):
            asm.dbladd(acc, W(1 + i), dst=acc[1])
        asm.sub(acc, W(0), dst=acc[1])
        asm.emit(acc)
        asm.release(acc)
    return build


def gate_poseidon(asm):
    """PoseidonGate (gates/poseidon.rs): one permutation per row, 135 wires, 123 constraints of degree 7, in the order of
    eval_unfiltered: swap booleanity, the 4 delta equations, then for every S-box that has a wire (full rounds 1-3, the 22
    partial rounds, full rounds 4-7) `state - sbox_in`, last the 12 outputs.  The constraints fall out of one forward pass
    over the rounds, so the program emits them first to last (GATE_EMIT_FORWARD).  The partial rounds are written in the
    naive form (add 12 constants, S-box on lane 0, MDS): plonky2's fast-partial-round form computes the same lane-0 values
    and is the same polynomial in the wires."""
    rc = pos.round_constants()
    asm.flags |= GATE_EMIT_FORWARD | GATE_NATIVE_POSEIDON
    asm.reserve(0, 12)  # the state window
    S = [R(i) for i in range(12)]
    asm.emit_bool(W(pos.W_SWAP))
    t = (K_REG, asm._alloc())
    for i in range(4):  # swap * (input[i+4] - input[i]) - delta_i
        asm.sub(W(pos.W_INPUT + i + 4), W(pos.W_INPUT + i), dst=t[1])
        asm.mul(t, W(pos.W_SWAP), dst=t[1])
        asm.sub(t, W(pos.W_DELTA + i), dst=t[1])
        asm.emit(t)
    # state after the swap, with the first round's constants added
    for i in range(4):
        asm.add(W(pos.W_INPUT + i), W(pos.W_DELTA + i), dst=i)
        asm.add(S[i], asm.imm(rc[i]), dst=i)
        asm.sub(W(pos.W_INPUT + i + 4), W(pos.W_DELTA + i), dst=i + 4)
        asm.add(S[i + 4], asm.imm(rc[i + 4]), dst=i + 4)
    for i in range(8, 12):
        asm.add(W(pos.W_INPUT + i), asm.imm(rc[i]), dst=i)
    rnd = 0

    def next_constants():
        return rc[12 * (rnd + 1):12 * (rnd + 2)] if rnd + 1 < pos.N_ROUNDS else [0] * 12

    for r in range(pos.N_FULL_HALF):
        for i in range(12):
            src = S[i]
            if r:
                src = W(pos.wire_full_sbox_0(r, i))
                asm.sub(S[i], src, dst=t[1])
                asm.emit(t)
            asm.sbox(src, dst=i)
        asm.pmds(0, 0, next_constants())
        rnd += 1
    for r in range(pos.N_PARTIAL):
        src = W(pos.wire_partial_sbox(r))
        asm.sub(S[0], src, dst=t[1])
        asm.emit(t)
        asm.sbox(src, dst=0)
        asm.pmds(0, 0, next_constants())
        rnd += 1
    for r in range(pos.N_FULL_HALF):
        for i in range(12):
            src = W(pos.wire_full_sbox_1(r, i))
            asm.sub(S[i], src, dst=t[1])
            asm.emit(t)
            asm.sbox(src, dst=i)
        asm.pmds(0, 0, next_constants())
        rnd += 1
    for i in range(12):
        asm.sub(S[i], W(pos.W_OUTPUT + i), dst=t[1])
        asm.emit(t)
    asm.release(t)


class GateSet:
    """Sorted gate list + plonky2's greedy selector grouping (gates/selectors.rs::selector_polynomials)."""

    def __init__(self, gates, max_degree=9, native=True):
        # gates: list of (name, degree, build_fn), already sorted by (degree, name) as plonky2 sorts its gate set
        # native: claim the generated device evaluator for every gate that has one (GENERATED_GATE_INDEX); False = interpreted
        self.names = [g[0] for g in gates]
        self.degrees = [g[1] for g in gates]
        self.imm_table = ImmTable()
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
            if native and name in GENERATED_GATE_INDEX and not (asm.flags & GATE_NATIVE_MASK):
                asm.flags |= gate_native_generated(GENERATED_GATE_INDEX[name])
            g = Gate(sel, gi, groups[sel][0], groups[sel][1], len(code) // 2, len(asm.words) // 2, asm.num_constraints, asm.flags)
            code += asm.words
            self.gates.append(g)
            self.max_regs = max(self.max_regs, asm.max_reg)
        self.code = np.array(code if code else [0, 0], dtype=np.uint32)
        self.code_len = len(code)
        self.imm = np.array(self.imm_table.values if self.imm_table.values else [0], dtype=np.uint64)

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


def standard_gateset():
    """The gate set of a plonky2 circuit without recursion under standard_recursion_config, sorted as plonky2 sorts it
    (degree, then id): two selector groups (gates of degree <= 3, PoseidonGate), so 4 constant columns."""
    return GateSet([
        ("NoopGate", 0, gate_noop),
        ("ConstantGate", 1, gate_constant),
        ("PublicInputGate", 1, gate_public_input),
        ("BaseSumGate", 2, gate_base_sum(BASE_SUM_LIMBS)),
        ("ArithmeticGate", 3, gate_arithmetic),
        ("PoseidonGate", 7, gate_poseidon),
    ])


def synthetic_circuit(params, seed, npi=4, small_values=False, extra=None, coset_shift_base=7):
    """A satisfiable circuit of 2^degree_bits rows over plonky2's own gate set with real copy constraints, laid out the way
    circuit_builder.rs::build lays a circuit out:
      row 0        PublicInputGate: wires 0..4 carry public_inputs_hash
      rows 1..     ConstantGate rows (one holds the constant 0 that pads the sponge and drives `swap`)
      next rows    PoseidonGate rows hashing the public inputs (hash_n_to_hash_no_pad: overwrite-mode sponge, rate 8), the
                   digest copy-constrained to row 0, the inputs copy-constrained to the cells that hold the public inputs
      every 16th   BaseSumGate<2> row (63 limbs) decomposing a random 63-bit value
      the rest     ArithmeticGate rows in pairs (the second row's multiplicands are the first row's outputs), the first one
                   reads the public inputs, one constant is fanned out to every 8th pair (a long permutation cycle)
      last rows    NoopGate padding
    extra (optional): more gate types in the same circuit (u32_gates.ReferenceMix: the plonky2_u32 / comparison rows the reference's
    SHA-256 and BigUint gadgets are made of).  An object with gateset() -> the GateSet of the whole circuit (it must contain the six
    gates above), assign(gate_of_row, G, rng): gives rows that would have been ArithmeticGate rows to its own gates, and
    fill(wires, gate_of_row, G, rng, link2): writes those rows' witness and their copy constraints.
    Returns (Circuit, wires [num_wires][n], public_inputs)."""
    rng = np.random.default_rng(seed)
    n = 1 << params.degree_bits
    Wn, NR = params.num_wires, params.num_routed_wires
    gs = extra.gateset() if extra is not None else standard_gateset()
    assert params.num_constants == gs.num_selectors + 2 and Wn >= pos.NUM_WIRES and NR >= 80
    G = {name: gs.index(name) for name in gs.names}
    rows = np.arange(n)
    nperm = -(-npi // 8) if npi else 0       # PoseidonGate rows of the public-input hash
    nconst = min(3, max(n - 2, 1))
    assert n >= 1 + nconst + nperm + 4, "circuit too small for the public-input hash"
    gate_of_row = np.full(n, G["ArithmeticGate"], dtype=np.int64)
    gate_of_row[rows % 16 == 5] = G["BaseSumGate"] if n >= 32 else G["ArithmeticGate"]
    gate_of_row[0] = G["PublicInputGate"]
    crow = np.arange(1, 1 + nconst)
    gate_of_row[crow] = G["ConstantGate"]
    prow = np.arange(1 + nconst, 1 + nconst + nperm)
    gate_of_row[prow] = G["PoseidonGate"]
    npad = min(4, n // 4)
    gate_of_row[n - npad:] = G["NoopGate"]
    if extra is not None:
        extra.assign(gate_of_row, G, rng)
    arith = np.nonzero(gate_of_row == G["ArithmeticGate"])[0]
    if arith.size % 2:
        gate_of_row[arith[-1]] = G["NoopGate"]
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

    def cycle(cells):  # one copy-constraint cycle through a list of (row, col) cells
        r = np.array([c[0] for c in cells])
        c = np.array([c[1] for c in cells])
        sig_row[c, r] = np.roll(r, -1)
        sig_col[c, r] = np.roll(c, -1)

    if extra is not None:
        extra.fill(wires, gate_of_row, G, rng, link2)

    # constants: wire_i = const_i on the ConstantGate rows; the last one provides the constant zero
    zero_row = int(crow[-1])
    c0[zero_row] = 0
    wires[0, crow], wires[1, crow] = c0[crow], c1[crow]
    pi_cells = [[] for _ in range(npi)]   # every cell that must equal public input k
    if arith.size:
        first, second = arith[0::2], arith[1::2]
        # the first arithmetic row reads the public inputs on its multiplicand_1 inputs
        for k in range(min(npi, ARITH_OPS)):
            wires[4 * k + 1, first[0]] = pis[k]
            pi_cells[k].append((int(first[0]), 4 * k + 1))
        # fan one constant out to the addend of op 0 of every 8th first-row: one long cycle
        fan = first[::8]
        wires[2, fan] = c0[1]
        cycle([(1, 0)] + [(int(r), 2) for r in fan])
        for k in range(ARITH_OPS):
            x, y, z = wires[4 * k, first], wires[4 * k + 1, first], wires[4 * k + 2, first]
            out = gl.add(gl.mul(gl.mul(x, y), c0[first]), gl.mul(z, c1[first]))
            wires[4 * k + 3, first] = out
            wires[4 * k, second] = out
            link2(second, 4 * k, first, 4 * k + 3)
            y2, z2 = wires[4 * k + 1, second], wires[4 * k + 2, second]
            wires[4 * k + 3, second] = gl.add(gl.mul(gl.mul(out, y2), c0[second]), gl.mul(z2, c1[second]))
    # BaseSumGate<2>: wire 0 = sum, wires 1..63 = its bits
    bs = np.nonzero(gate_of_row == G["BaseSumGate"])[0]
    if bs.size:
        val = rng.integers(0, 256 if small_values else 1 << 63, size=bs.size, dtype=np.uint64)
        wires[0, bs] = val
        for i in range(BASE_SUM_LIMBS):
            wires[1 + i, bs] = (val >> np.uint64(i)) & np.uint64(1)
    # the public-input hash, in-circuit: one PoseidonGate row per chunk of 8 inputs
    zero_cells = [(zero_row, 0)]
    state = [0] * 12
    prev_row = None
    for j, r in enumerate(prow):
        r = int(r)
        chunk = [int(v) for v in pis[8 * j:8 * j + 8]]
        state[:len(chunk)] = chunk
        row = pos.gate_row(state, 0)
        wires[:pos.NUM_WIRES, r] = np.array(row, dtype=np.uint64)
        for i in range(12):
            if i < len(chunk):
                pi_cells[8 * j + i].append((r, pos.W_INPUT + i))
            elif prev_row is None:
                zero_cells.append((r, pos.W_INPUT + i))          # the sponge starts from the zero state
            else:
                link2(np.array([r]), pos.W_INPUT + i, np.array([prev_row]), pos.W_OUTPUT + i)  # lanes the chunk does not overwrite
        zero_cells.append((r, pos.W_SWAP))
        state = row[pos.W_OUTPUT:pos.W_OUTPUT + 12]
        prev_row = r
    pi_hash = pos.hash_no_pad(pis)
    wires[:4, 0] = np.array(pi_hash, dtype=np.uint64)
    if prev_row is not None:
        assert list(state[:4]) == pi_hash
        for i in range(4):
            link2(np.array([0]), i, np.array([prev_row]), pos.W_OUTPUT + i)
    if len(zero_cells) > 1:
        cycle(zero_cells)
    for cells in pi_cells:
        if len(cells) > 1:
            cycle(cells)
    k_is = gl.powers(coset_shift_base, NR)  # plonky2: 7^j; any base whose powers lie in distinct cosets of the subgroup serves
    sig = sigma_values(sig_row, sig_col, k_is, params.degree_bits)
    consts = np.concatenate([gs.selector_columns(gate_of_row), c0[None, :], c1[None, :]])
    cs = np.concatenate([consts, sig])
    return Circuit(params, gs, cs, k_is, npi), wires, pis


def tag_witness(wires, tag):
    """A different witness of the same synthetic circuit: the last row is NoopGate padding, so its cells are free (no gate
    constraint, identity permutation).  Used to give every rank / update of a batch a proof of its own."""
    wires[100:, -1] = np.uint64(int(tag) % gl.P)
    return wires
