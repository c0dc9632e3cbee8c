"""More of plonky2's gate library as constraint programs: the gates its recursion circuits are built from
(plonky2 0.1.4 gates/arithmetic_extension.rs, multiplication_extension.rs, reducing.rs, reducing_extension.rs, random_access.rs,
exponentiation.rs, poseidon_mds.rs;
[RECALL] of the published source: wire layouts and the order of eval_unfiltered, D = 2, standard_recursion_config).  They run
on the device as generated straight-line evaluators (csrc/generated_gates_rec*.hpp; `native=False`: through the interpreter of K6) and through
the host verifier like any other program; the
recursive verifier of this repository (host/recursion.cpp) does not need them - it is made of ArithmeticGate operations - they
are here so that a fork can hand a circuit that contains them to lcp2_circuit_create.

Each gate comes with the generator that fills one of its rows (Python integers), used by the tests to check that the program
vanishes on a valid row and does not on a perturbed one, and by `recursion_gates_circuit` to build a small provable circuit.
"""
import numpy as np

from . import gl_np as gl
from .circuit import (GATE_EMIT_FORWARD, K_REG, Circuit, GateSet, W, C, gate_noop, sigma_values)

P = gl.P
EXT_W = 7  # F[X] / (X^2 - 7)

ARITH_EXT_OPS = 10       # num_routed_wires / (4 D)
MUL_EXT_OPS = 13         # num_routed_wires / (3 D)
REDUCING_COEFFS = 43     # min(num_routed - 3 D, (num_wires - 2 D) / (D + 1))
RANDOM_ACCESS_BITS, RANDOM_ACCESS_COPIES, RANDOM_ACCESS_EXTRA = 4, 4, 2
EXP_POWER_BITS = 66      # min(num_routed - 3, (num_wires - 2) / 2)
REDUCING_EXT_COEFFS = 32 # min((num_routed - 3 D) / D, (num_wires - 2 D) / (2 D))
MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11


# ---------------------------------------------------------------- extension arithmetic inside a program
def _ext_mul(asm, a, b):
    """(a0 + a1 X)(b0 + b1 X) -> two fresh registers"""
    t0 = asm.mul(a[0], b[0])
    t1 = asm.mul(a[1], b[1])
    asm.mul(t1, asm.imm(EXT_W), dst=t1[1])
    asm.add(t0, t1, dst=t0[1])
    asm.release(t1)
    u = asm.mul(a[0], b[1])
    asm.muladd(u, a[1], b[0])
    return t0, u


def _ext_scale(asm, a, s):
    return asm.mul(a[0], s), asm.mul(a[1], s)


def _emit_ext_diff(asm, x, y):
    """constraints x - y, component by component"""
    for k in range(2):
        d = asm.sub(x[k], y[k])
        asm.emit(d)
        asm.release(d)


def gate_arithmetic_extension(asm):
    """ArithmeticExtensionGate { num_ops: 10 }: wires 8i .. 8i+7 = multiplicand_0, multiplicand_1, addend, output (2 each);
    constraints output - (c0 * m0 * m1 + c1 * addend)"""
    asm.flags |= GATE_EMIT_FORWARD
    for i in range(ARITH_EXT_OPS):
        w = [(W(8 * i + 2 * k), W(8 * i + 2 * k + 1)) for k in range(4)]
        prod = _ext_mul(asm, w[0], w[1])
        for k in range(2):
            asm.mul(prod[k], C(0), dst=prod[k][1])
            asm.muladd(prod[k], w[2][k], C(1))
        _emit_ext_diff(asm, w[3], prod)
        asm.release(*prod)


def gate_mul_extension(asm):
    """MulExtensionGate { num_ops: 13 }: wires 6i .. 6i+5 = multiplicand_0, multiplicand_1, output; constraints output - c0 * m0 * m1"""
    asm.flags |= GATE_EMIT_FORWARD
    for i in range(MUL_EXT_OPS):
        w = [(W(6 * i + 2 * k), W(6 * i + 2 * k + 1)) for k in range(3)]
        prod = _ext_mul(asm, w[0], w[1])
        for k in range(2):
            asm.mul(prod[k], C(0), dst=prod[k][1])
        _emit_ext_diff(asm, w[2], prod)
        asm.release(*prod)


def _reducing_wires():
    out, alpha, old = (W(0), W(1)), (W(2), W(3)), (W(4), W(5))
    coeffs = [W(6 + i) for i in range(REDUCING_COEFFS)]
    start = 6 + REDUCING_COEFFS
    accs = [(W(start + 2 * i), W(start + 2 * i + 1)) for i in range(REDUCING_COEFFS - 1)] + [out]
    return out, alpha, old, coeffs, accs


def gate_reducing(asm):
    """ReducingGate<2> { num_coeffs: 43 }: output, alpha, old_acc (2 wires each), 43 base-field coefficients, 42 intermediate
    accumulators; constraints acc_i - (acc_{i-1} * alpha + coeff_i), acc_{-1} = old_acc, acc_42 = output"""
    asm.flags |= GATE_EMIT_FORWARD
    out, alpha, old, coeffs, accs = _reducing_wires()
    prev = old
    for i in range(REDUCING_COEFFS):
        t = _ext_mul(asm, prev, alpha)
        asm.add(t[0], coeffs[i], dst=t[0][1])
        _emit_ext_diff(asm, accs[i], t)
        asm.release(*t)
        prev = accs[i]


def gate_reducing_extension(asm):
    """ReducingExtensionGate<2> { num_coeffs: 32 }: as ReducingGate with extension-field coefficients (2 wires each)"""
    asm.flags |= GATE_EMIT_FORWARD
    out, alpha, prev = (W(0), W(1)), (W(2), W(3)), (W(4), W(5))
    start = 6 + 2 * REDUCING_EXT_COEFFS
    for i in range(REDUCING_EXT_COEFFS):
        t = _ext_mul(asm, prev, alpha)
        for k in range(2):
            asm.add(t[k], W(6 + 2 * i + k), dst=t[k][1])
        acc = (W(start + 2 * i), W(start + 2 * i + 1)) if i < REDUCING_EXT_COEFFS - 1 else out
        _emit_ext_diff(asm, acc, t)
        asm.release(*t)
        prev = acc


def gate_poseidon_mds(asm):
    """PoseidonMdsGate: 12 extension inputs (wires 2i, 2i+1), 12 extension outputs (wires 24 + 2i, 25 + 2i); constraints
    output - MDS(input).  The MDS layer is base-field linear, so it acts on each component: the PMDS instruction with zero
    constants on the components' register window; the constraints are listed output by output, as plonky2 lists them."""
    asm.flags |= GATE_EMIT_FORWARD
    asm.reserve(0, 48)
    zero = asm.imm(0)
    for k in range(2):
        for i in range(12):
            asm.add(W(2 * i + k), zero, dst=i)
        asm.pmds(12, 0, [0] * 12)
        for i in range(12):
            asm.sub(W(24 + 2 * i + k), (K_REG, 12 + i), dst=24 + 12 * k + i)
    for i in range(12):
        asm.emit((K_REG, 24 + i))
        asm.emit((K_REG, 36 + i))


def gate_random_access(asm):
    """RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2 }: per copy access_index, claimed_element and 16 list items
    (routed), then the 2 extra constants (routed), then 4 bit wires per copy.  Constraints per copy: the bits are boolean, they
    recompose to access_index, and folding the list by the bits leaves claimed_element; last constants - their wires."""
    asm.flags |= GATE_EMIT_FORWARD
    vec = 1 << RANDOM_ACCESS_BITS
    routed = (2 + vec) * RANDOM_ACCESS_COPIES + RANDOM_ACCESS_EXTRA
    for c in range(RANDOM_ACCESS_COPIES):
        base = (2 + vec) * c
        bits = [W(routed + RANDOM_ACCESS_BITS * c + i) for i in range(RANDOM_ACCESS_BITS)]
        for b in bits:
            asm.emit_bool(b)
        acc = asm.dbladd(bits[3], bits[2])
        asm.dbladd(acc, bits[1], dst=acc[1])
        asm.dbladd(acc, bits[0], dst=acc[1])
        asm.sub(acc, W(base), dst=acc[1])
        asm.emit(acc)
        asm.release(acc)
        items = [W(base + 2 + i) for i in range(vec)]
        for b in bits:
            nxt = []
            for k in range(0, len(items), 2):
                x, y = items[k], items[k + 1]
                d = asm.sub(y, x)
                asm.mul(d, b, dst=d[1])
                asm.add(d, x, dst=d[1])
                nxt.append(d)
            asm.release(*items)
            items = nxt
        asm.sub(items[0], W(base + 1), dst=items[0][1])
        asm.emit(items[0])
        asm.release(items[0])
    for i in range(RANDOM_ACCESS_EXTRA):
        d = asm.sub(C(i), W((2 + vec) * RANDOM_ACCESS_COPIES + i))
        asm.emit(d)
        asm.release(d)


def gate_exponentiation(asm):
    """ExponentiationGate { num_power_bits: 66 }: base (wire 0), 66 power bits little endian, output, 66 intermediate values;
    constraints intermediate'_i - intermediate_i with intermediate'_i = (i = 0 ? 1 : intermediate_{i-1}^2) * (bit * base + 1 - bit)
    over the bits from the top one down, last output - intermediate_65"""
    asm.flags |= GATE_EMIT_FORWARD
    n = EXP_POWER_BITS
    base, out = W(0), W(1 + n)
    inter = [W(2 + n + i) for i in range(n)]
    one = asm.imm(1)
    for i in range(n):
        bit = W(1 + (n - 1 - i))
        m = asm.sub(base, one)       # bit * base + 1 - bit = bit * (base - 1) + 1
        asm.mul(m, bit, dst=m[1])
        asm.add(m, one, dst=m[1])
        if i:
            sq = asm.mul(inter[i - 1], inter[i - 1])
            asm.mul(m, sq, dst=m[1])
            asm.release(sq)
        asm.sub(m, inter[i], dst=m[1])
        asm.emit(m)
        asm.release(m)
    d = asm.sub(out, inter[n - 1])
    asm.emit(d)
    asm.release(d)


def recursion_gateset(native=True):
    """sorted by (degree, name) as plonky2 sorts a gate set; two selector groups under max_degree 9.  native: claim the generated
    straight-line device evaluators (checked at build()); False: everything runs through the K6 interpreter"""
    return GateSet([
        ("NoopGate", 0, gate_noop),
        ("PoseidonMdsGate", 1, gate_poseidon_mds),
        ("ReducingExtensionGate", 2, gate_reducing_extension),
        ("ReducingGate", 2, gate_reducing),
        ("ArithmeticExtensionGate", 3, gate_arithmetic_extension),
        ("MulExtensionGate", 3, gate_mul_extension),
        ("ExponentiationGate", 4, gate_exponentiation),
        ("RandomAccessGate", 5, gate_random_access),
    ], native=native)


# ---------------------------------------------------------------- row generators (Python integers)
def _emul(a, b):
    return ((a[0] * b[0] + EXT_W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def row_arithmetic_extension(rng, c0, c1, num_wires=135):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    for i in range(ARITH_EXT_OPS):
        m0, m1, ad = (w[8 * i], w[8 * i + 1]), (w[8 * i + 2], w[8 * i + 3]), (w[8 * i + 4], w[8 * i + 5])
        pr = _emul(m0, m1)
        w[8 * i + 6], w[8 * i + 7] = (c0 * pr[0] + c1 * ad[0]) % P, (c0 * pr[1] + c1 * ad[1]) % P
    return w


def row_mul_extension(rng, c0, num_wires=135):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    for i in range(MUL_EXT_OPS):
        pr = _emul((w[6 * i], w[6 * i + 1]), (w[6 * i + 2], w[6 * i + 3]))
        w[6 * i + 4], w[6 * i + 5] = c0 * pr[0] % P, c0 * pr[1] % P
    return w


def row_reducing(rng, num_wires=135):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    alpha, acc = (w[2], w[3]), (w[4], w[5])
    start = 6 + REDUCING_COEFFS
    for i in range(REDUCING_COEFFS):
        t = _emul(acc, alpha)
        acc = ((t[0] + w[6 + i]) % P, t[1])
        if i < REDUCING_COEFFS - 1:
            w[start + 2 * i], w[start + 2 * i + 1] = acc
        else:
            w[0], w[1] = acc
    return w


def row_reducing_extension(rng, num_wires=135):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    alpha, acc = (w[2], w[3]), (w[4], w[5])
    start = 6 + 2 * REDUCING_EXT_COEFFS
    for i in range(REDUCING_EXT_COEFFS):
        t = _emul(acc, alpha)
        acc = ((t[0] + w[6 + 2 * i]) % P, (t[1] + w[7 + 2 * i]) % P)
        if i < REDUCING_EXT_COEFFS - 1:
            w[start + 2 * i], w[start + 2 * i + 1] = acc
        else:
            w[0], w[1] = acc
    return w


def row_poseidon_mds(rng, num_wires=135):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    for k in range(2):
        s = [w[2 * i + k] for i in range(12)]
        for r in range(12):
            w[24 + 2 * r + k] = (sum(s[(i + r) % 12] * MDS_CIRC[i] for i in range(12)) + s[r] * MDS_DIAG[r]) % P
    return w


def row_random_access(rng, c0, c1, num_wires=135):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    vec = 1 << RANDOM_ACCESS_BITS
    routed = (2 + vec) * RANDOM_ACCESS_COPIES + RANDOM_ACCESS_EXTRA
    for c in range(RANDOM_ACCESS_COPIES):
        base = (2 + vec) * c
        idx = int(rng.integers(0, vec))
        w[base] = idx
        w[base + 1] = w[base + 2 + idx]
        for i in range(RANDOM_ACCESS_BITS):
            w[routed + RANDOM_ACCESS_BITS * c + i] = (idx >> i) & 1
    w[(2 + vec) * RANDOM_ACCESS_COPIES], w[(2 + vec) * RANDOM_ACCESS_COPIES + 1] = c0, c1
    return w


def row_exponentiation(rng, num_wires=135):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    n = EXP_POWER_BITS
    base = w[0]
    power = int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 16)) << 62)
    cur = 1
    for i in range(n):
        w[1 + i] = (power >> i) & 1
    for i in range(n):
        bit = (power >> (n - 1 - i)) & 1
        cur = (cur * cur if i else 1) % P * (base if bit else 1) % P
        w[2 + n + i] = cur
    w[1 + n] = cur
    assert cur == pow(base, power, P)
    return w


def recursion_gates_circuit(params, seed, native=True):
    """A provable circuit whose rows cycle through the seven gates (no copy constraints: identity permutation, no public inputs).
    Returns (Circuit, wires [num_wires][n], public_inputs = [])."""
    rng = np.random.default_rng(seed)
    gs = recursion_gateset(native)
    n, Wn, NR = 1 << params.degree_bits, params.num_wires, params.num_routed_wires
    assert params.num_constants == gs.num_selectors + 2 and Wn >= 135 and NR >= 80
    kinds = ["ReducingGate", "ArithmeticExtensionGate", "MulExtensionGate", "ExponentiationGate", "RandomAccessGate", "ReducingExtensionGate",
             "PoseidonMdsGate"]
    gate_of_row = np.zeros(n, dtype=np.int64)  # NoopGate
    wires = np.zeros((Wn, n), dtype=np.uint64)
    c0 = rng.integers(0, P, size=n, dtype=np.uint64)
    c1 = rng.integers(0, P, size=n, dtype=np.uint64)
    for r in range(n - min(4, n // 4)):
        kind = kinds[r % len(kinds)]
        gate_of_row[r] = gs.index(kind)
        a, b = int(c0[r]), int(c1[r])
        row = {"ReducingExtensionGate": lambda: row_reducing_extension(rng, Wn), "PoseidonMdsGate": lambda: row_poseidon_mds(rng, Wn),
               "ReducingGate": lambda: row_reducing(rng, Wn), "ArithmeticExtensionGate": lambda: row_arithmetic_extension(rng, a, b, Wn),
               "MulExtensionGate": lambda: row_mul_extension(rng, a, Wn), "ExponentiationGate": lambda: row_exponentiation(rng, Wn),
               "RandomAccessGate": lambda: row_random_access(rng, a, b, Wn)}[kind]()
        wires[:, r] = np.array(row, dtype=np.uint64)
    rows = np.arange(n)
    k_is = gl.powers(7, NR)
    sig = sigma_values(np.tile(rows, (NR, 1)), np.tile(np.arange(NR)[:, None], (1, n)), k_is, params.degree_bits)
    cs = np.concatenate([gs.selector_columns(gate_of_row), c0[None, :], c1[None, :], sig])
    return Circuit(params, gs, cs, k_is, 0), wires, np.zeros(0, dtype=np.uint64)
