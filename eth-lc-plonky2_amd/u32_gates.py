"""The gates the REFERENCE's circuit is really made of, as constraint programs: plonky2_u32's U32ArithmeticGate, U32AddManyGate,
U32SubtractionGate, U32RangeCheckGate and ComparisonGate - what plonky2_crypto's `two_to_one_sha256`, `add_virtual_biguint_target`,
`cmp_biguint`, `div_rem_biguint` lower to (reference call sites: eth-lc-plonky2/src/merkle_tree_gadget.rs:37,57,77-79,
src/targets.rs:184-235,304-332, src/utils.rs:76-113) - and plonky2's CosetInterpolationGate, which `verify_proof`
(src/targets.rs:468-470) places once per FRI layer and query.

[RECALL] of the published sources (plonky2_u32 0.1 gates/{arithmetic_u32, add_many_u32, subtraction_u32, range_check_u32, comparison}.rs,
plonky2 0.1.4 gates/coset_interpolation.rs): wire layouts, the order of eval_unfiltered and num_constraints().  Neither crate is in
this container (un-vendored git dependencies, /root/reference/Cargo.lock:2388-2390, 2347-2350), so the layouts are PARITY UNPINNED; what the
tests pin is that each program vanishes exactly on the rows its generator fills, has the constraint count of the gate's
num_constraints() formula, and that a circuit of these gates proves on the GPU to the oracle's proof word for word.

On the device the programs run as generated straight-line evaluators (tools/gen/gen_native_gates.cpp -> csrc/generated_gates_u32*.hpp; the
claim is checked against the program at build()), with `native=False` through the interpreter of K6; the host verifier interprets them.
"""
import numpy as np

from . import gl_np as gl
from .circuit import (GATE_EMIT_FORWARD, K_REG, Circuit, GateSet, W, gate_noop, sigma_values)

P = gl.P
EXT_W = 7

U32_ARITH_OPS = 3        # min(num_wires / (6 + 32), num_routed / 6): multiplicand_0, multiplicand_1, addend, output_low, output_high, inverse
U32_ARITH_LIMBS = 32     # 64 output bits in 2-bit limbs
ADD_MANY_ADDENDS = 3
ADD_MANY_OPS = 5         # min(num_routed / (addends + 3), num_wires / (addends + 3 + 18))
ADD_MANY_RESULT_LIMBS, ADD_MANY_CARRY_LIMBS = 16, 2
SUB_OPS = 6              # min(num_routed / 5, num_wires / (5 + 16))
SUB_LIMBS = 16
RANGE_INPUTS = 7         # 7 x (1 + 16) = 119 wires
RANGE_AUX = 16
CMP_BITS, CMP_CHUNKS = 32, 16
CMP_CHUNK_BITS = CMP_BITS // CMP_CHUNKS
COSET_BITS, COSET_DEGREE = 4, 8    # arity-16 FRI layers; degree = the quotient degree factor
COSET_POINTS = 1 << COSET_BITS
COSET_INTERMEDIATES = (COSET_POINTS - 2) // (COSET_DEGREE - 1)


def _range_product(asm, limb, base=4):
    """prod_{k < base} (limb - k): the limb is a base-`base` digit"""
    acc = asm.sub(limb, asm.imm(1))
    asm.mul(acc, limb, dst=acc[1])
    for k in range(2, base):
        t = asm.sub(limb, asm.imm(k))
        asm.mul(acc, t, dst=acc[1])
        asm.release(t)
    asm.emit(acc)
    asm.release(acc)


def _horner(asm, limbs_high_to_low, base):
    """sum limb_j base^j over limbs given from the most significant down -> a fresh register"""
    b = asm.imm(base)
    acc = asm.add(limbs_high_to_low[0], asm.imm(0))
    for limb in limbs_high_to_low[1:]:
        asm.mul(acc, b, dst=acc[1])
        asm.add(acc, limb, dst=acc[1])
    return acc


# ---------------------------------------------------------------- plonky2_u32 gates
def gate_u32_arithmetic(asm):
    """U32ArithmeticGate { num_ops: 3 }.  Per operation i: routed wires 6i .. 6i+5 = multiplicand_0, multiplicand_1, addend, output_low,
    output_high, inverse; 32 two-bit limbs of the output at 6 num_ops + 32 i + j (the low 16 recompose output_low).  Constraints per
    operation, in the order of eval_unfiltered (36; num_constraints = num_ops (4 + num_limbs)):
      (inverse (u32::MAX - output_high) - 1) output_low        canonicity: high = 2^32 - 1 forces low = 0
      output_high 2^32 + output_low - (m0 m1 + addend)
      prod_{k<4} (limb_j - k) for j = 31 .. 0
      sum_{j<16} limb_j 4^j - output_low ;  sum_{j>=16} limb_j 4^(j-16) - output_high"""
    asm.flags |= GATE_EMIT_FORWARD
    for i in range(U32_ARITH_OPS):
        m0, m1, addend, lo, hi, inv = (W(6 * i + k) for k in range(6))
        limbs = [W(6 * U32_ARITH_OPS + U32_ARITH_LIMBS * i + j) for j in range(U32_ARITH_LIMBS)]
        d = asm.sub(asm.imm(0xFFFFFFFF), hi)
        asm.mul(d, inv, dst=d[1])
        asm.sub(d, asm.imm(1), dst=d[1])
        asm.mul(d, lo, dst=d[1])
        asm.emit(d)
        asm.release(d)
        c = asm.mul(hi, asm.imm(1 << 32))
        asm.add(c, lo, dst=c[1])
        t = asm.mul(m0, m1)
        asm.add(t, addend, dst=t[1])
        asm.sub(c, t, dst=c[1])
        asm.emit(c)
        asm.release(c, t)
        for j in reversed(range(U32_ARITH_LIMBS)):
            _range_product(asm, limbs[j])
        mid = U32_ARITH_LIMBS // 2
        for part, word in ((limbs[:mid], lo), (limbs[mid:], hi)):
            acc = _horner(asm, list(reversed(part)), 4)
            asm.sub(acc, word, dst=acc[1])
            asm.emit(acc)
            asm.release(acc)


def gate_u32_add_many(asm):
    """U32AddManyGate { num_addends: 3, num_ops: 5 }.  Per operation i: routed wires 6i .. 6i+5 = addend_0..2, carry (in), output_result,
    output_carry; 18 two-bit limbs at 6 num_ops + 18 i + j (16 of the result, 2 of the carry).  Constraints per operation (21):
      output_carry 2^32 + output_result - (sum addends + carry)
      prod_{k<4} (limb_j - k) for j = 17 .. 0
      sum_{j<16} limb_j 4^j - output_result ;  sum_{j>=16} limb_j 4^(j-16) - output_carry"""
    asm.flags |= GATE_EMIT_FORWARD
    per = ADD_MANY_ADDENDS + 3
    nl = ADD_MANY_RESULT_LIMBS + ADD_MANY_CARRY_LIMBS
    for i in range(ADD_MANY_OPS):
        addends = [W(per * i + k) for k in range(ADD_MANY_ADDENDS)]
        carry_in, out, out_carry = W(per * i + ADD_MANY_ADDENDS), W(per * i + ADD_MANY_ADDENDS + 1), W(per * i + ADD_MANY_ADDENDS + 2)
        limbs = [W(per * ADD_MANY_OPS + nl * i + j) for j in range(nl)]
        c = asm.mul(out_carry, asm.imm(1 << 32))
        asm.add(c, out, dst=c[1])
        for a in addends + [carry_in]:
            asm.sub(c, a, dst=c[1])
        asm.emit(c)
        asm.release(c)
        for j in reversed(range(nl)):
            _range_product(asm, limbs[j])
        for part, word in ((limbs[:ADD_MANY_RESULT_LIMBS], out), (limbs[ADD_MANY_RESULT_LIMBS:], out_carry)):
            acc = _horner(asm, list(reversed(part)), 4)
            asm.sub(acc, word, dst=acc[1])
            asm.emit(acc)
            asm.release(acc)


def gate_u32_subtraction(asm):
    """U32SubtractionGate { num_ops: 6 }.  Per operation i: routed wires 5i .. 5i+4 = x, y, borrow (in), output_result, output_borrow; 16
    two-bit limbs of the result at 5 num_ops + 16 i + j.  Constraints per operation (19):
      output_result - (x - y - borrow + 2^32 output_borrow)
      prod_{k<4} (limb_j - k) for j = 15 .. 0 ;  sum limb_j 4^j - output_result ;  output_borrow (1 - output_borrow)"""
    asm.flags |= GATE_EMIT_FORWARD
    for i in range(SUB_OPS):
        x, y, borrow, out, out_borrow = (W(5 * i + k) for k in range(5))
        limbs = [W(5 * SUB_OPS + SUB_LIMBS * i + j) for j in range(SUB_LIMBS)]
        t = asm.mul(out_borrow, asm.imm(1 << 32))
        asm.add(t, x, dst=t[1])
        asm.sub(t, y, dst=t[1])
        asm.sub(t, borrow, dst=t[1])
        asm.sub(out, t, dst=t[1])
        asm.emit(t)
        asm.release(t)
        for j in reversed(range(SUB_LIMBS)):
            _range_product(asm, limbs[j])
        acc = _horner(asm, list(reversed(limbs)), 4)
        asm.sub(acc, out, dst=acc[1])
        asm.emit(acc)
        asm.release(acc)
        nb = asm.sub(asm.imm(1), out_borrow)
        asm.mul(nb, out_borrow, dst=nb[1])
        asm.emit(nb)
        asm.release(nb)


def gate_u32_range_check(asm):
    """U32RangeCheckGate { num_input_limbs: 7 }.  Inputs on wires 0..6 (routed), 16 two-bit aux limbs of input i at 7 + 16 i + j.
    Constraints per input (17): sum aux_j 4^j - input, then prod_{k<4} (aux_j - k) for j = 0 .. 15"""
    asm.flags |= GATE_EMIT_FORWARD
    for i in range(RANGE_INPUTS):
        aux = [W(RANGE_INPUTS + RANGE_AUX * i + j) for j in range(RANGE_AUX)]
        acc = _horner(asm, list(reversed(aux)), 4)
        asm.sub(acc, W(i), dst=acc[1])
        asm.emit(acc)
        asm.release(acc)
        for a in aux:
            _range_product(asm, a)


def _cmp_wires():
    nc = CMP_CHUNKS
    first, second, result, msd = W(0), W(1), W(2), W(3)
    fc = [W(4 + i) for i in range(nc)]
    sc = [W(4 + nc + i) for i in range(nc)]
    dummy = [W(4 + 2 * nc + i) for i in range(nc)]
    eq = [W(4 + 3 * nc + i) for i in range(nc)]
    inter = [W(4 + 4 * nc + i) for i in range(nc)]
    bits = [W(4 + 5 * nc + i) for i in range(CMP_CHUNK_BITS + 1)]
    return first, second, result, msd, fc, sc, dummy, eq, inter, bits


def gate_comparison(asm):
    """ComparisonGate { num_bits: 32, num_chunks: 16 } (first <= second).  Wires: first_input, second_input, result_bool,
    most_significant_diff, then 16 each of first chunks, second chunks, equality dummies, chunks_equal flags, intermediate values, then
    chunk_bits + 1 = 3 bits of 2^chunk_bits + most_significant_diff.  Constraints (6 + 5 num_chunks + chunk_bits = 88):
      recomposition of both inputs from their chunks; per chunk, least significant first: both chunks are base-4 digits,
      difference dummy - (1 - equal), equal difference, intermediate - equal msd_so_far (msd_so_far <- intermediate + (1 - equal) difference);
      most_significant_diff - msd_so_far; the 3 bits are boolean and recompose to 4 + most_significant_diff; result_bool - top bit"""
    asm.flags |= GATE_EMIT_FORWARD
    first, second, result, msd, fc, sc, dummy, eq, inter, bits = _cmp_wires()
    for chunks, word in ((fc, first), (sc, second)):
        acc = _horner(asm, list(reversed(chunks)), 1 << CMP_CHUNK_BITS)
        asm.sub(acc, word, dst=acc[1])
        asm.emit(acc)
        asm.release(acc)
    one = asm.imm(1)
    so_far = asm.add(asm.imm(0), asm.imm(0))
    for i in range(CMP_CHUNKS):
        _range_product(asm, fc[i], 1 << CMP_CHUNK_BITS)
        _range_product(asm, sc[i], 1 << CMP_CHUNK_BITS)
        diff = asm.sub(sc[i], fc[i])
        t = asm.mul(diff, dummy[i])
        asm.sub(t, one, dst=t[1])
        asm.add(t, eq[i], dst=t[1])
        asm.emit(t)
        asm.mul(eq[i], diff, dst=t[1])
        asm.emit(t)
        asm.mul(eq[i], so_far, dst=t[1])
        asm.sub(inter[i], t, dst=t[1])
        asm.emit(t)
        asm.sub(one, eq[i], dst=t[1])
        asm.mul(t, diff, dst=t[1])
        asm.add(t, inter[i], dst=so_far[1])
        asm.release(t, diff)
    t = asm.sub(msd, so_far)
    asm.emit(t)
    asm.release(t, so_far)
    for b in bits:
        t = asm.sub(one, b)
        asm.mul(t, b, dst=t[1])
        asm.emit(t)
        asm.release(t)
    acc = _horner(asm, list(reversed(bits)), 2)
    t = asm.add(msd, asm.imm(1 << CMP_CHUNK_BITS))
    asm.sub(t, acc, dst=t[1])
    asm.emit(t)
    asm.release(t, acc)
    t = asm.sub(result, bits[CMP_CHUNK_BITS])
    asm.emit(t)
    asm.release(t)


# ---------------------------------------------------------------- CosetInterpolationGate
def _coset_domain():
    g = gl.root_of_unity(COSET_BITS)
    return [pow(g, i, P) for i in range(COSET_POINTS)]


def _barycentric_weights(points):
    out = []
    for i, xi in enumerate(points):
        d = 1
        for j, xj in enumerate(points):
            if i != j:
                d = d * (xi - xj) % P
        out.append(pow(d, P - 2, P))
    return out


def _coset_wires():
    shift = W(0)
    values = [(W(1 + 2 * i), W(2 + 2 * i)) for i in range(COSET_POINTS)]
    s = 1 + 2 * COSET_POINTS
    point, value = (W(s), W(s + 1)), (W(s + 2), W(s + 3))
    s += 4
    evals = [(W(s + 2 * i), W(s + 2 * i + 1)) for i in range(COSET_INTERMEDIATES)]
    prods = [(W(s + 2 * (COSET_INTERMEDIATES + i)), W(s + 2 * (COSET_INTERMEDIATES + i) + 1)) for i in range(COSET_INTERMEDIATES)]
    s += 4 * COSET_INTERMEDIATES
    shifted = (W(s), W(s + 1))
    return shift, values, point, value, evals, prods, shifted


def _ext_mul(asm, a, b):
    t0 = asm.mul(a[0], b[0])
    t1 = asm.mul(a[1], b[1])
    asm.mul(t1, asm.imm(EXT_W), dst=t1[1])
    asm.add(t0, t1, dst=t0[1])
    asm.release(t1)
    u = asm.mul(a[0], b[1])
    asm.muladd(u, a[1], b[0])
    return t0, u


def gate_coset_interpolation(asm):
    """CosetInterpolationGate { subgroup_bits: 4, degree: 8 } (D = 2): shift (wire 0), the 16 values on the coset shift * H (2 wires
    each), evaluation_point, evaluation_value, then num_intermediates = (16 - 2) / (degree - 1) = 2 intermediate (eval, prod) pairs and
    the shifted evaluation point x = evaluation_point / shift.  The interpolant is evaluated in barycentric form, chunk by chunk:
      (eval, prod) <- (eval (x - x_i) + value_i w_i prod, prod (x - x_i))  over the chunk's points, from (0, 1),
    the first chunk `degree` points long, the next ones degree - 1; between chunks the pair is pinned to an intermediate wire pair,
    which keeps the degree at 8.  Constraints (2 + 4 num_intermediates + 2 = 12, extension components one by one):
      evaluation_point - shift x ;  per chunk boundary: intermediate_eval - eval, intermediate_prod - prod ;  evaluation_value - eval"""
    asm.flags |= GATE_EMIT_FORWARD
    shift, values, point, value, evals, prods, shifted = _coset_wires()
    domain = _coset_domain()
    weights = _barycentric_weights(domain)
    for k in range(2):
        t = asm.mul(shifted[k], shift)
        asm.sub(point[k], t, dst=t[1])
        asm.emit(t)
        asm.release(t)

    def partial(lo, hi, ev, pr):
        """fold points [lo, hi); ev / pr: operand pairs, or None for (0, 1)"""
        for i in range(lo, hi):
            term = (asm.sub(shifted[0], asm.imm(domain[i])), asm.add(shifted[1], asm.imm(0)))
            wv = (asm.mul(values[i][0], asm.imm(weights[i])), asm.mul(values[i][1], asm.imm(weights[i])))
            if ev is None:          # eval = 0, prod = 1: next eval = value w, next prod = term
                ev, pr = wv, term
                continue
            ne = _ext_mul(asm, ev, term)
            add = _ext_mul(asm, wv, pr)
            for k in range(2):
                asm.add(ne[k], add[k], dst=ne[k][1])
            np_ = _ext_mul(asm, pr, term)
            asm.release(*add, *wv, *term, *[o for o in ev + pr if o[0] == K_REG])
            ev, pr = ne, np_
        return ev, pr

    ev, pr = partial(0, COSET_DEGREE, None, None)
    for i in range(COSET_INTERMEDIATES):
        for pair, got in ((evals[i], ev), (prods[i], pr)):
            for k in range(2):
                t = asm.sub(pair[k], got[k])
                asm.emit(t)
                asm.release(t)
        asm.release(*ev, *pr)
        start = 1 + (COSET_DEGREE - 1) * (i + 1)
        ev, pr = partial(start, min(start + COSET_DEGREE - 1, COSET_POINTS), evals[i], prods[i])
    for k in range(2):
        t = asm.sub(value[k], ev[k])
        asm.emit(t)
        asm.release(t)
    asm.release(*ev, *pr)


def reference_gateset(native=True):
    """sorted by (degree, name) as plonky2 sorts a gate set; three selector groups under max_degree 9.  native: claim the generated
    straight-line device evaluators (checked at build()); False: everything runs through the K6 interpreter"""
    return GateSet([
        ("NoopGate", 0, gate_noop),
        ("ComparisonGate", 4, gate_comparison),
        ("U32AddManyGate", 4, gate_u32_add_many),
        ("U32ArithmeticGate", 4, gate_u32_arithmetic),
        ("U32RangeCheckGate", 4, gate_u32_range_check),
        ("U32SubtractionGate", 4, gate_u32_subtraction),
        ("CosetInterpolationGate", 8, gate_coset_interpolation),
    ], native=native)


# ---------------------------------------------------------------- row generators (Python integers)
def _digits(x, count, base=4):
    return [(x // base ** j) % base for j in range(count)]


def row_u32_arithmetic(rng, num_wires=135, force_high_max=False):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    for i in range(U32_ARITH_OPS):
        m0, m1, addend = (int(rng.integers(0, 1 << 32)) for _ in range(3))
        if force_high_max and i == 0:
            m0 = m1 = 0xFFFFFFFF          # (2^32 - 1)^2 + addend: the high half can be at most 2^32 - 2, so the inverse exists
        out = m0 * m1 + addend
        lo, hi = out & 0xFFFFFFFF, out >> 32
        w[6 * i:6 * i + 5] = [m0, m1, addend, lo, hi]
        w[6 * i + 5] = pow((0xFFFFFFFF - hi) % P, P - 2, P)
        limbs = _digits(out, U32_ARITH_LIMBS)
        for j in range(U32_ARITH_LIMBS):
            w[6 * U32_ARITH_OPS + U32_ARITH_LIMBS * i + j] = limbs[j]
    return w


def row_u32_add_many(rng, num_wires=135):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    per, nl = ADD_MANY_ADDENDS + 3, ADD_MANY_RESULT_LIMBS + ADD_MANY_CARRY_LIMBS
    for i in range(ADD_MANY_OPS):
        addends = [int(rng.integers(0, 1 << 32)) for _ in range(ADD_MANY_ADDENDS)]
        carry = int(rng.integers(0, 4))
        total = sum(addends) + carry
        w[per * i:per * i + per] = addends + [carry, total & 0xFFFFFFFF, total >> 32]
        limbs = _digits(total & 0xFFFFFFFF, ADD_MANY_RESULT_LIMBS) + _digits(total >> 32, ADD_MANY_CARRY_LIMBS)
        for j in range(nl):
            w[per * ADD_MANY_OPS + nl * i + j] = limbs[j]
    return w


def row_u32_subtraction(rng, num_wires=135):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    for i in range(SUB_OPS):
        x, y, borrow = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32)), int(rng.integers(0, 2))
        d = x - y - borrow
        out_borrow = 1 if d < 0 else 0
        res = d + (out_borrow << 32)
        w[5 * i:5 * i + 5] = [x, y, borrow, res, out_borrow]
        for j, limb in enumerate(_digits(res, SUB_LIMBS)):
            w[5 * SUB_OPS + SUB_LIMBS * i + j] = limb
    return w


def row_u32_range_check(rng, num_wires=135):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    for i in range(RANGE_INPUTS):
        x = int(rng.integers(0, 1 << 32))
        w[i] = x
        for j, limb in enumerate(_digits(x, RANGE_AUX)):
            w[RANGE_INPUTS + RANGE_AUX * i + j] = limb
    return w


def row_comparison(rng, num_wires=135, equal=False):
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    nc = CMP_CHUNKS
    a, b = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))
    if equal:
        b = a
    fc, sc = _digits(a, nc), _digits(b, nc)
    w[0], w[1] = a, b
    so_far = 0
    for i in range(nc):
        diff = (sc[i] - fc[i]) % P
        eq = 1 if diff == 0 else 0
        w[4 + i], w[4 + nc + i] = fc[i], sc[i]
        w[4 + 2 * nc + i] = 0 if eq else pow(diff, P - 2, P)   # difference * dummy = 1 - equal
        w[4 + 3 * nc + i] = eq
        inter = eq * so_far % P
        w[4 + 4 * nc + i] = inter
        so_far = (inter + (1 - eq) * diff) % P
    w[3] = so_far
    total = ((1 << CMP_CHUNK_BITS) + so_far) % P   # the most significant difference is in (-4, 4): 4 + it is in [1, 7]
    assert total < 8
    for i in range(CMP_CHUNK_BITS + 1):
        w[4 + 5 * nc + i] = (total >> i) & 1
    w[2] = (total >> CMP_CHUNK_BITS) & 1
    assert w[2] == (1 if a <= b else 0)
    return w


def _emul(a, b):
    return ((a[0] * b[0] + EXT_W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _rand_field(rng, nonzero=False):
    return int(rng.integers(1 if nonzero else 0, P, dtype=np.uint64))


def row_coset_interpolation(rng, num_wires=135):
    """a random polynomial of degree < 16 over F_{p^2} on the coset shift * H, evaluated at a random extension point"""
    w = [int(v) for v in rng.integers(0, P, size=num_wires, dtype=np.uint64)]
    domain = _coset_domain()
    weights = _barycentric_weights(domain)
    shift = _rand_field(rng, nonzero=True)
    coeffs = [(_rand_field(rng), _rand_field(rng)) for _ in range(COSET_POINTS)]

    def poly(x):  # x: extension element
        acc = (0, 0)
        for c in reversed(coeffs):
            acc = _emul(acc, x)
            acc = ((acc[0] + c[0]) % P, (acc[1] + c[1]) % P)
        return acc

    w[0] = shift
    values = [poly((shift * d % P, 0)) for d in domain]
    for i, v in enumerate(values):
        w[1 + 2 * i], w[2 + 2 * i] = v
    s = 1 + 2 * COSET_POINTS
    point = (_rand_field(rng), _rand_field(rng))
    inv_shift = pow(shift, P - 2, P)
    x = (point[0] * inv_shift % P, point[1] * inv_shift % P)
    w[s], w[s + 1] = point
    ev, pr = (0, 0), (1, 0)
    chunks = [(0, COSET_DEGREE)] + [(1 + (COSET_DEGREE - 1) * (i + 1), min(1 + (COSET_DEGREE - 1) * (i + 2), COSET_POINTS)) for i in range(COSET_INTERMEDIATES)]
    base = s + 4
    for ci, (lo, hi) in enumerate(chunks):
        for i in range(lo, hi):
            term = ((x[0] - domain[i]) % P, x[1])
            wv = (values[i][0] * weights[i] % P, values[i][1] * weights[i] % P)
            e1, e2 = _emul(ev, term), _emul(wv, pr)
            ev, pr = ((e1[0] + e2[0]) % P, (e1[1] + e2[1]) % P), _emul(pr, term)
        if ci < COSET_INTERMEDIATES:
            w[base + 2 * ci], w[base + 2 * ci + 1] = ev
            w[base + 2 * (COSET_INTERMEDIATES + ci)], w[base + 2 * (COSET_INTERMEDIATES + ci) + 1] = pr
    w[s + 2], w[s + 3] = ev
    assert ev == poly(point), "barycentric evaluation disagrees with Horner"
    end = base + 4 * COSET_INTERMEDIATES
    w[end], w[end + 1] = x
    return w


ROW_GENERATORS = {
    "U32ArithmeticGate": row_u32_arithmetic, "U32AddManyGate": row_u32_add_many, "U32SubtractionGate": row_u32_subtraction,
    "U32RangeCheckGate": row_u32_range_check, "ComparisonGate": row_comparison, "CosetInterpolationGate": row_coset_interpolation,
}


def reference_gates_circuit(params, seed, native=True):
    """A provable circuit whose rows cycle through the six gates (no copy constraints: identity permutation, no public inputs).
    Returns (Circuit, wires [num_wires][n], public_inputs = [])."""
    rng = np.random.default_rng(seed)
    gs = reference_gateset(native)
    n, Wn, NR = 1 << params.degree_bits, params.num_wires, params.num_routed_wires
    assert params.num_constants == gs.num_selectors + 2 and Wn >= 135 and NR >= 80
    kinds = list(ROW_GENERATORS)
    gate_of_row = np.zeros(n, dtype=np.int64)  # NoopGate
    wires = np.zeros((Wn, n), dtype=np.uint64)
    for r in range(n - min(4, n // 4)):
        kind = kinds[r % len(kinds)]
        gate_of_row[r] = gs.index(kind)
        wires[:, r] = np.array(ROW_GENERATORS[kind](rng, Wn), dtype=np.uint64)
    rows = np.arange(n)
    k_is = gl.powers(7, NR)
    sig = sigma_values(np.tile(rows, (NR, 1)), np.tile(np.arange(NR)[:, None], (1, n)), k_is, params.degree_bits)
    consts = np.zeros((2, n), dtype=np.uint64)
    cs = np.concatenate([gs.selector_columns(gate_of_row), consts, sig])
    return Circuit(params, gs, cs, k_is, 0), wires, np.zeros(0, dtype=np.uint64)


# ---------------------------------------------------------------- a circuit with the gate mix of the reference's SHA-256 / BigUint gadgets
_M32 = np.uint64(0xFFFFFFFF)


def _u32(rng, size):
    return rng.integers(0, 1 << 32, size=size, dtype=np.uint64)


def _digits_into(wires, first_wire, rows, value, count):
    """two-bit limbs of `value` (uint64 array over `rows`), least significant first, on wires first_wire .. first_wire + count"""
    for j in range(count):
        wires[first_wire + j, rows] = (value >> np.uint64(2 * j)) & np.uint64(3)


def fill_u32_arithmetic(wires, rows, rng):
    """U32ArithmeticGate rows (row_u32_arithmetic, vectorised over `rows`)"""
    for i in range(U32_ARITH_OPS):
        m0, m1, addend = _u32(rng, rows.size), _u32(rng, rows.size), _u32(rng, rows.size)
        out = m0 * m1 + addend          # < 2^64: (2^32 - 1)^2 + 2^32 - 1
        lo, hi = out & _M32, out >> np.uint64(32)
        for k, v in enumerate((m0, m1, addend, lo, hi)):
            wires[6 * i + k, rows] = v
        wires[6 * i + 5, rows] = gl.inv(_M32 - hi)   # hi <= 2^32 - 2: the inverse exists
        _digits_into(wires, 6 * U32_ARITH_OPS + U32_ARITH_LIMBS * i, rows, out, U32_ARITH_LIMBS)


def fill_u32_add_many(wires, rows, rng, first_addend=None):
    """U32AddManyGate rows; first_addend[i] (optional): the value of addend_0 of operation i.  Returns the output_result of every operation."""
    per, nl = ADD_MANY_ADDENDS + 3, ADD_MANY_RESULT_LIMBS + ADD_MANY_CARRY_LIMBS
    results = []
    for i in range(ADD_MANY_OPS):
        addends = [_u32(rng, rows.size) for _ in range(ADD_MANY_ADDENDS)]
        if first_addend is not None:
            addends[0] = first_addend[i]
        carry = rng.integers(0, 4, size=rows.size, dtype=np.uint64)
        total = addends[0] + addends[1] + addends[2] + carry
        res, out_carry = total & _M32, total >> np.uint64(32)
        for k, v in enumerate(addends + [carry, res, out_carry]):
            wires[per * i + k, rows] = v
        _digits_into(wires, per * ADD_MANY_OPS + nl * i, rows, res, ADD_MANY_RESULT_LIMBS)
        _digits_into(wires, per * ADD_MANY_OPS + nl * i + ADD_MANY_RESULT_LIMBS, rows, out_carry, ADD_MANY_CARRY_LIMBS)
        results.append(res)
    return results


def fill_u32_subtraction(wires, rows, rng):
    for i in range(SUB_OPS):
        x, y = _u32(rng, rows.size), _u32(rng, rows.size)
        borrow = rng.integers(0, 2, size=rows.size, dtype=np.uint64)
        out_borrow = (x < y + borrow).astype(np.uint64)
        res = (x + (out_borrow << np.uint64(32))) - y - borrow
        for k, v in enumerate((x, y, borrow, res, out_borrow)):
            wires[5 * i + k, rows] = v
        _digits_into(wires, 5 * SUB_OPS + SUB_LIMBS * i, rows, res, SUB_LIMBS)


def fill_u32_range_check(wires, rows, rng):
    for i in range(RANGE_INPUTS):
        x = _u32(rng, rows.size)
        wires[i, rows] = x
        _digits_into(wires, RANGE_INPUTS + RANGE_AUX * i, rows, x, RANGE_AUX)


def fill_comparison(wires, rows, rng):
    """ComparisonGate rows (row_comparison, vectorised): the chunk differences are in (-4, 4), so every intermediate is a small
    signed integer; every eighth row compares equal inputs"""
    nc = CMP_CHUNKS
    a, b = _u32(rng, rows.size), _u32(rng, rows.size)
    b[::8] = a[::8]
    wires[0, rows], wires[1, rows] = a, b
    inv_of = {d: pow(d % P, P - 2, P) for d in (-3, -2, -1, 1, 2, 3)}
    inv_table = np.array([inv_of.get(d, 0) for d in range(-3, 4)], dtype=np.uint64)   # index diff + 3
    so_far = np.zeros(rows.size, dtype=np.int64)

    def field(v):  # small signed integers -> field elements
        return np.where(v < 0, np.uint64(P) - (-v).astype(np.uint64), v.astype(np.uint64))

    for i in range(nc):
        fc = ((a >> np.uint64(2 * i)) & np.uint64(3)).astype(np.int64)
        sc = ((b >> np.uint64(2 * i)) & np.uint64(3)).astype(np.int64)
        diff = sc - fc
        eq = (diff == 0).astype(np.int64)
        wires[4 + i, rows], wires[4 + nc + i, rows] = fc.astype(np.uint64), sc.astype(np.uint64)
        wires[4 + 2 * nc + i, rows] = inv_table[diff + 3]
        wires[4 + 3 * nc + i, rows] = eq.astype(np.uint64)
        inter = eq * so_far
        wires[4 + 4 * nc + i, rows] = field(inter)
        so_far = inter + (1 - eq) * diff
    wires[3, rows] = field(so_far)
    total = so_far + (1 << CMP_CHUNK_BITS)   # in [1, 7]
    for i in range(CMP_CHUNK_BITS + 1):
        wires[4 + 5 * nc + i, rows] = ((total >> i) & 1).astype(np.uint64)
    wires[2, rows] = ((total >> CMP_CHUNK_BITS) & 1).astype(np.uint64)


class ReferenceMix:
    """`extra` of circuit.synthetic_circuit: the gate mix of a circuit built from the reference's own gadgets - plonky2_crypto's
    two_to_one_sha256 (reference call sites src/merkle_tree_gadget.rs:37,57,77-79) lowers to U32AddMany rows, bit decompositions
    (BaseSumGate<2>) and ArithmeticGate rows for the bitwise operations; the BigUint slot logic (src/targets.rs:184-235,304-332)
    to U32Arithmetic / U32Subtraction / U32RangeCheck / Comparison rows; build() adds the PoseidonGate rows of the public-input hash.
    Of every 16 rows: 6 U32AddMany (in pairs: the second row's first addends are copy-constrained to the first row's results),
    2 U32Arithmetic, 1 U32RangeCheck, 1 U32Subtraction, 1 Comparison, 1 BaseSum, 4 Arithmetic.  The gate set sorts as plonky2 sorts
    it (degree, name): three selector groups, 5 constant columns."""
    PATTERN = {0: "U32AddManyGate", 1: "U32AddManyGate", 2: "U32AddManyGate", 3: "U32AddManyGate", 4: "U32AddManyGate", 6: "U32AddManyGate",
               7: "U32ArithmeticGate", 8: "U32ArithmeticGate", 9: "U32RangeCheckGate", 10: "U32SubtractionGate", 11: "ComparisonGate"}

    def __init__(self, native=True):
        self.native = native

    def gateset(self):
        from .circuit import (BASE_SUM_LIMBS, gate_arithmetic, gate_base_sum, gate_constant, gate_poseidon, gate_public_input)
        return GateSet([
            ("NoopGate", 0, gate_noop),
            ("ConstantGate", 1, gate_constant),
            ("PublicInputGate", 1, gate_public_input),
            ("BaseSumGate", 2, gate_base_sum(BASE_SUM_LIMBS)),
            ("ArithmeticGate", 3, gate_arithmetic),
            ("ComparisonGate", 4, gate_comparison),
            ("U32AddManyGate", 4, gate_u32_add_many),
            ("U32ArithmeticGate", 4, gate_u32_arithmetic),
            ("U32RangeCheckGate", 4, gate_u32_range_check),
            ("U32SubtractionGate", 4, gate_u32_subtraction),
            ("PoseidonGate", 7, gate_poseidon),
        ], native=self.native)

    def assign(self, gate_of_row, G, rng):
        rows = np.arange(gate_of_row.size)
        free = gate_of_row == G["ArithmeticGate"]
        for r, name in self.PATTERN.items():
            gate_of_row[free & (rows % 16 == r)] = G[name]

    def fill(self, wires, gate_of_row, G, rng, link2):
        am = np.nonzero(gate_of_row == G["U32AddManyGate"])[0]
        if am.size % 2:   # the row without a partner stands alone
            fill_u32_add_many(wires, am[-1:], rng)
            am = am[:-1]
        first, second = am[0::2], am[1::2]
        if first.size:
            res = fill_u32_add_many(wires, first, rng)
            fill_u32_add_many(wires, second, rng, first_addend=res)
            per = ADD_MANY_ADDENDS + 3
            for i in range(ADD_MANY_OPS):
                link2(second, per * i, first, per * i + ADD_MANY_ADDENDS + 1)
        for name, fn in (("U32ArithmeticGate", fill_u32_arithmetic), ("U32RangeCheckGate", fill_u32_range_check),
                         ("U32SubtractionGate", fill_u32_subtraction), ("ComparisonGate", fill_comparison)):
            rows = np.nonzero(gate_of_row == G[name])[0]
            if rows.size:
                fn(wires, rows, rng)


def reference_mix_circuit(params, seed, native=True, small_values=False):
    """circuit.synthetic_circuit with the reference's gate mix (ReferenceMix): (Circuit, wires, public_inputs); params.num_constants = 5"""
    from .circuit import synthetic_circuit
    return synthetic_circuit(params, seed, small_values=small_values, extra=ReferenceMix(native))
