"""Vectorised Goldilocks arithmetic on numpy uint64 arrays (host-side plumbing for building
circuits / synthetic witnesses; the prover's arithmetic runs in the HIP kernels)."""
import numpy as np

P = 0xFFFFFFFF00000001
_P = np.uint64(P)
_EPS = np.uint64(0xFFFFFFFF)
_M32 = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def canon(a):
    a = np.asarray(a, dtype=np.uint64)
    return np.where(a >= _P, a - _P, a)


def add(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = a + b
        return np.where((s < a) | (s >= _P), s - _P, s)


def sub(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        d = a - b
        return np.where(a < b, d + _P, d)


def mul(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    a, b = np.broadcast_arrays(a, b)
    with np.errstate(over="ignore"):
        a0, a1 = a & _M32, a >> _S32
        b0, b1 = b & _M32, b >> _S32
        ll, lh, hl, hh = a0 * b0, a0 * b1, a1 * b0, a1 * b1
        mid = lh + hl
        mid_carry = (mid < lh).astype(np.uint64)  # 2^64 overflow of the middle sum
        lo = ll + (mid << _S32)
        lo_carry = (lo < ll).astype(np.uint64)
        hi = hh + (mid >> _S32) + (mid_carry << _S32) + lo_carry
        # reduce lo + 2^64 * hi
        hi_hi, hi_lo = hi >> _S32, hi & _M32
        t0 = lo - hi_hi
        t0 = np.where(lo < hi_hi, t0 - _EPS, t0)
        t1 = (hi_lo << _S32) - hi_lo
        r = t0 + t1
        r = np.where(r < t0, r + _EPS, r)
        return np.where(r >= _P, r - _P, r)


def pow_scalar(b, e):
    return pow(int(b) % P, int(e), P)


def inv_scalar(a):
    return pow(int(a) % P, P - 2, P)


def inv(a):
    """element-wise inverse (0 -> 0) by a^(p - 2): 63 squarings and 62 multiplications, vectorised"""
    a = np.asarray(a, dtype=np.uint64)
    r = np.ones(a.shape, dtype=np.uint64)
    base, e = a.copy(), P - 2
    while e:
        if e & 1:
            r = mul(r, base)
        e >>= 1
        if e:
            base = mul(base, base)
    return r


def root_of_unity(bits):
    return pow(pow(7, (P - 1) >> 32, P), 1 << (32 - bits), P)


def powers(base, count):
    """[base^0, base^1, ... base^(count-1)] by repeated doubling (vectorised)"""
    out = np.ones(count, dtype=np.uint64)
    if count <= 1:
        return out
    filled = 1
    step = np.uint64(int(base) % P)  # base^filled
    while filled < count:
        m = min(filled, count - filled)
        out[filled:filled + m] = mul(out[:m], step)
        step = np.uint64(pow_scalar(step, 2))
        filled += m
    return out
