"""CPU emulation of the device kernels' phases (csrc/*.hpp compiled by g++) against the oracle:
checks field arithmetic, the Poseidon MDS folding and the multi-pass NTT index logic without a GPU."""
import ctypes

import numpy as np
import pytest

from oracle_lib import P, bitrev_perm, lde_leaf_order, rand_field, vp


def test_emu_field_mul(oracle, emu):
    rng = np.random.default_rng(0)
    edge = [0, 1, P - 1, 2 ** 32 - 1, 2 ** 32, 0xFFFFFFFF00000000, P + 3, 2 ** 64 - 1]
    vals = edge + [int(x) for x in rng.integers(0, 2 ** 64, size=300, dtype=np.uint64)]
    for a in vals[:60]:
        for b in vals[:60]:
            assert emu.emu_gl_mul(a, b) == (a % P) * (b % P) % P


def test_emu_shift_reduce(emu):
    """gl_shl: multiplication by 2^s as shifts and one canonical subtraction (the 16th roots of unity are +-2^s); the carry cases
    of the three-word forms: y0 + y1 and y1 + y2 overflowing 32 bits, borrow in the final subtraction, x = p - 1"""
    rng = np.random.default_rng(5)
    edge = [0, 1, P - 1, P - 2, 2 ** 32 - 1, 2 ** 32, 0xFFFFFFFF00000000, 0xBFFFFFFFC0000000, 0xBFFFFFFFFFFFFFFF, 0x7FFFFFFFFFFFFFFF,
            0xFFFFFFFE00000000, 0xFFFFFFFEFFFFFFFF, 0x00000000FFFFFFFF, 0x8000000000000000, 0xFFFF0000FFFF0000, 0x0000FFFF0000FFFF]
    vals = edge + [int(x) % P for x in rng.integers(0, 2 ** 64, size=2000, dtype=np.uint64)]
    for s in (12, 24, 32, 33, 36, 48, 60, 63, 65, 72, 84, 95):
        m = pow(2, s, P)
        for x in vals:
            assert emu.emu_gl_shl(x, s) == x * m % P, (hex(x), s)
    # plonky2 POWER_OF_TWO_GENERATOR^(2^28) = w_16 = 2^156 = -2^60: the shift table in csrc/ntt.hpp
    assert pow(1753635133440165772, 2 ** 28, P) == pow(2, 156, P) == P - 2 ** 60


def test_emu_poseidon(oracle, emu):
    rng = np.random.default_rng(1)
    for k in range(64):
        s = rand_field(rng, 12, canonical=(k % 2 == 0))
        if k == 0:
            s[:] = 0
        if k == 1:
            s[:] = P - 1
        a, b = s.copy(), s.copy()
        oracle.orc_poseidon_permute(vp(a))
        emu.emu_poseidon_permute(vp(b))
        assert (a == b).all()


def test_emu_poseidon_grouped_partial_rounds(oracle, emu):
    """csrc/poseidon.hpp runs 21 of the 22 partial rounds three at a time (A^3 y + ... with entries below 2^22, constants folded on the
    host): the portable restatement of that form must be the permutation, bit for bit, on canonical and non-canonical states"""
    assert emu.emu_poseidon_partial_max_entry() < 1 << 22
    rng = np.random.default_rng(11)
    for k in range(200):
        s = rand_field(rng, 12, canonical=(k % 2 == 0))
        if k == 0:
            s[:] = 0
        if k == 1:
            s[:] = P - 1
        if k == 2:
            s[:] = 2 ** 64 - 1
        a, b = s.copy(), s.copy()
        oracle.orc_poseidon_permute(vp(a))
        emu.emu_poseidon_permute_grouped(vp(b))
        assert (a == b).all(), k


@pytest.mark.parametrize("lg", [1, 2, 4, 9, 12, 13, 14, 17])
def test_emu_ntt_all_directions(oracle, emu, lg):
    rng = np.random.default_rng(lg)
    n, ncols = 1 << lg, 2
    x = rand_field(rng, (ncols, n), canonical=False)
    xc = x % np.uint64(P)
    perm = bitrev_perm(lg)
    out = np.zeros_like(x)
    emu.emu_ntt_forward(vp(x), vp(out), lg, ncols, 1, 0)
    ref = xc.copy()
    oracle.orc_fft_batch(vp(ref), ncols, n)
    assert (out == ref[:, perm]).all()
    back = np.zeros_like(x)
    emu.emu_ntt_inverse_natural(vp(ref), vp(back), lg, ncols)
    assert (back == xc).all()
    co = xc.copy()
    for c in range(ncols):
        oracle.orc_coset_fft(vp(co[c]), n, 7)
    cob = np.ascontiguousarray(co[:, perm])
    back2 = np.zeros_like(x)
    emu.emu_ntt_inverse_bitrev(vp(cob), vp(back2), lg, ncols, 7)
    assert (back2 == xc).all()
    if lg <= 14:
        lde = np.zeros((ncols, n * 8), dtype=np.uint64)
        emu.emu_ntt_forward(vp(x), vp(lde), lg, ncols, 7, 3)
        assert (lde == lde_leaf_order(oracle, xc)).all()


def test_emu_ntt_three_pass_plan(oracle, emu):
    # 2^23 needs two strided passes + the contiguous pass (the shape of the 8n quotient iNTT at n = 2^22 is 2^25)
    lg, n = 23, 1 << 23
    rng = np.random.default_rng(23)
    x = rand_field(rng, (1, n))
    out = np.zeros_like(x)
    emu.emu_ntt_forward(vp(x), vp(out), lg, 1, 1, 0)
    back = np.zeros_like(x)
    emu.emu_ntt_inverse_bitrev(vp(out), vp(back), lg, 1, 1)
    assert (back == x).all()
    ref = x.copy()
    oracle.orc_fft(vp(ref), n)
    perm = bitrev_perm(lg)
    assert (out[0] == ref[0][perm]).all()


@pytest.mark.parametrize("computed_scale", [False, True])
def test_emu_lde_headline_shape_takes_the_prefetching_kernel(oracle, emu, computed_scale):
    """n = 2^22 (the headline size) is the one size whose passes have the shapes of k_ntt_pass_pf (9 strided + 13 contiguous bits):
    the fused first step on the prefetched registers, the register store of the strided pass, the wave-local store of the contiguous
    pass, the coset scale from the table and computed (FMODE 1 / 2) - one column of the LDE against the oracle's textbook transform"""
    lg, n = 22, 1 << 22
    rng = np.random.default_rng(2200 + computed_scale)
    x = rand_field(rng, (1, n), canonical=False)
    lde = np.zeros((1, n * 8), dtype=np.uint64)
    assert emu.emu_ntt_forward(vp(x), vp(lde), lg, 1, 7, 3 | (0x100 if computed_scale else 0)) == 2  # both passes
    assert (lde == lde_leaf_order(oracle, x % np.uint64(P))).all()


@pytest.mark.parametrize("computed_scale", [False, True])
def test_emu_coset_forward_headline_shape(oracle, emu, computed_scale):
    # the coset NTT (one coset: z = 0) through the strided prefetching pass with a factor: table (FMODE 1) and computed (FMODE 2)
    lg, n = 22, 1 << 22
    rng = np.random.default_rng(2230 + computed_scale)
    x = rand_field(rng, (8, n))
    out = np.zeros_like(x)
    assert emu.emu_ntt_forward(vp(x), vp(out), lg, 8, 7, 0x100 if computed_scale else 0) == 2
    perm = bitrev_perm(lg)
    ref = x[3].copy()
    oracle.orc_coset_fft(vp(ref), n, 7)
    assert (out[3] == ref[perm]).all()


@pytest.mark.parametrize("computed_scale", [False, True])
def test_emu_lde_light_client_step_shape(oracle, emu, computed_scale):
    """n = 2^19, the size of the light-client step: 6 strided + 13 contiguous bits.  The strided pass takes the two-step form of
    the prefetching kernel (first step on the prefetched registers, then the last step straight from LDS to the stores); 8 columns
    are the fewest for which the launches take it"""
    lg, n = 19, 1 << 19
    rng = np.random.default_rng(1900 + computed_scale)
    x = rand_field(rng, (8, n), canonical=False)
    lde = np.zeros((8, n * 8), dtype=np.uint64)
    assert emu.emu_ntt_forward(vp(x), vp(lde), lg, 8, 7, 3 | (0x100 if computed_scale else 0)) == 2  # both passes
    for c in (0, 6):
        assert (lde[c:c + 1] == lde_leaf_order(oracle, x[c:c + 1] % np.uint64(P))).all()


def test_emu_plain_and_coset_forward_light_client_step_shape(oracle, emu):
    # the same strided shape without a factor (FMODE 0) and with the table factor of a single coset (FMODE 1 / 2)
    lg, n = 19, 1 << 19
    rng = np.random.default_rng(1919)
    x = rand_field(rng, (64, n))
    perm = bitrev_perm(lg)
    for shift, flags in ((1, 0), (7, 0), (7, 0x100)):
        out = np.zeros_like(x)
        assert emu.emu_ntt_forward(vp(x), vp(out), lg, 64, shift, flags) == 2
        ref = x[41].copy()
        if shift == 1:
            oracle.orc_fft(vp(ref), n)
        else:
            oracle.orc_coset_fft(vp(ref), n, shift)
        assert (out[41] == ref[perm]).all()


def test_emu_plain_forward_headline_shape(oracle, emu):
    # the same shapes without a coset scale (FMODE 0): 8 columns are the fewest for which the launch takes the prefetching form
    lg, n = 22, 1 << 22
    rng = np.random.default_rng(2222)
    x = rand_field(rng, (8, n))
    out = np.zeros_like(x)
    assert emu.emu_ntt_forward(vp(x), vp(out), lg, 8, 1, 0) == 2
    perm = bitrev_perm(lg)
    for c in (0, 5):
        ref = x[c].copy()
        oracle.orc_fft(vp(ref), n)
        assert (out[c] == ref[perm]).all()
