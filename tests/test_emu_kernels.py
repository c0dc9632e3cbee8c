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
