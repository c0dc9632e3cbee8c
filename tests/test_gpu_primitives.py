"""GPU parity tests (run with -m gpu on an MI355X): every primitive goes through the C ABI
(include/lcp2.h) and is compared bit for bit with the oracle on the same seeded inputs."""
import hashlib
import json
import ctypes
import os

import numpy as np
import pytest

from oracle_lib import P, bitrev_perm, commit_reference, lde_leaf_order, merkle_cap, merkle_verify, rand_field, vp

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_poseidon_batch_parity(gpu_ctx, oracle):
    rng = np.random.default_rng(10)
    n = 5000  # not a multiple of the block size
    s = rand_field(rng, (n, 12), canonical=False)
    s[0] = 0
    s[1] = np.arange(12)
    s[2] = P - 1
    got = gpu_ctx.poseidon_permute_batch(s)
    want = np.zeros_like(s)
    oracle.orc_poseidon_permute_batch(vp(s), vp(want), n)
    assert (got == want).all()
    pos = json.load(open(os.path.join(G, "poseidon_kat.json")))
    for i, v in enumerate(pos["vectors"]):
        assert ["%016x" % x for x in got[i]] == v["out"]
    assert gpu_ctx.poseidon_permute_batch(np.zeros((0, 12), dtype=np.uint64)).shape == (0, 12)


def _field_mul_operands(rng):
    """operand pairs for every branch of the device multiply's reduction (gl64.hpp gl_mul_halves): with (hi : lo) = a * b,
    the borrow of lo - (hi >> 32) needs lo < 2^32 - one pair in 2^32 at random, so b is solved for from a chosen lo"""
    M = 1 << 64
    edge = [0, 1, 2, 7, P - 2, P - 1, P, P + 1, M - 1, M - 2, 1 << 63, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, 0xFFFFFFFF00000000, 0xFFFFFFFE00000001,
            0x00000001FFFFFFFF, 0xFFFFFFFF, 0x100000000, 0x8000000080000000, 0x7FFFFFFF7FFFFFFF, 0xFFFF0001, 0x11111111]
    a = [x for x in edge for _ in edge]
    b = [y for _ in edge for y in edge]
    kinds = {"borrow": 0, "borrow_then_carry": 0, "low_word_only": 0}
    for _ in range(6000):
        x = int(rng.integers(0, M, dtype=np.uint64)) | 1
        lo = int(rng.integers(0, 1 << int(rng.integers(1, 33))))  # the low 64 bits of the product, below 2^32
        y = lo * pow(x, -1, M) % M
        prod = x * y
        assert prod % M == lo
        hi = prod >> 64
        if lo < (hi >> 32):
            kinds["borrow"] += 1
            t = (lo - (hi >> 32) - 0xFFFFFFFF) % M
            kinds["borrow_then_carry"] += t + (hi & 0xFFFFFFFF) * 0xFFFFFFFF >= M
        else:
            kinds["low_word_only"] += 1
        a.append(x); b.append(y)
    assert min(kinds.values()) > 100, kinds
    for _ in range(4000):  # hi < 2^32: no borrow, and hi_hi = 0 ; and the high halves alone
        x, y = int(rng.integers(0, 1 << 48)), int(rng.integers(0, 1 << 48))
        a.append(x); b.append(y)
        a.append(x << 16 & (M - 1)); b.append(y << 32 & (M - 1))
    r = rng.integers(0, M, size=(2, 20000), dtype=np.uint64)
    return np.concatenate([np.array(a, dtype=np.uint64), r[0]]), np.concatenate([np.array(b, dtype=np.uint64), r[1]])


def test_field_arithmetic_reaches_every_reduction_branch(gpu_ctx):
    """the arithmetic all kernels share (csrc/gl64.hpp), through the C ABI, against Python integers"""
    rng = np.random.default_rng(2024)
    a, b = _field_mul_operands(rng)
    ints = lambda v: [int(x) for x in v]
    A, B = ints(a), ints(b)

    def check(op, want, second=True):
        got = gpu_ctx.field_op_batch(op, a, b if second else None)
        w = np.array(want, dtype=np.uint64)
        bad = np.nonzero(got != w)[0]
        assert bad.size == 0, (op, [(hex(A[i]), hex(B[i]), hex(int(got[i])), hex(int(w[i]))) for i in bad[:4]])

    check("mul", [x * y % P for x, y in zip(A, B)])
    check("pow7", [pow(x, 7, P) for x in A], second=False)
    check("add", [(x + y) % P for x, y in zip(A, B)])
    check("sub", [(x - y) % P for x, y in zip(A, B)])
    check("canon", [x % P for x in A], second=False)
    check("add_lazy", [(x + y) % P for x, y in zip(A, B)])
    check("sub_lazy", [(x - y) % P for x, y in zip(A, B)])
    for k in range(1, 8):
        check("shl%d" % (12 * k), [(x << (12 * k)) % P for x in A], second=False)
    check("shl32_lazy", [(x << 32) % P for x in A], second=False)
    check("mul_u32", [x * (y & 0xFFFFFFFF) % P for x, y in zip(A, B)])
    assert gpu_ctx.field_op_batch("mul", np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64)).shape == (0,)


@pytest.mark.parametrize("leaf_len", [1, 3, 4, 5, 8, 9, 16, 20, 32, 135])
@pytest.mark.parametrize("log_leaves,cap_height", [(4, 4), (5, 4), (6, 0), (9, 4), (11, 2)])
def test_merkle_cap_parity(gpu_ctx, oracle, leaf_len, log_leaves, cap_height):
    rng = np.random.default_rng(leaf_len * 100 + log_leaves)
    leaves = rand_field(rng, (1 << log_leaves, leaf_len), canonical=False)
    got = gpu_ctx.merkle_cap(leaves, cap_height)
    want = merkle_cap(oracle, leaves, cap_height)
    assert (got == want).all()


def test_merkle_cap_rejects_bad_shapes(gpu_ctx):
    import eth_lc_plonky2_amd as m
    with pytest.raises(m.Lcp2Error):
        gpu_ctx.merkle_cap(np.zeros((8, 5), dtype=np.uint64), 4)  # cap higher than the tree
    with pytest.raises(m.Lcp2Error):
        gpu_ctx.merkle_cap(np.zeros((6, 5), dtype=np.uint64), 1)  # not a power of two


@pytest.mark.parametrize("lg", [1, 2, 3, 6, 10, 11, 12, 13, 14, 16, 18])
def test_ntt_parity(gpu_ctx, oracle, lg):
    rng = np.random.default_rng(lg)
    n, ncols = 1 << lg, 3
    x = rand_field(rng, (ncols, n), canonical=False)
    xc = x % np.uint64(P)
    want = xc.copy()
    oracle.orc_fft_batch(vp(want), ncols, n)
    assert (gpu_ctx.ntt_batch(x) == want).all()
    assert (gpu_ctx.ntt_batch(want, inverse=True) == xc).all()
    co = xc.copy()
    for c in range(ncols):
        oracle.orc_coset_fft(vp(co[c]), n, 7)
    assert (gpu_ctx.ntt_batch(x, shift=7) == co).all()
    assert (gpu_ctx.ntt_batch(co, inverse=True, shift=7) == xc).all()


@pytest.mark.parametrize("lg,ncols", [(1, 2), (4, 5), (10, 3), (13, 2), (14, 2), (16, 1)])
def test_lde_parity(gpu_ctx, oracle, lg, ncols):
    rng = np.random.default_rng(lg + 50)
    c = rand_field(rng, (ncols, 1 << lg), canonical=False)
    got = gpu_ctx.lde_batch(c, 3)
    assert (got == lde_leaf_order(oracle, c % np.uint64(P), 3)).all()


def test_lde_headline_size_parity(gpu_ctx, oracle):
    """The LDE of the headline run: 2^22 coefficients, rate 8 - the strided pass over 16-element runs with the coset scale table,
    then the contiguous pass, both in the prefetching multi-slab kernel (csrc/kernels_ntt.hip k_ntt_pass_pf).  One column against
    the oracle; the slab count decides how many slabs a workgroup chains (2 for one column, 8 for four, 16 for eight), so the
    batched results must equal the single-column ones."""
    rng = np.random.default_rng(2222)
    c = rand_field(rng, (8, 1 << 22), canonical=False)
    one = gpu_ctx.lde_batch(c[0:1], 3)
    assert (one == lde_leaf_order(oracle, c[0:1] % np.uint64(P), 3)).all()
    four = gpu_ctx.lde_batch(c[0:4], 3)
    assert (four[0] == one[0]).all()
    eight = gpu_ctx.lde_batch(c, 3)
    assert (eight[0:4] == four).all()
    assert (eight[6] == gpu_ctx.lde_batch(c[6:7], 3)[0]).all()


def test_lde_light_client_step_size_parity(gpu_ctx, oracle):
    """2^19 coefficients, the size of the light-client step: 6 strided + 13 contiguous bits; from 8 columns on the strided pass
    takes the two-step form of the prefetching kernel (k_ntt_pass_pf<true, FMODE, 7>), below that the plain kernel: both against
    the oracle, and the plain / coset transforms of lcp2_ntt_batch at 64 columns (the fewest that take it without cosets)"""
    rng = np.random.default_rng(1919)
    lg, n = 19, 1 << 19
    c = rand_field(rng, (16, n), canonical=False)
    cc = c % np.uint64(P)
    two = gpu_ctx.lde_batch(c[0:2], 3)   # plain kernel
    assert (two == lde_leaf_order(oracle, cc[0:2], 3)).all()
    full = gpu_ctx.lde_batch(c, 3)       # prefetching kernel
    assert (full[0:2] == two).all()
    for k in (7, 15):
        assert (full[k:k + 1] == lde_leaf_order(oracle, cc[k:k + 1], 3)).all()
    x = rand_field(rng, (64, n), canonical=False)
    xc = x % np.uint64(P)
    for shift in (1, 7):
        got = gpu_ctx.ntt_batch(x, shift=shift)
        for k in (0, 37, 63):
            want = xc[k].copy()
            if shift == 1:
                oracle.orc_fft(vp(want), n)
            else:
                oracle.orc_coset_fft(vp(want), n, shift)
            assert (got[k] == want).all()
        assert (gpu_ctx.ntt_batch(got, inverse=True, shift=shift) == xc).all()


def test_ntt_large_roundtrip_and_linearity(gpu_ctx):
    # full-size property test (n = 2^22, the BASELINE degree): inverse(forward(x)) = x, and NTT(a+b) = NTT(a)+NTT(b)
    rng = np.random.default_rng(22)
    n = 1 << 22
    a = rand_field(rng, (1, n))
    b = rand_field(rng, (1, n))
    fa = gpu_ctx.ntt_batch(a, shift=7)
    assert (gpu_ctx.ntt_batch(fa, inverse=True, shift=7) == a).all()
    fb = gpu_ctx.ntt_batch(b, shift=7)
    s = ((a.astype(object) + b.astype(object)) % P).astype(np.uint64)
    fs = gpu_ctx.ntt_batch(s, shift=7)
    assert (fs == ((fa.astype(object) + fb.astype(object)) % P).astype(np.uint64)).all()
    # spot-check the definition on a sparse polynomial: c_5 = 1  ->  v_i = (7 w^i)^5
    sp = np.zeros((1, n), dtype=np.uint64)
    sp[0, 5] = 1
    v = gpu_ctx.ntt_batch(sp, shift=7)
    w = pow(pow(7, (P - 1) >> 32, P), 1 << (32 - 22), P)
    for i in (0, 1, 12345, n - 1):
        assert int(v[0, i]) == pow(7 * pow(w, i, P), 5, P)


@pytest.mark.parametrize("lg", [23, 24, 25])
def test_ntt_three_pass_plans_parity(gpu_ctx, oracle, lg):
    """Direct evidence for the multi-pass plans the headline proof runs: lg 23-25 are cut 5/5/13 .. 6/6/13 (two strided passes +
    the contiguous one; 2^25 is the quotient's coset iNTT at n = 2^22).  One column against the oracle's textbook radix-2
    transform in both directions, with and without the coset shift, plus the definition on a sparse polynomial."""
    rng = np.random.default_rng(lg)
    n = 1 << lg
    x = rand_field(rng, (1, n), canonical=False)
    xc = x % np.uint64(P)
    want = xc.copy()
    oracle.orc_fft(vp(want[0]), n)
    got = gpu_ctx.ntt_batch(x)
    assert (got == want).all()
    assert (gpu_ctx.ntt_batch(want, inverse=True) == xc).all()
    del got
    # coset inverse (compute_quotient_polys: values on 7<w> -> coefficients), checked against the oracle's coset_ifft
    back = xc.copy()
    oracle.orc_coset_ifft(vp(back[0]), n, 7)
    assert (gpu_ctx.ntt_batch(x, inverse=True, shift=7) == back).all()
    assert (gpu_ctx.ntt_batch(back, shift=7) == xc).all()
    sp = np.zeros((1, n), dtype=np.uint64)
    sp[0, 3] = 1
    v = gpu_ctx.ntt_batch(sp, shift=7)
    w = pow(pow(7, (P - 1) >> 32, P), 1 << (32 - lg), P)
    for i in (0, 1, 54321, n // 2 + 1, n - 1):
        assert int(v[0, i]) == pow(7 * pow(w, i, P), 3, P)


def test_ntt_batch_headline_size_takes_the_prefetching_kernel(gpu_ctx, oracle):
    """2^22 is the size whose passes have the shapes of the fused prefetching kernel (9 strided + 13 contiguous bits), and 8 columns
    are the fewest for which a plain transform takes it: lcp2_ntt_batch forward without a shift (no factor), with the coset shift
    (the computed scale with a single coset: z = 0) and back, two columns against the oracle's textbook transforms"""
    lg, n, ncols = 22, 1 << 22, 8
    rng = np.random.default_rng(2222)
    x = rand_field(rng, (ncols, n), canonical=False)
    xc = x % np.uint64(P)
    got = gpu_ctx.ntt_batch(x)
    for c in (0, 7):
        want = xc[c].copy()
        oracle.orc_fft(vp(want), n)
        assert (got[c] == want).all()
    assert (gpu_ctx.ntt_batch(got, inverse=True) == xc).all()
    got = gpu_ctx.ntt_batch(x, shift=7)
    for c in (1, 6):
        want = xc[c].copy()
        oracle.orc_coset_fft(vp(want), n, 7)
        assert (got[c] == want).all()
    assert (gpu_ctx.ntt_batch(got, inverse=True, shift=7) == xc).all()


def test_sha256_tree_golden_roots(gpu_ctx, oracle):
    kat = json.load(open(os.path.join(G, "sha256_kat.json")))
    for n, root in kat["zero_leaf_merkle_roots"].items():
        h = int(n).bit_length() - 1
        nodes = gpu_ctx.sha256_tree(np.zeros((int(n), 32), dtype=np.uint8), h)
        assert bytes(nodes[0, -1]).hex() == root
    # SyncCommitteeSSZ (config 2): 1024-leaf tree + the two 2-leaf trees, root pinned by the reference's test
    sc = kat["sync_committee"]
    pks = np.frombuffer(bytes.fromhex("".join(sc["pubkeys"])), dtype=np.uint8).copy()
    leaves = np.zeros((1024, 32), dtype=np.uint8)
    oracle.orc_ssz_sync_committee_leaves(vp(pks), vp(leaves))
    nodes, trace = gpu_ctx.sha256_tree(leaves, 10, trace=True)
    pk_root = nodes[0, -1]
    agg = np.zeros(64, dtype=np.uint8)
    agg[:48] = np.frombuffer(bytes.fromhex(sc["aggregate_pubkey"]), dtype=np.uint8)
    agg_root = gpu_ctx.sha256_tree(agg.reshape(2, 32), 1)[0, -1]
    top = gpu_ctx.sha256_tree(np.stack([pk_root, agg_root]), 1)[0, -1]
    assert bytes(top).hex() == sc["ssz_root"]
    # every node equals the oracle's, every round-trace word equals the oracle's compression trace
    want_nodes = np.zeros((2047, 32), dtype=np.uint8)
    root = np.zeros(32, dtype=np.uint8)
    oracle.orc_sha256_merkle_root(vp(leaves), 10, vp(root), vp(want_nodes))
    assert (nodes[0] == want_nodes).all()
    H0 = np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], dtype=np.uint32)
    for hid in (0, 1, 511, 512, 1000, 1022):
        child = 2 * hid  # node index of the left child in level-major order
        blk = np.frombuffer(bytes(want_nodes[child]) + bytes(want_nodes[child + 1]), dtype=">u4").astype(np.uint32)
        st = H0.copy()
        tr = np.zeros(176, dtype=np.uint32)
        oracle.orc_sha256_compress(vp(st), vp(blk), vp(tr))
        assert (trace[0, hid, 0] == tr).all()
        pad = np.zeros(16, dtype=np.uint32)
        pad[0], pad[15] = 0x80000000, 512
        oracle.orc_sha256_compress(vp(st), vp(pad), vp(tr))
        assert (trace[0, hid, 1] == tr).all()


def test_sha256_many_trees(gpu_ctx):
    rng = np.random.default_rng(7)
    trees, h = 5, 3
    leaves = rng.integers(0, 256, size=(trees, 1 << h, 32), dtype=np.uint8)
    nodes = gpu_ctx.sha256_tree(leaves, h, trees=trees)
    for t in range(trees):
        lvl = [bytes(x) for x in leaves[t]]
        while len(lvl) > 1:
            lvl = [hashlib.sha256(lvl[2 * i] + lvl[2 * i + 1]).digest() for i in range(len(lvl) // 2)]
        assert bytes(nodes[t, -1]) == lvl[0]


@pytest.mark.parametrize("lg,ncols", [(1, 3), (3, 4), (5, 135), (9, 20), (12, 16), (13, 7), (14, 3)])
def test_commit_values_parity(gpu_ctx, oracle, lg, ncols):
    rng = np.random.default_rng(lg * 1000 + ncols)
    vals = rand_field(rng, (ncols, 1 << lg), canonical=False)
    cap_h = min(4, lg + 3)
    o = gpu_ctx.commit_values(vals, 3, cap_h)
    coeffs, lde, cap = commit_reference(oracle, vals, 3, cap_h)
    assert (o.cap == cap).all()
    gc, gl = o.read()
    assert (gc == coeffs).all() and (gl == lde).all()
    idx = np.array([0, 1, (8 << lg) - 1, (8 << lg) // 3], dtype=np.uint64)
    leaves, sib = o.open(idx)
    for q, i in enumerate(idx):
        assert (leaves[q] == lde[:, int(i)]).all()
        assert merkle_verify(oracle, leaves[q], i, sib[q], cap)
        bad = leaves[q].copy()
        bad[0] ^= np.uint64(1)
        assert not merkle_verify(oracle, bad, i, sib[q], cap)
    o2 = gpu_ctx.commit_coeffs(coeffs, 3, cap_h)
    assert (o2.cap == cap).all()
    o.close()
    o2.close()


def test_commit_small_values_trace(gpu_ctx, oracle):
    # SHA-256 traces are byte- and u32-valued: mixed small columns catch canonicalisation shortcuts
    rng = np.random.default_rng(99)
    lg, ncols = 10, 24
    vals = np.zeros((ncols, 1 << lg), dtype=np.uint64)
    vals[0::2] = rng.integers(0, 256, size=(ncols // 2, 1 << lg), dtype=np.uint64)
    vals[1::2] = rng.integers(0, 2 ** 32, size=(ncols // 2, 1 << lg), dtype=np.uint64)
    o = gpu_ctx.commit_values(vals)
    assert (o.cap == commit_reference(oracle, vals)[2]).all()


def test_commit_large_properties(gpu_ctx, oracle):
    # n = 2^20, 16 columns: Merkle openings verify against the cap with the oracle's verifier and the LDE rows
    # equal the committed polynomials evaluated at 7 * w^bitrev(i)
    rng = np.random.default_rng(20)
    lg, ncols = 20, 16
    n = 1 << lg
    vals = rand_field(rng, (ncols, n))
    o = gpu_ctx.commit_values(vals)
    coeffs, _ = o.read(lde=False)
    # coefficients interpolate the values: evaluate two columns at w^k by Horner for a few k
    w = pow(pow(7, (P - 1) >> 32, P), 1 << (32 - lg), P)
    for c in (0, ncols - 1):
        cc = [int(x) for x in coeffs[c]]
        for k in (0, 1, 77777):
            x, acc = pow(w, k, P), 0
            for a in reversed(cc):
                acc = (acc * x + a) % P
            assert acc == int(vals[c, k])
    idx = rng.integers(0, 8 * n, size=6, dtype=np.uint64)
    leaves, sib = o.open(idx)
    W = pow(pow(7, (P - 1) >> 32, P), 1 << (32 - lg - 3), P)
    perm_bits = lg + 3
    for q, i in enumerate(idx):
        assert merkle_verify(oracle, leaves[q], i, sib[q], o.cap)
        r = int(format(int(i), "0%db" % perm_bits)[::-1], 2)
        x, acc = 7 * pow(W, r, P) % P, 0
        for a in reversed([int(v) for v in coeffs[3]]):
            acc = (acc * x + a) % P
        assert acc == int(leaves[q, 3])
    o.close()


@pytest.mark.parametrize("count", [1, 7, 64, 65, 256])
def test_poseidon_single_wave_batches(gpu_ctx, oracle, count):
    # one wave per SIMD is the worst case for VALU hazards in the hand-scheduled carry chains: nothing else issues in between
    rng = np.random.default_rng(1000 + count)
    for rep in range(4):
        s = rand_field(rng, (count, 12), canonical=(rep % 2 == 0))
        if rep == 3:
            s[:] = np.uint64(P - 1)  # all carries / borrows taken
        got = gpu_ctx.poseidon_permute_batch(s)
        want = np.zeros_like(s)
        oracle.orc_poseidon_permute_batch(vp(s), vp(want), count)
        assert (got == want).all()


@pytest.mark.gpu
def test_context_ordered_with_the_default_stream():
    """LCP2_CTX_ORDER_WITH_DEFAULT_STREAM: work queued on stream 0 (torch's default stream) right before a library call, with no
    synchronisation in between, is seen by the library's kernels; and torch reads the result right after, again without one"""
    import torch
    import eth_lc_plonky2_amd as m
    ctx = m.Context(0, order_with_default_stream=True)
    lg, ncols = 18, 8
    n = 1 << lg
    rng = np.random.default_rng(3)
    host = rng.integers(0, m.GOLDILOCKS_P, size=(ncols, n), dtype=np.uint64)
    for trial in range(5):
        buf = torch.zeros((ncols, n), dtype=torch.int64, device="cuda")
        src = torch.from_numpy(host.view(np.int64)).cuda()
        torch.cuda.synchronize()
        buf.copy_(src)                         # on stream 0, not synchronised
        ctx._check(ctx.lib.lcp2_ntt_batch(ctx.handle, ctypes.c_void_p(buf.data_ptr()), ncols, lg, 0, 1, m.MEM_DEVICE))   # forward
        ctx._check(ctx.lib.lcp2_ntt_batch(ctx.handle, ctypes.c_void_p(buf.data_ptr()), ncols, lg, 1, 1, m.MEM_DEVICE))   # and back
        back = buf.clone()                     # on stream 0 again, no ctx.sync()
        torch.cuda.synchronize()
        assert (back.cpu().numpy().view(np.uint64) == host).all()
    ctx.close()
    with pytest.raises(m.Lcp2Error):
        m.Context(0, stream=torch.cuda.Stream().cuda_stream, order_with_default_stream=True)  # the flag is for the private stream only
