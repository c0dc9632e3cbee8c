"""One proof sharded by LDE coset over 2 / 4 / 8 ranks (BASELINE configs[3]; include/lcp2.h "one proof sharded
over the GPUs of a node").  On the single test GPU the ranks are separate sharded circuit handles stepped in lockstep in
one process, the collectives are emulated on the host (uint64 sums of disjoint shares); the assembled proof must equal the single-GPU proof word for word
(which itself equals the oracle's, test_gpu_prover.py).  The torch.distributed collectives are covered on CPU with gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class LockstepComm:
    """collectives of `world` ranks that live in one process: requests are collected from every rank, then answered"""

    def __init__(self, ctx):
        self.ctx = ctx

    def sum_host_all(self, arrays):
        acc = np.zeros_like(arrays[0])
        for a in arrays:
            acc += a  # uint64 wrap-around, as the int64 SUM all-reduce does
        return acc

    def all_gather_device_all(self, reqs):
        """reqs[r] = (pointer, total words, words per rank) of rank r: every buffer ends up with every rank's part"""
        parts = [self.ctx.buffer_read(ptr + 8 * r * wpr, wpr) for r, (ptr, total, wpr) in enumerate(reqs)]
        for ptr, total, wpr in reqs:
            assert total == wpr * len(reqs)
            for r, part in enumerate(parts):
                self.ctx.buffer_write(ptr + 8 * r * wpr, part)

    def all_to_all_device_all(self, reqs):
        """reqs[r] = (send pointer, receive pointer, words per pair) of rank r: part d of rank s's send buffer becomes part s of
        rank d's receive buffer"""
        world = len(reqs)
        sends = [self.ctx.buffer_read(send, world * wpp).reshape(world, wpp) for send, recv, wpp in reqs]
        for d, (send, recv, wpp) in enumerate(reqs):
            self.ctx.buffer_write(recv, np.ascontiguousarray(np.stack([sends[src][d] for src in range(world)])).ravel())


def _drive(comm, gens, world):
    """steps the ranks' generators in lockstep, answering their exchange requests on the host"""
    replies = [None] * world
    while True:
        reqs = []
        for g, rep in zip(gens, replies):
            try:
                reqs.append(g.send(rep))
            except StopIteration:
                reqs.append(None)
        if all(q is None for q in reqs):
            return
        assert all(q is not None and q[0] == reqs[0][0] for q in reqs), "ranks diverged"
        if reqs[0][0] == "sum_host":
            merged = comm.sum_host_all([q[1] for q in reqs])
            replies = [merged.copy() for _ in range(world)]
        elif reqs[0][0] == "all_to_all_device":
            comm.all_to_all_device_all([(q[1], q[2], q[3]) for q in reqs])
            replies = [None] * world
        elif reqs[0][0] == "wait":  # the lockstep gathers are done on the spot
            replies = [None] * world
        else:
            comm.all_gather_device_all([(q[1], q[2], q[3]) for q in reqs])
            replies = [None] * world


def _run_lockstep(m, ctx, circ, wires, pis, world, sharded_columns=False, row_exchange=False, chunked=False):
    comm = LockstepComm(ctx)
    ranks = [m.parallel.ShardedProver(ctx, circ, r, world, None) for r in range(world)]
    cap = comm.sum_host_all([r.cap_share for r in ranks])
    for r in ranks:
        r.comm = type("C", (), {"sum_host": staticmethod(lambda a, cap=cap: cap)})()
        r.finish_build()
    if chunked:  # every rank brings the columns the chunked exchange assigns to it
        gens = [r.prove_steps(np.ascontiguousarray(wires[m.parallel.chunk_columns(circ.params.num_wires, r.rank, world)]), pis, sharded_columns=True,
                              row_exchange=True, chunked=True) for r in ranks]
    elif sharded_columns:  # every rank brings only its column shard of the witness
        gens = [r.prove_steps(np.ascontiguousarray(wires[slice(*r.column_shard())]), pis, sharded_columns=True, row_exchange=row_exchange)
                for r in ranks]
    else:
        gens = [r.prove_steps(wires, pis) for r in ranks]
    _drive(comm, gens, world)
    return ranks


@pytest.mark.gpu
@pytest.mark.parametrize("world,degree_bits,sharded_columns", [(2, 9, False), (4, 8, True), (8, 10, True), (8, 5, False), (2, 6, True), (8, 6, True)])
def test_sharded_proof_equals_single_gpu(gpu_ctx, world, degree_bits, sharded_columns):
    """sharded_columns: the witness arrives column-sharded, is all-gathered, transformed polynomial-parallel and the
    coefficients all-gathered (135 columns over 8 ranks: shards of 17 and one of 16; over 2 and 4: 68/67 and 34/34/34/33)"""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(degree_bits, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=900 + world)
    single = m.CircuitData.build(gpu_ctx, circ)
    want = single.prove(wires, pis)
    ranks = _run_lockstep(m, gpu_ctx, circ, wires, pis, world, sharded_columns)
    for r in ranks:
        assert (r.digest == single.digest()[0]).all()
        bad = np.nonzero(r.proof != want)[0]
        assert bad.size == 0, f"rank {r.rank}/{world}: {bad.size} proof words differ, first at {bad[0]}"
    single.verify(ranks[0].proof, pis)
    # the shares really are shares: a rank answers only the queries whose leaves it holds
    with pytest.raises(m.Lcp2Error):
        ranks[0].data.prove(wires, pis)  # a sharded handle refuses the monolithic call
    for r in ranks:
        r.close()
    single.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,degree_bits,num_wires", [(8, 10, 135), (4, 8, 135), (2, 7, 135), (1, 6, 135), (8, 4, 135), (8, 6, 140), (4, 6, 144), (8, 5, 145)])
def test_chunked_exchange_proof_equals_single_gpu(gpu_ctx, world, degree_bits, num_wires):
    """the row exchange form with the coefficient exchange in chunks of 8 columns (lcp2_commit_wires_rows_begin / _chunk / _finish: coset
    LDE of a chunk and absorption into a persistent sponge state per leaf while the next chunks are gathered): the assembled proof is the
    single-GPU proof word for word.  Chunks are 16 columns (parallel.CHUNK_COLS).  135 wires: the last chunk has 7 columns (at 8 ranks: ranks 0-2
    bring 18 columns, rank 3 17, ranks 4-7 16); 140: 12 in the last chunk; 144: none short; 145: a last chunk of one column"""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(degree_bits, 4)
    params.num_wires = num_wires
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=1500 + world + degree_bits)
    single = m.CircuitData.build(gpu_ctx, circ)
    want = single.prove(wires, pis)
    ranks = _run_lockstep(m, gpu_ctx, circ, wires, pis, world, True, True, chunked=True)
    for r in ranks:
        bad = np.nonzero(r.proof != want)[0]
        assert bad.size == 0, f"rank {r.rank}/{world}: {bad.size} proof words differ, first at {bad[0]}"
    # a second chunked proof on the same handles (the persistent states and the exchange buffers are reused), then the whole-column form
    _drive(LockstepComm(gpu_ctx), [r.prove_steps(np.ascontiguousarray(wires[m.parallel.chunk_columns(num_wires, r.rank, world)]), pis, sharded_columns=True,
                                                 row_exchange=True, chunked=True) for r in ranks], world)
    assert all((r.proof == want).all() for r in ranks)
    assert (_rerun(m, gpu_ctx, ranks, wires, pis, world) == want).all()
    for r in ranks:
        r.close()
    single.close()


@pytest.mark.gpu
def test_chunk_entry_points_enforce_their_order(gpu_ctx):
    """lcp2_commit_wires_chunk: chunks in column order, starting at multiples of 8, only after _begin; _finish only after the last column"""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(6, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=5)
    n = 64
    d = m.CircuitData.build_sharded(gpu_ctx, circ, 0, 8)
    rows = gpu_ctx.buffer_alloc(135 * n)
    coeffs = gpu_ctx.buffer_alloc(136 * n)
    gpu_ctx.buffer_write(rows, wires)
    with pytest.raises(m.Lcp2Error):
        d.commit_wires_chunk(coeffs, 0, 8)       # no _begin
    d.commit_wires_rows_begin(rows)
    with pytest.raises(m.Lcp2Error):
        d.commit_wires_chunk(coeffs, 8, 8)       # out of order
    with pytest.raises(m.Lcp2Error):
        d.commit_wires_chunk(coeffs, 0, 7)       # a short chunk that is not the last
    d.commit_wires_chunk(coeffs, 0, 16)
    with pytest.raises(m.Lcp2Error):
        d.commit_wires_rows_finish()             # columns missing
    with pytest.raises(m.Lcp2Error):
        d.commit_wires_chunk(coeffs, 16, 128)    # past the last wire
    d.commit_wires_chunk(coeffs + 8 * 16 * n, 16, 119)
    d.commit_wires_rows_finish()
    gpu_ctx.buffer_free(rows)
    gpu_ctx.buffer_free(coeffs)
    d.close()


@pytest.mark.gpu
def test_sharded_columns_with_a_short_middle_shard(gpu_ctx):
    """130 wires over 8 ranks: shards 17, 17, 16 x 6: the padded all-gather layout has gaps that are closed before the commitment"""
    import eth_lc_plonky2_amd as m
    from eth_lc_plonky2_amd import poseidon_py  # noqa: F401
    params = m.standard_params(7, 4)
    params.num_wires = 140
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=31)
    assert [e - s for s, e in m.parallel.column_shards(140, 8)] == [18, 18, 18, 18, 17, 17, 17, 17]
    single = m.CircuitData.build(gpu_ctx, circ)
    want = single.prove(wires, pis)
    ranks = _run_lockstep(m, gpu_ctx, circ, wires, pis, 8, True)
    assert (ranks[3].proof == want).all()
    for r in ranks:
        r.close()
    single.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,degree_bits,num_wires", [(8, 10, 135), (4, 8, 135), (2, 6, 135), (8, 5, 135), (8, 4, 135), (8, 7, 140), (1, 9, 135)])
def test_row_exchange_proof_equals_single_gpu(gpu_ctx, world, degree_bits, num_wires):
    """the row exchange form (include/lcp2.h): the witness values cross the ranks as row blocks, K5 and the gate check run on a
    rank's own rows, the Z / partial-product rows are all-gathered.  (8, 4): two rows per rank; (8, 7, 140): gaps in the padded
    layouts of the all-gather and of the all-to-all"""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(degree_bits, 4)
    params.num_wires = num_wires
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=1200 + world + degree_bits)
    single = m.CircuitData.build(gpu_ctx, circ)
    want = single.prove(wires, pis)
    ranks = _run_lockstep(m, gpu_ctx, circ, wires, pis, world, True, True)
    for r in ranks:
        bad = np.nonzero(r.proof != want)[0]
        assert bad.size == 0, f"rank {r.rank}/{world}: {bad.size} proof words differ, first at {bad[0]}"
    # the same handles take the whole-column form again afterwards (the row mode is per proof)
    again = _rerun(m, gpu_ctx, ranks, wires, pis, world)
    assert (again == want).all()
    for r in ranks:
        r.close()
    single.close()


def _rerun(m, ctx, ranks, wires, pis, world):
    comm = LockstepComm(ctx)
    gens = [r.prove_steps(np.ascontiguousarray(wires[slice(*r.column_shard())]), pis, sharded_columns=True) for r in ranks]
    replies = [None] * world
    while True:
        reqs = []
        for g, rep in zip(gens, replies):
            try:
                reqs.append(g.send(rep))
            except StopIteration:
                reqs.append(None)
        if all(q is None for q in reqs):
            return ranks[-1].proof
        if reqs[0][0] == "sum_host":
            merged = comm.sum_host_all([q[1] for q in reqs])
            replies = [merged.copy() for _ in range(world)]
        else:
            comm.all_gather_device_all([(q[1], q[2], q[3]) for q in reqs])
            replies = [None] * world


@pytest.mark.gpu
def test_row_exchange_entry_points_refuse_misuse(gpu_ctx):
    """lcp2_commit_wires_rows needs a sharded circuit; the rows_* calls need it to have run; the whole-column lcp2_perm_zs refuses
    a handle that holds row blocks only; nothing is left half-done: the handle proves normally afterwards"""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(6, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=3)
    n = 1 << 6
    single = m.CircuitData.build(gpu_ctx, circ)
    want = single.prove(wires, pis)
    buf = gpu_ctx.buffer_alloc(params.num_wires * n)
    gpu_ctx.buffer_write(buf, wires)
    with pytest.raises(m.Lcp2Error):
        single.commit_wires_rows(buf, buf)            # not a sharded circuit
    with pytest.raises(m.Lcp2Error):
        single.perm_zs_rows_begin(np.ones(2, np.uint64), np.ones(2, np.uint64), 1)
    assert (single.prove(wires, pis) == want).all()
    rank0 = m.parallel.ShardedProver(gpu_ctx, circ, 0, 1, None)  # one rank holding all 8 blocks
    rank0.comm = type("C", (), {"sum_host": staticmethod(lambda a: a)})()
    rank0.finish_build()
    d = rank0.data
    with pytest.raises(m.Lcp2Error):
        d.perm_zs_rows_begin(np.ones(2, np.uint64), np.ones(2, np.uint64), 1)   # no wires committed
    coeffs = gpu_ctx.buffer_alloc(params.num_wires * n)
    gpu_ctx.buffer_copy(coeffs, buf, params.num_wires * n)
    gpu_ctx._check(gpu_ctx.lib.lcp2_ntt_batch(gpu_ctx.handle, __import__("ctypes").c_void_p(coeffs), params.num_wires, 6, 1, 1, m.MEM_DEVICE))
    d.commit_wires_rows(buf, coeffs)
    with pytest.raises(m.Lcp2Error):
        d.perm_zs(np.ones(2, np.uint64), np.ones(2, np.uint64))                  # the handle holds row blocks: rows_* only
    with pytest.raises(m.Lcp2Error):
        d.perm_zs_commit()                                                       # nothing to commit yet (no begin / finish)
    with pytest.raises(m.Lcp2Error):
        d.perm_zs_rows_finish(np.ones(2, np.uint64))                             # no begin
    betas, gammas = np.array([3, 5], np.uint64), np.array([7, 11], np.uint64)
    products = d.perm_zs_rows_begin(betas, gammas, 1)
    with pytest.raises(m.Lcp2Error):
        d.perm_zs_commit()                                                       # begin, but no finish
    ptr, words = d.perm_zs_rows_finish(products)
    assert words == 20 * n and ptr
    with pytest.raises(m.Lcp2Error):
        d.perm_zs_rows_finish(products)                                          # twice
    cap = d.perm_zs_commit()
    assert (cap.reshape(-1, 4) != 0).any(axis=1).all()                           # one rank: the whole cap
    with pytest.raises(m.Lcp2Error):
        d.perm_zs_commit()                                                       # twice
    gpu_ctx.buffer_free(buf)
    gpu_ctx.buffer_free(coeffs)
    rank0.close()
    single.close()


@pytest.mark.gpu
def test_row_exchange_takes_a_non_canonical_witness(gpu_ctx):
    """values in [p, 2^64) in the column shards: the rank's own iNTT canonicalises the coefficients, the row blocks are scanned by
    lcp2_commit_wires_rows and K5 / the gate check continue from a canonical copy of the block: the same proof"""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(7, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=23, small_values=True)
    single = m.CircuitData.build(gpu_ctx, circ)
    want = single.prove(wires, pis)
    rng = np.random.default_rng(6)
    P = np.uint64(m.GOLDILOCKS_P)
    lifted = wires.copy()
    idx = np.argwhere(lifted < np.uint64(2 ** 32 - 1))  # x + p < 2^64
    for r, c in idx[rng.choice(len(idx), size=400, replace=False)]:
        lifted[r, c] += P
    assert (lifted >= P).sum() == 400
    ranks = _run_lockstep(m, gpu_ctx, circ, lifted, pis, 4, True, True)
    for r in ranks:
        assert (r.proof == want).all()
        r.close()
    single.close()


@pytest.mark.gpu
def test_row_exchange_reports_an_unsatisfied_witness_on_every_rank(gpu_ctx):
    """a violated gate constraint is found by the rank that holds the row and reaches the others through the verdict exchange;
    a broken copy constraint shows in the product of the block products, which every rank computes"""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(8, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=77)
    single = m.CircuitData.build(gpu_ctx, circ)
    for what in ("gate", "copy"):
        bad = wires.copy()
        row = 5 * 32 + 5  # a BaseSumGate row in rank 5's block; its limbs are in no copy class
        if what == "gate":
            bad[1, row] ^= np.uint64(1)  # still a bit, but the limbs no longer sum to wire 0
            with pytest.raises(m.Lcp2Error, match=f"gate constraint on row {row}"):
                single.prove(bad, pis)
        else:
            first = int(np.nonzero(circ.constants_sigmas[0] == circ.gateset.index("ArithmeticGate"))[0][0])
            bad[1, first] = np.uint64(12345)  # the first arithmetic row's copy of public input 0 (the permutation argument runs first)
            with pytest.raises(m.Lcp2Error, match="copy constraint"):
                single.prove(bad, pis)
        comm = LockstepComm(gpu_ctx)
        ranks = [m.parallel.ShardedProver(gpu_ctx, circ, r, 8, None) for r in range(8)]
        cap = comm.sum_host_all([r.cap_share for r in ranks])
        for r in ranks:
            r.comm = type("C", (), {"sum_host": staticmethod(lambda a, cap=cap: cap)})()
            r.finish_build()
        gens = [r.prove_steps(np.ascontiguousarray(bad[slice(*r.column_shard())]), pis, sharded_columns=True, row_exchange=True) for r in ranks]
        replies, raised = [None] * 8, {}
        for _ in range(40):
            reqs = []
            for k, (g, rep) in enumerate(zip(gens, replies)):
                if k in raised:
                    reqs.append(None)
                    continue
                try:
                    reqs.append(g.send(rep))
                except m.Lcp2Error as e:
                    raised[k] = e
                    reqs.append(None)
            live = [q for q in reqs if q is not None]
            if not live:
                break
            assert len(live) == 8, f"{what}: only some ranks stopped: {sorted(raised)}"   # all or none at every exchange point
            if live[0][0] == "sum_host":
                merged = comm.sum_host_all([q[1] for q in reqs])
                replies = [merged.copy() for _ in range(8)]
            elif live[0][0] == "all_to_all_device":
                comm.all_to_all_device_all([(q[1], q[2], q[3]) for q in reqs]); replies = [None] * 8
            else:
                comm.all_gather_device_all([(q[1], q[2], q[3]) for q in reqs]); replies = [None] * 8
        assert sorted(raised) == list(range(8)) and all(e.status == m.binding.E_UNSAT for e in raised.values())
        if what == "gate":
            assert f"row {row}" in str(raised[5]) and all("another rank" in str(raised[k]) for k in range(8) if k != 5)
        else:
            assert all("copy constraint" in str(e) for e in raised.values())
        for r in ranks:
            r.close()
    single.close()


@pytest.mark.gpu
def test_buffer_copy_2d(gpu_ctx):
    """lcp2_buffer_copy_2d: row blocks out of whole columns and back, incl. runs of 2^22 words (the one-rank row exchange at the
    headline size copies whole columns with it)"""
    import eth_lc_plonky2_amd as m
    rng = np.random.default_rng(5)
    for cols, n, world in ((7, 64, 4), (3, 1 << 22, 1), (5, 1 << 16, 8)):
        rows = n // world
        a = rng.integers(0, 2 ** 63, size=(cols, n), dtype=np.uint64)
        src, dst = gpu_ctx.buffer_alloc(cols * n), gpu_ctx.buffer_alloc(cols * n)
        gpu_ctx.buffer_write(src, a)
        for d in range(world):  # [column][n] -> [rank][column][rows]
            gpu_ctx.buffer_copy_2d(dst + 8 * d * cols * rows, rows, src + 8 * d * rows, n, rows, cols)
        got = gpu_ctx.buffer_read(dst, cols * n).reshape(world, cols, rows)
        assert all((got[d] == a[:, d * rows:(d + 1) * rows]).all() for d in range(world))
        gpu_ctx.buffer_free(src)
        gpu_ctx.buffer_free(dst)
    with pytest.raises(m.Lcp2Error):
        p = gpu_ctx.buffer_alloc(64)
        try:
            gpu_ctx.buffer_copy_2d(p, 4, p + 256, 8, 8, 2)  # runs wider than the destination pitch
        finally:
            gpu_ctx.buffer_free(p)


@pytest.mark.gpu
def test_device_pointer_view_aliases_library_memory(gpu_ctx):
    """TorchComm.all_gather_device hands RCCL a tensor that IS the library's buffer (CUDA array interface, no staging copy)"""
    import torch
    import eth_lc_plonky2_amd as m
    p = gpu_ctx.buffer_alloc(1024)
    gpu_ctx.buffer_write(p, np.arange(1024, dtype=np.uint64))
    t = torch.as_tensor(m.parallel._DevicePtr(p, 1024), device=torch.device("cuda", 0))
    assert t.dtype == torch.int64 and t.data_ptr() == p and int(t[1000].item()) == 1000
    t[7] = -1
    torch.cuda.synchronize()
    assert int(gpu_ctx.buffer_read(p, 8)[7]) == 2 ** 64 - 1
    gpu_ctx.buffer_free(p)


def test_host_transcript_helpers_match_oracle(oracle):
    """lcp2_challenger_* / lcp2_hash_no_pad (host C++) against the oracle's Poseidon"""
    import eth_lc_plonky2_amd as m
    import oracle_lib
    rng = np.random.default_rng(5)
    P = m.GOLDILOCKS_P
    vals = rng.integers(0, P, size=41, dtype=np.uint64)
    want = np.zeros(4, dtype=np.uint64)
    oracle.orc_hash_no_pad(oracle_lib.vp(vals), len(vals), oracle_lib.vp(want))
    assert (m.binding.hash_no_pad(vals) == want).all()
    # duplex challenger: observe 13, draw 3, observe 2, draw 9 (crosses a squeeze boundary)
    ch = m.binding.Challenger()
    s, inp, out, got = np.zeros(12, dtype=np.uint64), [], [], []

    def duplex():
        nonlocal inp, out
        for i, v in enumerate(inp):
            s[i] = v
        inp = []
        oracle.orc_poseidon_permute(oracle_lib.vp(s))
        out = [int(v) for v in s[:8]]

    def observe(xs):
        nonlocal inp, out
        for x in xs:
            out = []
            inp.append(int(x))
            if len(inp) == 8:
                duplex()

    def get(k):
        r = []
        for _ in range(k):
            if inp or not out:
                duplex()
            r.append(out.pop())
        return r

    ch.observe(vals[:13]); observe(vals[:13])
    assert list(ch.get(3)) == get(3)
    ch.observe(vals[13:15]); observe(vals[13:15])
    assert list(ch.get(9)) == get(9)


_GLOO_PRELUDE = r"""
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import eth_lc_plonky2_amd as m
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
comm = m.parallel.TorchComm(dist)
class FakeCtx:  # "device" buffers are numpy arrays, addressed by their data pointer
    def __init__(self): self.bufs = {}
    def buffer_alloc(self, words):
        a = np.zeros(words, dtype=np.uint64); self.bufs[a.ctypes.data] = a; return a.ctypes.data
    def buffer_free(self, ptr): del self.bufs[ptr]
    def _at(self, ptr, words):
        for base, a in self.bufs.items():
            if base <= ptr < base + 8 * a.size: return a[(ptr - base) // 8:(ptr - base) // 8 + words]
        raise KeyError(ptr)
    def buffer_write(self, ptr, arr): self._at(ptr, arr.size)[:] = np.asarray(arr).ravel()
    def buffer_read(self, ptr, words): return self._at(ptr, words).copy()
    def buffer_copy(self, dst, src, words): self._at(dst, words)[:] = self._at(src, words).copy()
    def buffer_copy_2d(self, dst, dst_pitch, src, src_pitch, width, height):
        for h in range(height): self._at(dst + 8 * h * dst_pitch, width)[:] = self._at(src + 8 * h * src_pitch, width).copy()
    def sync(self): pass
"""

_GLOO_WORKER = _GLOO_PRELUDE + r"""
# cap shares: rank r owns entries [8r, 8r + 8) of a 16-entry cap
share = np.zeros((16, 4), dtype=np.uint64)
share[8 * rank:8 * rank + 8] = np.arange(32, dtype=np.uint64).reshape(8, 4) + np.uint64(1000 * (rank + 1)) + (np.uint64(1) << np.uint64(63))
full = comm.sum_host(share)
want = np.zeros((16, 4), dtype=np.uint64)
for r in range(2):
    want[8 * r:8 * r + 8] = np.arange(32, dtype=np.uint64).reshape(8, 4) + np.uint64(1000 * (r + 1)) + (np.uint64(1) << np.uint64(63))
assert (full == want).all() and full.dtype == np.uint64
# replicated words come from rank 0 only (lcp2_fri_open zeroes them on the other ranks): full 64-bit values survive the int64 sum
rep = np.array([5, 2**64 - 1, 0, 2**63 + 77], dtype=np.uint64)
assert (comm.sum_host(rep if rank == 0 else np.zeros_like(rep)) == rep).all()
assert m.parallel.block_range(rank, 2) == (4 * rank, 4)

# ShardedProver.prove() end to end over real gloo collectives, with the per-rank compute replaced by a stand-in that returns
# disjoint shares the way the C ABI does (own entries at their global position, zeros elsewhere; replicated words from rank 0)
class FakeParams:
    cap_height, num_challenges, rate_bits, degree_bits, num_wires = 4, 2, 3, 2, 135
class FakeCirc:
    params = FakeParams()
class FakeData:
    proof_words = 3 * 64 + 40
    def __init__(self, rank): self.rank, self.qbuf = rank, np.zeros(64, dtype=np.uint64)
    def _share(self, tag):
        c = np.zeros((16, 4), dtype=np.uint64)
        c[8 * self.rank:8 * self.rank + 8] = np.arange(32, dtype=np.uint64).reshape(8, 4) * np.uint64(7) + np.uint64(tag + 100 * self.rank)
        return c
    def commit_wires(self, wires, mem): return self._share(1)
    def perm_zs(self, betas, gammas): self.seen = [int(betas[0]), int(gammas[1])]; return self._share(2)
    def quotient_values(self, alphas, pi_hash):  # this rank's blocks of both challenge planes (32 words each), zeros elsewhere
        for c in range(2):
            self.qbuf[32 * c + 16 * self.rank:32 * c + 16 * self.rank + 16] = np.arange(16, dtype=np.uint64) + np.uint64(int(alphas[0]) % 1000 + 50 * c)
    def quotient_buffer(self): return (self.qbuf.ctypes.data, self.qbuf.size)
    def quotient_commit(self): return self._share(3 + int(self.qbuf.sum() % 5))  # depends on the exchanged buffer
    # proof body (after the three caps): [0, 8) openings, [8, 20) FRI cap of layer 0 + replicated words, [20, 40) query answers
    def proof_section(self, which): return {0: (192, 8), 1: (200, 4), 2: (192, 40)}[which]
    def fri_open_begin(self, zeta, state, proof):
        self.z = int(zeta[0]) % 97
        proof[192 + 4 * self.rank:196 + 4 * self.rank] = np.arange(4, dtype=np.uint64) + np.uint64(4 * self.rank + self.z)  # own columns
    def fri_open_commit(self, proof):
        assert (proof[192:200] == np.arange(8, dtype=np.uint64) + np.uint64(self.z)).all()        # the openings arrived complete
        proof[200 + 2 * self.rank:202 + 2 * self.rank] = np.uint64(8 + self.z) + np.arange(2, dtype=np.uint64) + np.uint64(2 * self.rank)
    def fri_open_finish(self, proof):
        assert (proof[200:204] == np.arange(4, dtype=np.uint64) + np.uint64(8 + self.z)).all()    # and so did the layer-0 cap
        body = proof[3 * 64:]
        if self.rank == 0: body[12:20] = np.arange(8, dtype=np.uint64) + np.uint64(12 + self.z)   # replicated part
        else: body[:20] = 0                                                                       # ... from rank 0 only
        body[20 + 10 * self.rank:30 + 10 * self.rank] = np.uint64(555 + self.rank)                # this rank's query answers
class GlooComm(m.parallel.TorchComm):
    def all_gather_device(self, ptr, total_words, words_per_rank):  # the "device" buffer of the stand-in is a numpy array
        import torch
        off = (ptr - sp.data.qbuf.ctypes.data) // 8
        view = torch.from_numpy(sp.data.qbuf.view(np.int64))[off:off + total_words]
        self.all_gather_tensor(view, rank)
sp = object.__new__(m.parallel.ShardedProver)
sp.b, sp.ctx, sp.circ, sp.rank, sp.world, sp.comm = m.binding, None, FakeCirc(), rank, 2, GlooComm(dist)
sp.data = FakeData(rank)
sp.digest = np.arange(4, dtype=np.uint64)
proof = sp.prove(None, np.array([3, 4], dtype=np.uint64))
# both ranks hold the same assembled proof: caps complete, quotient buffer complete, replicated words once, every query answered
gathered = [None, None]
dist.all_gather_object(gathered, proof.tobytes())
assert gathered[0] == gathered[1]
caps = proof[:192].reshape(3, 16, 4)
assert (caps[0] != 0).all() and (caps[1][8:] != 0).all() and (sp.data.qbuf != 0).sum() >= 62
# the two planes of the quotient buffer were completed by in-place all-gathers of the ranks' contiguous runs
a0 = int(sp.data.qbuf[0])
want_q = np.concatenate([np.arange(16, dtype=np.uint64) + np.uint64(a0 + 50 * c) for c in range(2) for r in range(2)])
assert (sp.data.qbuf == want_q).all()
# and the in-place all-gather primitive on its own
import torch
t = torch.zeros(8, dtype=torch.int64)
t[4 * rank:4 * rank + 4] = torch.arange(4) + 10 * (rank + 1)
sp.comm.all_gather_tensor(t, rank)
assert t.tolist() == [10, 11, 12, 13, 20, 21, 22, 23]
body = proof[192:]
assert (body[20:30] == 555).all() and (body[30:40] == 556).all() and body[1] == body[0] + 1

# TorchComm.self_check (what bench.py runs before the sharded proof): a known-answer all-gather on a library buffer; the in-place
# form is kept when it works, the staged form is chosen - by every rank alike - when the in-place form returns wrong data or the
# framework refuses it, and the check raises when neither form works
class CheckComm(m.parallel.TorchComm):
    broken, refuse = False, False
    def all_gather_device(self, ptr, total_words, words_per_rank):
        view = torch.from_numpy(self.ctx._at(ptr, total_words).view(np.int64))
        if not self.staged and self.refuse: raise RuntimeError("in-place all-gather refused")
        self.all_gather_tensor(view, rank)
        if not self.staged and self.broken and rank == 1: view[3] += 1   # the aliased form "returns wrong data" on one rank only
        self.bytes_gathered += 8 * words_per_rank
    a2a_broken = False
    def all_to_all_device(self, send_ptr, recv_ptr, words_per_pair):
        send = torch.from_numpy(self.ctx._at(send_ptr, 2 * words_per_pair).view(np.int64)).clone()
        recv = torch.from_numpy(self.ctx._at(recv_ptr, 2 * words_per_pair).view(np.int64))
        self.dist.all_to_all_single(recv, send)
        if self.a2a_broken and rank == 0: recv[1] ^= 1
        self.bytes_gathered += 8 * words_per_pair
fc = FakeCtx()
cc = CheckComm(dist, None, fc)
assert cc.self_check(256) == "in-place" and cc.row_exchange_ok and cc.bytes_gathered == 0
cc = CheckComm(dist, None, fc); cc.a2a_broken = True   # wrong data on one rank: both ranks give the row exchange form up
assert cc.self_check(256) == "in-place" and not cc.row_exchange_ok
assert CheckComm(dist, None, fc, staged=True).self_check(256) == "staged"
for attr in ("broken", "refuse"):
    cc = CheckComm(dist, None, fc); setattr(cc, attr, True)
    assert cc.self_check(256) == "staged" and cc.staged and cc.bytes_gathered == 0
class Hopeless(CheckComm):
    def all_gather_device(self, ptr, total_words, words_per_rank): pass     # gathers nothing in either form
try:
    Hopeless(dist, None, fc).self_check(64)
    raise SystemExit("self_check accepted a collective that gathers nothing")
except RuntimeError:
    pass
assert not fc.bufs  # every check buffer was freed
dist.destroy_process_group()
print("ok", rank)
"""


def test_torch_comm_sum_allreduce_gloo(tmp_path):
    """world_size-2 gloo: the SUM all-reduce (int64 wrap-around) that assembles caps and the proof from the ranks' shares, and
    ShardedProver.prove() driven end to end over those collectives with a stand-in for the per-rank GPU compute"""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    import socket
    with socket.socket() as sk:  # a port the OS says is free (the other gloo tests do the same)
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ok {r}" in o, o


# The row exchange form over real gloo collectives: 5 columns of 8 rows over 2 ranks (column shards 3 + 2, row blocks of 4).  The
# stand-in for the per-rank compute checks what arrives: all coefficients, exactly its own rows of every column, the complete
# table of block products and the complete Z / partial-product buffer; the gate-check verdict is exchanged before the quotient planes.
_GLOO_ROWS_WORKER = _GLOO_PRELUDE + r"""
import ctypes, torch
N, W, CH, NCZ = 8, 5, 2, 4
witness = (np.arange(W * N, dtype=np.uint64).reshape(W, N) + np.uint64(1)) * np.uint64(0x100000001)
fc = FakeCtx()
class FakeLib:
    def lcp2_ntt_batch(self, handle, ptr, ncols, log_n, inverse, coset, mem):  # stand-in "iNTT": x -> 3 x + 1, in place
        v = fc._at(ptr.value, ncols * N); v[:] = v * np.uint64(3) + np.uint64(1); return 0
fc.lib, fc.handle, fc._check = FakeLib(), None, (lambda rc: None)
class FakeParams:
    cap_height, num_challenges, rate_bits, degree_bits, num_wires = 4, CH, 3, 3, W
class FakeCirc:
    params = FakeParams()
class RowsData:
    proof_words = 3 * 64 + 40
    fail_check = False
    def __init__(self, rank): self.rank, self.qbuf, self.zbuf = rank, np.zeros(64, dtype=np.uint64), np.zeros(NCZ * N, dtype=np.uint64)
    def _share(self, tag):
        c = np.zeros((16, 4), dtype=np.uint64)
        c[8 * self.rank:8 * self.rank + 8] = np.arange(32, dtype=np.uint64).reshape(8, 4) * np.uint64(7) + np.uint64(tag + 100 * self.rank)
        return c
    def commit_wires_rows(self, rows_ptr, coeffs_ptr):
        rows = fc.buffer_read(rows_ptr, W * 4).reshape(W, 4)
        assert (rows == witness[:, 4 * self.rank:4 * self.rank + 4]).all(), rows          # this rank's rows of EVERY column
        coeffs = fc.buffer_read(coeffs_ptr, W * N).reshape(W, N)
        assert (coeffs == witness * np.uint64(3) + np.uint64(1)).all()                    # every column's coefficients
        return self._share(1)
    def perm_zs_rows_begin(self, betas, gammas, world):
        out = np.zeros(world * CH, dtype=np.uint64); out[CH * self.rank:CH * self.rank + CH] = [11 + self.rank, 21 + self.rank]; return out
    def perm_zs_rows_finish(self, products):
        assert list(products) == [11, 21, 12, 22]
        z = self.zbuf.reshape(2, NCZ, 4); z[self.rank] = np.arange(NCZ * 4, dtype=np.uint64).reshape(NCZ, 4) + np.uint64(1000 * (self.rank + 1))
        return self.zbuf.ctypes.data, self.zbuf.size
    def perm_zs_commit(self):
        want = np.stack([np.arange(NCZ * 4, dtype=np.uint64).reshape(NCZ, 4) + np.uint64(1000 * (r + 1)) for r in range(2)])
        assert (self.zbuf.reshape(2, NCZ, 4) == want).all()
        return self._share(2)
    def quotient_values(self, alphas, pi_hash):
        if self.fail_check and self.rank == 1: raise m.binding.Lcp2Error(m.binding.E_UNSAT, "row 5")
        for c in range(2): self.qbuf[32 * c + 16 * self.rank:32 * c + 16 * self.rank + 16] = np.uint64(7 + c)
    def quotient_buffer(self): return (self.qbuf.ctypes.data, self.qbuf.size)
    def quotient_commit(self): assert (self.qbuf != 0).all(); return self._share(3)
    def proof_section(self, which): return {0: (192, 8), 1: (200, 4), 2: (192, 40)}[which]
    def fri_open_begin(self, zeta, state, proof): proof[192 + 4 * self.rank:196 + 4 * self.rank] = 5
    def fri_open_commit(self, proof): proof[200 + 2 * self.rank:202 + 2 * self.rank] = 6
    def fri_open_finish(self, proof): proof[212 + 10 * self.rank:222 + 10 * self.rank] = 9
class RowsComm(m.parallel.TorchComm):  # the library buffers of the stand-in are numpy arrays
    def _view(self, ptr, words):
        for a in (sp.data.qbuf, sp.data.zbuf):
            if a.ctypes.data <= ptr < a.ctypes.data + 8 * a.size: return torch.from_numpy(a.view(np.int64))[(ptr - a.ctypes.data) // 8:][:words]
        return torch.from_numpy(fc._at(ptr, words).view(np.int64))
    def all_gather_device(self, ptr, total_words, words_per_rank):
        self.all_gather_tensor(self._view(ptr, total_words), rank); self.bytes_gathered += 8 * words_per_rank
    def all_to_all_device(self, send_ptr, recv_ptr, words_per_pair):
        self.all_to_all_tensor(self._view(recv_ptr, 2 * words_per_pair), self._view(send_ptr, 2 * words_per_pair))
        self.bytes_gathered += 8 * words_per_pair
sp = object.__new__(m.parallel.ShardedProver)
sp.b, sp.ctx, sp.circ, sp.rank, sp.world, sp.comm = m.binding, fc, FakeCirc(), rank, 2, RowsComm(dist)
sp._vals = sp._coeffs = sp._row_bufs = None
sp.data = RowsData(rank)
sp.digest = np.arange(4, dtype=np.uint64)
first, end = sp.column_shard()
assert (first, end) == ((0, 3), (3, 5))[rank]
proof = sp.prove(witness[first:end].copy(), np.array([3, 4], dtype=np.uint64), sharded_columns=True, row_exchange=True)
gathered = [None, None]
dist.all_gather_object(gathered, proof.tobytes())
assert gathered[0] == gathered[1] and (proof[:192].reshape(3, 16, 4) != 0).all()
# received: coefficients of the other rank's slot (3 columns of 8), its row blocks (3 x 4), half of the Z buffer, half of both planes
assert sp.comm.bytes_gathered == 8 * (3 * 8 + 3 * 4 + NCZ * 4 + 32), sp.comm.bytes_gathered
# the all-to-all primitive on its own: whole, in pieces (a message above the per-pair limit), staged
for limit, staged in ((1 << 26, False), (5, False), (1 << 26, True)):
    cm = RowsComm(dist, staged=staged); cm.A2A_WORDS_PER_PAIR = limit
    snd = torch.arange(24, dtype=torch.int64) + 100 * rank
    rcv = torch.zeros(24, dtype=torch.int64)
    cm.all_to_all_tensor(rcv, snd)
    assert rcv.tolist() == [12 * rank + j for j in range(12)] + [100 + 12 * rank + j for j in range(12)], (limit, staged, rcv.tolist())
for limit in (1 << 27, 5):  # and the all-gather, whole and in pieces
    cm = RowsComm(dist); cm.GATHER_WORDS_PER_RANK = limit
    buf = torch.zeros(24, dtype=torch.int64); buf[12 * rank:12 * rank + 12] = torch.arange(12) + 50 * (rank + 1)
    cm.all_gather_tensor(buf, rank)
    assert buf.tolist() == [50 + j for j in range(12)] + [100 + j for j in range(12)], (limit, buf.tolist())
# ---- the chunked coefficient exchange (ShardedProver._commit_wires_chunked): 35 wires = chunks of 16, 16 and 3 columns, 8 per rank and chunk
W2 = 35
witness2 = (np.arange(W2 * N, dtype=np.uint64).reshape(W2, N) + np.uint64(5)) * np.uint64(0x10001)
class FakeParams2(FakeParams):
    num_wires = W2
class FakeCirc2:
    params = FakeParams2()
class ChunkData(RowsData):
    def commit_wires_rows_begin(self, rows_ptr):
        rows = fc.buffer_read(rows_ptr, 48 * 4).reshape(48, 4)[:W2]
        assert (rows == witness2[:, 4 * self.rank:4 * self.rank + 4]).all(), rows     # this rank's rows of EVERY column, in column order
        self.seen = []
    def commit_wires_chunk(self, coeffs_ptr, first_col, ncols):
        got = fc.buffer_read(coeffs_ptr, ncols * N).reshape(ncols, N)
        assert (got == witness2[first_col:first_col + ncols] * np.uint64(3) + np.uint64(1)).all(), (first_col, got)   # complete when it is absorbed
        self.seen.append((first_col, ncols))
    def commit_wires_rows_finish(self):
        assert self.seen == [(0, 16), (16, 16), (32, 3)], self.seen
        return self._share(1)
sp2 = object.__new__(m.parallel.ShardedProver)
sp2.b, sp2.ctx, sp2.circ, sp2.rank, sp2.world, sp2.comm = m.binding, fc, FakeCirc2(), rank, 2, RowsComm(dist)
sp2._vals = sp2._coeffs = sp2._row_bufs = sp2._chunk_bufs = None
sp2.data = ChunkData(rank)
sp2.digest = np.arange(4, dtype=np.uint64)
mine = m.parallel.chunk_columns(W2, rank, 2)
assert mine == (list(range(0, 8)) + list(range(16, 24)) + [32, 33, 34], list(range(8, 16)) + list(range(24, 32)))[rank]
_sp_saved, sp = sp, sp2   # RowsComm._view looks the stand-in's own buffers up through `sp`
proof2 = sp2.prove(witness2[mine].copy(), np.array([3, 4], dtype=np.uint64), sharded_columns=True, row_exchange=True, chunked=True)
gathered = [None, None]
dist.all_gather_object(gathered, proof2.tobytes())
assert gathered[0] == gathered[1] and (proof2[:192].reshape(3, 16, 4) != 0).all()
sp = _sp_saved
# a gate violation found by one rank stops both, after the verdict exchange and before the next collective
sp.data.fail_check = True
try:
    sp.prove(witness[first:end].copy(), np.array([3, 4], dtype=np.uint64), sharded_columns=True, row_exchange=True)
    raise SystemExit("an unsatisfied row on rank 1 went unnoticed on rank %d" % rank)
except m.binding.Lcp2Error as e:
    assert e.status == m.binding.E_UNSAT and (("row 5" in str(e)) == (rank == 1))
dist.barrier()
sp.data = None
m.parallel.ShardedProver.close  # (buffers of the stand-in context are numpy arrays: nothing to free on a device)
dist.destroy_process_group()
print("ok", rank)
"""


def _run_gloo_pair(tmp_path, text):
    script = tmp_path / "worker.py"
    script.write_text(text)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ok {r}" in o, o


def test_row_exchange_over_gloo(tmp_path):
    """world_size-2 gloo: the all-to-all of row blocks, the all-gather of the Z / partial-product rows, the block-product and
    gate-check-verdict all-reduces of ShardedProver.prove(..., row_exchange=True); then the chunked form (chunked=True): 35 wires in
    chunks of 16, 16 and 3 columns, every chunk complete - gathered from both ranks - when it is handed to the commitment, in column order"""
    _run_gloo_pair(tmp_path, _GLOO_ROWS_WORKER)


# The chunked coefficient exchange and the collectives' piece-wise forms over FOUR gloo ranks (the 2-rank workers above fix the world
# size in their stand-ins): chunk_columns / the column-order copies / the row blocks of _commit_wires_chunked with k = 4 columns per
# rank and chunk and a ragged last chunk (37 wires: 16 + 16 + 5 columns, ranks 2 and 3 bring fewer), and all_to_all_tensor /
# all_gather_tensor whole, in pieces and staged with four peers.
_GLOO_WORLD4_WORKER = _GLOO_PRELUDE.replace("world_size=2)", "world_size=int(sys.argv[4]))") + r"""
import torch
WORLD, N, W, CH = int(sys.argv[4]), 8, 37, 2
KC = 16 // WORLD   # columns per rank and chunk
ROWS = N // WORLD
witness = (np.arange(W * N, dtype=np.uint64).reshape(W, N) + np.uint64(9)) * np.uint64(0x10003)
fc = FakeCtx()
class FakeLib:
    def lcp2_ntt_batch(self, handle, ptr, ncols, log_n, inverse, coset, mem):  # stand-in "iNTT": x -> 5 x + 2, in place
        v = fc._at(ptr.value, ncols * N); v[:] = v * np.uint64(5) + np.uint64(2); return 0
fc.lib, fc.handle, fc._check = FakeLib(), None, (lambda rc: None)
class FakeParams:
    cap_height, num_challenges, rate_bits, degree_bits, num_wires = 4, CH, 3, 3, W
class FakeCirc:
    params = FakeParams()
class ChunkData:
    def __init__(self, rank): self.rank = rank
    def commit_wires_rows_begin(self, rows_ptr):
        rows = fc.buffer_read(rows_ptr, 48 * ROWS).reshape(48, ROWS)[:W]
        assert (rows == witness[:, ROWS * self.rank:ROWS * (self.rank + 1)]).all(), rows   # this rank's rows of EVERY column, in column order
        self.seen = []
    def commit_wires_chunk(self, coeffs_ptr, first_col, ncols):
        got = fc.buffer_read(coeffs_ptr, ncols * N).reshape(ncols, N)
        assert (got == witness[first_col:first_col + ncols] * np.uint64(5) + np.uint64(2)).all(), (first_col, got)  # complete when it is absorbed
        self.seen.append((first_col, ncols))
    def commit_wires_rows_finish(self):
        assert self.seen == [(0, 16), (16, 16), (32, 5)], self.seen
        return np.full((16, 4), 40 + self.rank, dtype=np.uint64)
class Comm4(m.parallel.TorchComm):  # the library buffers of the stand-in are numpy arrays
    def _view(self, ptr, words): return torch.from_numpy(fc._at(ptr, words).view(np.int64))
    def all_gather_device(self, ptr, total_words, words_per_rank):
        self.all_gather_tensor(self._view(ptr, total_words), rank); self.bytes_gathered += 8 * words_per_rank * (WORLD - 1)
    def all_to_all_device(self, send_ptr, recv_ptr, words_per_pair):
        self.all_to_all_tensor(self._view(recv_ptr, WORLD * words_per_pair), self._view(send_ptr, WORLD * words_per_pair))
        self.bytes_gathered += 8 * words_per_pair * (WORLD - 1)
sp = object.__new__(m.parallel.ShardedProver)
sp.b, sp.ctx, sp.circ, sp.rank, sp.world, sp.comm = m.binding, fc, FakeCirc(), rank, WORLD, Comm4(dist)
sp._vals = sp._coeffs = sp._row_bufs = sp._chunk_bufs = None
sp.data = ChunkData(rank)
mine = m.parallel.chunk_columns(W, rank, WORLD)
assert mine == [c for j in range(3) for c in range(16 * j + KC * rank, 16 * j + KC * rank + KC) if c < W], mine
assert sorted(c for r in range(WORLD) for c in m.parallel.chunk_columns(W, r, WORLD)) == list(range(W))
steps = sp._commit_wires_chunked(witness[mine].copy(), 0)
reply = None
try:  # the dispatch of ShardedProver.prove() for the requests this generator makes
    while True:
        req = steps.send(reply)
        if req[0] == "all_gather_async": reply = sp.comm.all_gather_device(req[1], req[2], req[3])  # gloo: gathered on the spot
        elif req[0] == "all_to_all_device": reply = sp.comm.all_to_all_device(req[1], req[2], req[3])
        elif req[0] == "wait": assert req[1] is None; reply = None
        else: raise SystemExit("unexpected request %r" % (req[0],))
except StopIteration as done:
    assert (done.value == 40 + rank).all()
# received: 3 chunks of 16 columns minus the own 4 per chunk slot (padded slots included), and the other ranks' row blocks
assert sp.comm.bytes_gathered == 8 * (3 * (16 - KC) * N + 3 * (16 - KC) * ROWS), sp.comm.bytes_gathered
# ---- the whole proof over the same ranks: ShardedProver.prove(..., row_exchange=True, chunked=True) with a stand-in for the per-rank compute
# that checks what each exchange delivers (the complete table of block products, the complete Z / partial-product buffer, complete
# quotient planes) and returns disjoint shares the way the C ABI does
NCZ, E = 4, 16 // WORLD
class ProofData(ChunkData):
    proof_words = 3 * 64 + 40
    def __init__(self, rank):
        self.rank, self.qbuf, self.zbuf = rank, np.zeros(CH * 8 * WORLD, dtype=np.uint64), np.zeros(NCZ * N, dtype=np.uint64)
        fc.bufs[self.qbuf.ctypes.data], fc.bufs[self.zbuf.ctypes.data] = self.qbuf, self.zbuf   # "device" buffers of the stand-in
    def _share(self, tag):
        c = np.zeros((16, 4), dtype=np.uint64)
        c[E * self.rank:E * self.rank + E] = np.arange(4 * E, dtype=np.uint64).reshape(E, 4) * np.uint64(7) + np.uint64(tag + 100 * self.rank)
        return c
    def commit_wires_rows_finish(self):
        ChunkData.commit_wires_rows_finish(self); return self._share(1)
    def perm_zs_rows_begin(self, betas, gammas, world):
        assert world == WORLD
        out = np.zeros(world * CH, dtype=np.uint64); out[CH * self.rank:CH * self.rank + CH] = [11 + self.rank, 21 + self.rank]; return out
    def perm_zs_rows_finish(self, products):
        assert list(products) == [v for r in range(WORLD) for v in (11 + r, 21 + r)]     # every rank's block products
        z = self.zbuf.reshape(WORLD, NCZ, ROWS); z[self.rank] = np.arange(NCZ * ROWS, dtype=np.uint64).reshape(NCZ, ROWS) + np.uint64(1000 * (self.rank + 1))
        return self.zbuf.ctypes.data, self.zbuf.size
    def perm_zs_commit(self):
        want = np.stack([np.arange(NCZ * ROWS, dtype=np.uint64).reshape(NCZ, ROWS) + np.uint64(1000 * (r + 1)) for r in range(WORLD)])
        assert (self.zbuf.reshape(WORLD, NCZ, ROWS) == want).all()                        # every rank's rows of the Z columns
        return self._share(2)
    def quotient_values(self, alphas, pi_hash):  # this rank's contiguous run of each challenge plane
        for c in range(CH): self.qbuf[8 * WORLD * c + 8 * self.rank:8 * WORLD * c + 8 * self.rank + 8] = np.uint64(70 + 10 * c + self.rank)
    def quotient_buffer(self): return (self.qbuf.ctypes.data, self.qbuf.size)
    def quotient_commit(self):
        assert self.qbuf.tolist() == [70 + 10 * c + r for c in range(CH) for r in range(WORLD) for _ in range(8)]
        return self._share(3)
    def proof_section(self, which): return {0: (192, 8), 1: (200, 8), 2: (192, 40)}[which]
    def fri_open_begin(self, zeta, state, proof): proof[192 + self.rank % 8] = 5 + self.rank
    def fri_open_commit(self, proof): proof[200 + self.rank % 8] = 60 + self.rank
    def fri_open_finish(self, proof):  # like the library: words every rank already holds in full (summed sections) stay on rank 0 only
        if self.rank: proof[192:208] = 0
        proof[208 + 3 * self.rank:211 + 3 * self.rank] = 9
sp.data = ProofData(rank)
sp.digest = np.arange(4, dtype=np.uint64)
sp.comm.bytes_gathered = 0
proof = sp.prove(witness[mine].copy(), np.array([3, 4], dtype=np.uint64), sharded_columns=True, row_exchange=True, chunked=True)
everyone = [None] * WORLD
dist.all_gather_object(everyone, proof.tobytes())
assert all(p == everyone[0] for p in everyone)                                           # one proof on every rank
assert (proof[:192].reshape(3, 16, 4) != 0).all() and proof[192:192 + WORLD].tolist() == [5 + r for r in range(WORLD)]
assert proof[200:200 + WORLD].tolist() == [60 + r for r in range(WORLD)] and proof[208:208 + 3 * WORLD].tolist() == [9] * (3 * WORLD)
# received on top of the coefficient chunks and row blocks: the other ranks' Z rows and their runs of both quotient planes
assert sp.comm.bytes_gathered == 8 * (3 * (16 - KC) * N + 3 * (16 - KC) * ROWS + NCZ * ROWS * (WORLD - 1) + CH * 8 * (WORLD - 1)), sp.comm.bytes_gathered
# the primitives with four peers: whole, in pieces (per-pair limit, per-call limit), staged
K = 6
for pair, call, staged in ((1 << 26, 1 << 27, False), (4, 1 << 27, False), (1 << 26, 8, False), (1 << 26, 1 << 27, True)):
    cm = Comm4(dist, staged=staged); cm.A2A_WORDS_PER_PAIR, cm.A2A_WORDS_PER_CALL = pair, call
    snd = torch.arange(WORLD * K, dtype=torch.int64) + 1000 * rank    # part d goes to rank d
    rcv = torch.zeros(WORLD * K, dtype=torch.int64)
    cm.all_to_all_tensor(rcv, snd)
    assert rcv.tolist() == [1000 * s + K * rank + j for s in range(WORLD) for j in range(K)], (pair, call, staged, rcv.tolist())
for limit, staged in ((1 << 27, False), (4, False), (1 << 27, True)):
    cm = Comm4(dist, staged=staged); cm.GATHER_WORDS_PER_RANK = limit
    buf = torch.zeros(WORLD * K, dtype=torch.int64); buf[K * rank:K * rank + K] = torch.arange(K) + 50 * (rank + 1)
    cm.all_gather_tensor(buf, rank)
    assert buf.tolist() == [50 * (r + 1) + j for r in range(WORLD) for j in range(K)], (limit, staged, buf.tolist())
# shares of a 16-entry cap from four ranks (four entries each) and the verdict word
share = np.zeros((16, 4), dtype=np.uint64); share[E * rank:E * rank + E] = np.uint64(2**63 + 5 + rank)
full = sp.comm.sum_host(share)
assert [int(full[E * r, 0]) for r in range(WORLD)] == [2**63 + 5 + r for r in range(WORLD)]
assert m.parallel.block_range(rank, WORLD) == (8 // WORLD * rank, 8 // WORLD)
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
"""


@pytest.mark.parametrize("world", [4, 8])
def test_chunked_exchange_over_four_and_eight_gloo_ranks(tmp_path, world):
    """world_size-4 and -8 gloo: ShardedProver._commit_wires_chunked (37 wires in chunks of 16, 16 and 5 columns, four or two columns per rank and chunk,
    ragged last chunk, one row per rank at eight) with every chunk complete and in column order when it reaches the commitment and every rank holding exactly its own
    rows of every column; then the WHOLE proof (prove(..., row_exchange=True, chunked=True)): the complete table of block products, the complete Z rows and
    complete quotient planes reach every rank and all ranks assemble the same proof; all_to_all_tensor / all_gather_tensor whole, in pieces and staged between the peers"""
    script = tmp_path / "worker4.py"
    script.write_text(_GLOO_WORLD4_WORKER)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), str(world)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ok {r}" in o, o
