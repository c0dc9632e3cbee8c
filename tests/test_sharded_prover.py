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


def _run_lockstep(m, ctx, circ, wires, pis, world, sharded_columns=False):
    comm = LockstepComm(ctx)
    ranks = [m.parallel.ShardedProver(ctx, circ, r, world, None) for r in range(world)]
    cap = comm.sum_host_all([r.cap_share for r in ranks])
    for r in ranks:
        r.comm = type("C", (), {"sum_host": staticmethod(lambda a, cap=cap: cap)})()
        r.finish_build()
    if sharded_columns:  # every rank brings only its column shard of the witness
        gens = [r.prove_steps(np.ascontiguousarray(wires[slice(*r.column_shard())]), pis, sharded_columns=True) for r in ranks]
    else:
        gens = [r.prove_steps(wires, pis) for r in ranks]
    replies = [None] * world
    while True:
        reqs = []
        for g, rep in zip(gens, replies):
            try:
                reqs.append(g.send(rep))
            except StopIteration:
                reqs.append(None)
        if all(q is None for q in reqs):
            break
        assert all(q is not None and q[0] == reqs[0][0] for q in reqs), "ranks diverged"
        if reqs[0][0] == "sum_host":
            merged = comm.sum_host_all([q[1] for q in reqs])
            replies = [merged.copy() for _ in range(world)]
        else:
            comm.all_gather_device_all([(q[1], q[2], q[3]) for q in reqs])
            replies = [None] * world
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


_GLOO_WORKER = r"""
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import eth_lc_plonky2_amd as m
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
comm = m.parallel.TorchComm(dist)
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
class FakeCtx:  # "device" buffers are numpy arrays, addressed by their data pointer
    def __init__(self): self.bufs = {}
    def buffer_alloc(self, words):
        a = np.zeros(words, dtype=np.uint64); self.bufs[a.ctypes.data] = a; return a.ctypes.data
    def buffer_free(self, ptr): del self.bufs[ptr]
    def _at(self, ptr, words):
        for base, a in self.bufs.items():
            if base <= ptr < base + 8 * a.size: return a[(ptr - base) // 8:(ptr - base) // 8 + words]
        raise KeyError(ptr)
    def buffer_write(self, ptr, arr): self._at(ptr, arr.size)[:] = arr
    def buffer_read(self, ptr, words): return self._at(ptr, words).copy()
    def sync(self): pass
class CheckComm(m.parallel.TorchComm):
    broken, refuse = False, False
    def all_gather_device(self, ptr, total_words, words_per_rank):
        view = torch.from_numpy(self.ctx._at(ptr, total_words).view(np.int64))
        if not self.staged and self.refuse: raise RuntimeError("in-place all-gather refused")
        self.all_gather_tensor(view, rank)
        if not self.staged and self.broken and rank == 1: view[3] += 1   # the aliased form "returns wrong data" on one rank only
        self.bytes_gathered += 8 * words_per_rank
fc = FakeCtx()
assert CheckComm(dist, None, fc).self_check(256) == "in-place"
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
