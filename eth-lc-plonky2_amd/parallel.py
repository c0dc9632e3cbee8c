"""Coset-sharded polynomial commitment across the GPUs of one node (BASELINE configs[3]).

    rank g owns a contiguous shard of the trace COLUMNS           (polynomial-parallel: K1 iNTT, no communication)
    -> RCCL all-gather of the coefficient shards over xGMI        (the one exchange step of a commitment:
                                                                   4.5 GB for the 135 wire columns at n = 2^22)
    -> rank g computes the LDE COSETS of its leaf blocks for every column, hashes those leaves and builds its own
       Merkle subtrees                                            (coset-parallel: K2, K4; zero cross-GPU hashing)
    -> all-gather of 2^(cap_height - rate_bits) cap entries per block (512 B in total)

In Merkle leaf order coset r of the LDE is the contiguous leaf block bitrev(r), and with cap_height = 4 >= rate_bits = 3
every block is two whole cap subtrees, so the commitment (the cap) is bit-identical to the single-GPU one.
The compute steps are passed in as callables so that the same orchestration runs on RCCL with the HIP kernels
(`gpu_ops`) and on gloo with the oracle in the CPU tests.
"""
import ctypes

import numpy as np


def column_shards(ncols, world_size):
    base, extra = divmod(ncols, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


CHUNK_COLS = 16  # columns per chunk of the chunked coefficient exchange: two permutations of a leaf's sponge (a multiple of 8)


def chunk_columns(num_wires, rank, world_size, chunk=CHUNK_COLS):
    """Chunked coefficient exchange (ShardedProver.prove_steps(..., chunked=True)): chunk j is the columns [C j, C j + C), C = 16 - what
    two permutations of a leaf's sponge absorb - and inside a chunk rank r owns the k = C / world columns [C j + r k, C j + (r + 1) k):
    every chunk is an in-place all-gather with one contiguous piece per rank.  Returns the columns rank `rank` brings, in increasing order.
    (8 columns per chunk cost 1.5 ms more per rank at 8 ranks - 17 small LDE and absorption launches instead of 9 - and hide nothing more.)"""
    if world_size not in (1, 2, 4, 8) or chunk % 8 or chunk % world_size:
        raise ValueError("chunked exchange: 1, 2, 4 or 8 ranks, chunks of a multiple of 8 columns")
    k = chunk // world_size
    return [c for c in range(num_wires) if (c % chunk) // k == rank]


def block_range(rank, world_size, rate_bits=3):
    nblocks = 1 << rate_bits
    if world_size < 1 or nblocks % world_size or (world_size & (world_size - 1)):
        raise ValueError("world size must be a power of two that divides 2^rate_bits (1, 2, 4 or 8 GPUs)")
    count = nblocks // world_size
    return rank * count, count


class ShardedCommitment:
    def __init__(self, local, cap, block_first, block_count, ncols, log_n):
        self.local, self.cap, self.block_first, self.block_count, self.ncols, self.log_n = local, cap, block_first, block_count, ncols, log_n

    def owner_of_leaf(self, leaf_index, world_size, rate_bits=3):
        """rank that holds a global leaf index, and its local index there"""
        n = 1 << self.log_n
        per_rank = ((1 << rate_bits) // world_size) * n
        return leaf_index // per_rank, leaf_index % per_rank


def sharded_commit(values_shard, ncols, log_n, rank, world_size, ops, dist=None, rate_bits=3, cap_height=4):
    """values_shard: this rank's columns [my_cols][n] (whatever array type `ops` understands).
    ops: intt(values_shard) -> coefficient shard ; gather_columns(shard, shards, dist) -> all coefficients [ncols][n] ;
         commit_blocks(coeffs, block_first, block_count) -> (local oracle, cap_part numpy [count * 2^(cap-rate)][4]) ;
         gather_caps(cap_part, dist) -> full cap numpy [2^cap_height][4]"""
    shards = column_shards(ncols, world_size)
    first, count = block_range(rank, world_size, rate_bits)
    coeff_shard = ops.intt(values_shard)
    coeffs = ops.gather_columns(coeff_shard, shards, rank, world_size, dist)
    local, cap_part = ops.commit_blocks(coeffs, first, count)
    cap = ops.gather_caps(cap_part, rank, world_size, dist)
    return ShardedCommitment(local, cap, first, count, ncols, log_n)


class GpuOps:
    """HIP kernels through the C ABI + torch.distributed (nccl = RCCL) for the exchange; tensors are torch int64 on the GPU."""

    def __init__(self, ctx, n, rate_bits=3, cap_height=4):
        self.ctx, self.n, self.rate_bits, self.cap_height = ctx, n, rate_bits, cap_height

    def intt(self, shard):
        import ctypes
        from . import binding as b
        if shard.shape[0]:
            self.ctx._check(self.ctx.lib.lcp2_ntt_batch(self.ctx.handle, ctypes.c_void_p(shard.data_ptr()), shard.shape[0],
                                                        int(self.n).bit_length() - 1, 1, 1, b.MEM_DEVICE))
        return shard  # in place: values -> coefficients

    def gather_columns(self, shard, shards, rank, world, dist):
        import torch
        if world == 1:
            return shard
        most = max(e - s for s, e in shards)
        pad = torch.zeros((most, self.n), dtype=shard.dtype, device=shard.device)
        pad[:shard.shape[0]] = shard
        buf = torch.empty((world, most, self.n), dtype=shard.dtype, device=shard.device)
        dist.all_gather_into_tensor(buf, pad)  # one message per peer: xGMI links are driven concurrently
        return torch.cat([buf[r, :e - s] for r, (s, e) in enumerate(shards)])

    def commit_blocks(self, coeffs, first, count):
        from . import binding as b
        o = self.ctx.commit_cosets(coeffs.data_ptr(), first, count, self.rate_bits, self.cap_height, mem=b.MEM_DEVICE,
                                   shape=tuple(coeffs.shape))
        return o, o.cap

    def gather_caps(self, cap_part, rank, world, dist):
        import torch
        if world == 1:
            return cap_part
        t = torch.from_numpy(cap_part.view(np.int64)).cuda()
        buf = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf, t)
        return buf.cpu().numpy().view(np.uint64).reshape(-1, 4)


# ---------------------------------------------------------------------------------------------------------------------
# One whole proof sharded by LDE coset (include/lcp2.h "one proof sharded over the GPUs of a node").
class ShardedProver:
    """Rank `rank` of `world` in a coset-sharded proof: the rank holds 8 / world leaf blocks of every LDE and Merkle tree and
    runs the seams of data.prove() on them.  What crosses the ranks:
        witness        column-sharded arrival (prove_steps(..., sharded_columns=True)): every rank uploads only its column
                       shard, the values are all-gathered over xGMI, every rank transforms its own columns (polynomial-parallel
                       iNTT) and the coefficients are all-gathered: the commitment's "column transpose" 
        caps, proof    shares (own entries at their global position, zeros elsewhere): SUM all-reduce of 512-byte caps and of
                       the 140 KB proof array (uint64 wrap-around; RCCL has no bitwise reductions)
        openings, FRI  a rank evaluates its share of the columns at zeta and commits its leaf blocks of FRI layer 0 (folding is in
                       coefficient form: no FRI values cross the ranks); shares of the 4 KB of openings and of the layer-0 cap
        witness, row exchange form (prove_steps(..., sharded_columns=True, row_exchange=True)): the values are needed only
                       by the permutation argument and the gate check, both row-wise, so instead of all-gathering them each rank
                       receives its block of n / world rows of every column by an all-to-all (1 / world of the bytes), runs K5 and
                       the check on those rows, and the Z / partial-product rows are all-gathered (20 columns instead of 135):
                       8.4 -> 5.5 GB received per rank at n = 2^22 over 8 ranks
        quotient       each challenge plane of the value buffer is completed by an in-place all-gather of the ranks' contiguous
                       leaf blocks (537 MB in total at n = 2^22, each byte crosses the fabric once)
    `comm` supplies the collectives, so that the same orchestration runs over torch.distributed (TorchComm: "nccl" = RCCL on
    the node, "gloo" in the CPU tests) and, in the single-GPU tests, over ranks stepped in lockstep inside one process:
        comm.sum_host(numpy uint64 array) -> numpy uint64 array
        comm.all_gather_device(device pointer, total words, words per rank)    in place: rank r's part sits at r * words per rank
        comm.all_to_all_device(send pointer, receive pointer, words per pair)  part d of the send buffer goes to rank d, part s of
                                                                               the receive buffer comes from rank s"""

    def __init__(self, ctx, circ, rank, world, comm, constants_sigmas_ptr=None, mem=0):
        from . import binding as b
        self.b, self.ctx, self.circ, self.rank, self.world, self.comm = b, ctx, circ, rank, world, comm
        first, count = block_range(rank, world, circ.params.rate_bits)
        self.data = b.CircuitData.build_sharded(ctx, circ, first, count, constants_sigmas_ptr, mem)
        self.cap_share = self.data.digest()[1]
        self._vals = self._coeffs = None
        self._row_bufs = None

    def finish_build(self):
        self.data.set_constants_cap(self.comm.sum_host(self.cap_share))
        self.digest = self.data.digest()[0]

    def column_shard(self):
        """(first, end) of the witness columns this rank brings in the column-sharded arrival"""
        return column_shards(self.circ.params.num_wires, self.world)[self.rank]

    def _witness_buffers(self, most):
        n = 1 << self.circ.params.degree_bits
        words = self.world * most * n
        if self._vals is None:
            self._vals = self.ctx.buffer_alloc(words)
        if self._coeffs is None:
            self._coeffs = self.ctx.buffer_alloc(words)
        return self._vals, self._coeffs

    def _close_gaps(self, buf, shards, most, run):
        """a padded per-rank layout [rank][most][run] -> [column][run] when a shard in the middle is short: column by column, in
        increasing order (source and destination never overlap)"""
        if all(e - s == most for s, e in shards[:-1]):
            return
        for r, (s0, e0) in enumerate(shards):
            for col in range(e0 - s0):
                if r * most != s0:
                    self.ctx.buffer_copy(buf + 8 * (s0 + col) * run, buf + 8 * (r * most + col) * run, run)

    # the proof as a generator of exchange points, so that a test can interleave several ranks in one process
    def prove_steps(self, wires, public_inputs, mem=0, sharded_columns=False, row_exchange=False, chunked=False):
        """wires: the whole witness [num_wires][n] (numpy, or a device pointer with mem=MEM_DEVICE); with sharded_columns=True
        only this rank's columns [column_shard()][n].  row_exchange (with sharded_columns): the values cross the ranks as row
        blocks (all-to-all) instead of whole columns (all-gather), see the class comment.  chunked (with row_exchange): this rank
        brings the columns chunk_columns(num_wires, rank, world) and the coefficient exchange is overlapped with the commitment,
        chunk by chunk (_commit_wires_chunked)."""
        b, d, p, ctx = self.b, self.data, self.circ.params, self.ctx
        capw = 4 << p.cap_height
        n = 1 << p.degree_bits
        proof = np.zeros(d.proof_words, dtype=np.uint64)
        pis = np.asarray(public_inputs, dtype=np.uint64)
        ch = b.Challenger()
        ch.observe(self.digest)
        pi_hash = b.hash_no_pad(pis)
        ch.observe(pi_hash)
        if row_exchange and not (sharded_columns and n >= self.world):
            raise ValueError("row_exchange needs the column-sharded arrival and at least one row per rank")
        if chunked and not row_exchange:
            raise ValueError("the chunked coefficient exchange belongs to the row exchange form")
        if chunked:
            share = yield from self._commit_wires_chunked(wires, mem)
        elif sharded_columns:
            shards = column_shards(p.num_wires, self.world)
            most = max(e - s for s, e in shards)
            first, end = shards[self.rank]
            mine = self.rank * most * n  # word offset of this rank's slot
            if row_exchange:
                # own columns [most][n]; their row blocks for every rank [world][most][rows]; the blocks received [world][most][rows]
                if self._row_bufs is None:
                    self._row_bufs = [ctx.buffer_alloc(most * n) for _ in range(3)]
                if self._coeffs is None:
                    self._coeffs = ctx.buffer_alloc(self.world * most * n)
                own, send, recv = self._row_bufs
                coeffs = self._coeffs
            else:
                vals, coeffs = self._witness_buffers(most)
                own = vals + 8 * mine
            if mem == b.MEM_HOST:
                ctx.buffer_write(own, np.ascontiguousarray(wires, dtype=np.uint64).reshape(end - first, n))
            else:
                ctx.buffer_copy(own, wires, (end - first) * n)
            if not row_exchange:
                yield ("all_gather_device", vals, self.world * most * n, most * n)
            # polynomial-parallel iNTT: own columns only, then the coefficient all-gather
            ctx.buffer_copy(coeffs + 8 * mine, own, (end - first) * n)
            if end > first:
                ctx._check(ctx.lib.lcp2_ntt_batch(ctx.handle, ctypes.c_void_p(coeffs + 8 * mine), end - first, p.degree_bits, 1, 1, b.MEM_DEVICE))
            yield ("all_gather_device", coeffs, self.world * most * n, most * n)
            self._close_gaps(coeffs, shards, most, n)
            if row_exchange:
                rows = n // self.world
                for dst in range(self.world):  # rank dst's rows of this rank's columns
                    ctx.buffer_copy_2d(send + 8 * dst * most * rows, rows, own + 8 * dst * rows, n, rows, end - first)
                yield ("all_to_all_device", send, recv, most * rows)
                self._close_gaps(recv, shards, most, rows)
                share = d.commit_wires_rows(recv, coeffs)
            else:
                self._close_gaps(vals, shards, most, n)
                share = d.commit_wires_coeffs(vals, coeffs)
        else:
            share = d.commit_wires(wires, mem)
        proof[0:capw] = (yield ("sum_host", share)).ravel()
        ch.observe(proof[0:capw])
        betas, gammas = ch.get(p.num_challenges), ch.get(p.num_challenges)
        if row_exchange:
            products = yield ("sum_host", d.perm_zs_rows_begin(betas, gammas, self.world))
            zptr, zwords = d.perm_zs_rows_finish(products)  # raises on every rank alike if the product does not return to 1
            yield ("all_gather_device", zptr, zwords, zwords // self.world)
            share = d.perm_zs_commit()
        else:
            share = d.perm_zs(betas, gammas)
        proof[capw:2 * capw] = (yield ("sum_host", share)).ravel()
        ch.observe(proof[capw:2 * capw])
        alphas = ch.get(p.num_challenges)
        if row_exchange:  # each rank checks the gates on its own rows: the verdict is exchanged before the next collective
            failure = None
            try:
                d.quotient_values(alphas, pi_hash)
            except b.Lcp2Error as e:
                if e.status != b.E_UNSAT:
                    raise
                failure = e
            verdicts = yield ("sum_host", np.array([failure is not None], dtype=np.uint64))
            if failure is not None:
                raise failure
            if int(verdicts[0]):
                raise b.Lcp2Error(b.E_UNSAT, "another rank reports a violated gate constraint")
        else:
            d.quotient_values(alphas, pi_hash)
        qptr, qwords = d.quotient_buffer()
        plane = qwords // p.num_challenges  # 8n words per challenge; this rank's blocks are one contiguous run of it
        for c in range(p.num_challenges):
            yield ("all_gather_device", qptr + 8 * c * plane, plane, plane // self.world)
        share = d.quotient_commit()
        proof[2 * capw:3 * capw] = (yield ("sum_host", share)).ravel()
        ch.observe(proof[2 * capw:3 * capw])
        zeta = ch.get(2)
        # the opening stage in its three phases: this rank's columns of the openings, then its leaf blocks of FRI layer 0, then
        # the rest; what the phases leave in the proof array is this rank's share of the section
        d.fri_open_begin(zeta, ch.state, proof)
        lo, cnt = d.proof_section(b.SECTION_OPENINGS)
        proof[lo:lo + cnt] = yield ("sum_host", proof[lo:lo + cnt])
        d.fri_open_commit(proof)
        lo, cnt = d.proof_section(b.SECTION_FRI_CAP0)
        if cnt:
            proof[lo:lo + cnt] = yield ("sum_host", proof[lo:lo + cnt])
        d.fri_open_finish(proof)
        lo, cnt = d.proof_section(b.SECTION_AFTER_CAPS)
        proof[lo:lo + cnt] = yield ("sum_host", proof[lo:lo + cnt])
        self.proof = proof
        return

    def _commit_wires_chunked(self, wires, mem):
        """The wires commitment of the row exchange form with the 4 GB coefficient all-gather OVERLAPPED: the sponge of a leaf absorbs the
        columns in order, 8 per permutation, so chunk j (columns 16 j .. 16 j + 15, two or more per rank) is gathered - asynchronously, on the
        communicator's own stream - while chunk j - 1 runs its coset LDE and is absorbed into the persistent leaf states
        (lcp2_commit_wires_chunk).  Only the first chunk's exchange is exposed."""
        b, d, p, ctx = self.b, self.data, self.circ.params, self.ctx
        n, W, world, rank = 1 << p.degree_bits, p.num_wires, self.world, self.rank
        C = CHUNK_COLS
        k, nch = C // world, -(-W // C)
        cols = chunk_columns(W, rank, world)
        most, rows = nch * k, n // world
        if getattr(self, "_chunk_bufs", None) is None:
            # own values [most][n], own coefficients [most][n], row blocks to send / received [world][most][rows], the rows in column
            # order [C nch][rows], the coefficient columns in column order [C nch][n]
            self._chunk_bufs = [ctx.buffer_alloc(most * n), ctx.buffer_alloc(most * n), ctx.buffer_alloc(world * most * rows),
                                ctx.buffer_alloc(world * most * rows), ctx.buffer_alloc(C * nch * rows), ctx.buffer_alloc(C * nch * n)]
        own, cown, send, recv, rowsbuf, cbase = self._chunk_bufs
        if mem == b.MEM_HOST:
            ctx.buffer_write(own, np.ascontiguousarray(wires, dtype=np.uint64).reshape(len(cols), n))
        else:
            ctx.buffer_copy(own, wires, len(cols) * n)
        # polynomial-parallel iNTT of the own columns, then every own coefficient column to its place in column order
        ctx.buffer_copy(cown, own, len(cols) * n)
        if cols:
            ctx._check(ctx.lib.lcp2_ntt_batch(ctx.handle, ctypes.c_void_p(cown), len(cols), p.degree_bits, 1, 1, b.MEM_DEVICE))
        count = lambda r, t: sum(1 for j in range(nch) if C * j + r * k + t < W)  # noqa: E731  chunks in which column slot (r, t) exists
        for t in range(k):
            if count(rank, t):
                ctx.buffer_copy_2d(cbase + 8 * (rank * k + t) * n, C * n, cown + 8 * t * n, k * n, n, count(rank, t))
        # What this rank contributes to EVERY chunk is in place now (the copies above), so only the first gather is ordered against the
        # library's stream; the later ones ("later": the communicator's stream has already waited for that point) do not wait for the chunk
        # that is being extended when they are issued.  That matters when the host blocks inside a collective (TORCH_NCCL_BLOCKING_WAIT, what
        # bench.py sets): the gather of chunk j + 2 then runs while chunk j - 1 is still on the device, and chunk j is queued behind it in time.
        handles = []
        for j in range(min(2, nch)):  # two chunks in flight
            handles.append((yield ("all_gather_async", cbase + 8 * C * j * n, C * n, k * n, "later" if j else "first")))
        # the witness values as row blocks (all-to-all), then into column order
        for dst in range(world):
            if cols:
                ctx.buffer_copy_2d(send + 8 * dst * most * rows, rows, own + 8 * dst * rows, n, rows, len(cols))
        yield ("all_to_all_device", send, recv, most * rows)
        for src in range(world):
            for t in range(k):
                if count(src, t):
                    ctx.buffer_copy_2d(rowsbuf + 8 * (src * k + t) * rows, C * rows, recv + 8 * (src * most + t) * rows, k * rows, rows, count(src, t))
        d.commit_wires_rows_begin(rowsbuf)
        for j in range(nch):
            yield ("wait", handles[j])
            if j + 2 < nch:
                handles.append((yield ("all_gather_async", cbase + 8 * C * (j + 2) * n, C * n, k * n, "later")))
            d.commit_wires_chunk(cbase + 8 * C * j * n, C * j, min(C, W - C * j))
        return d.commit_wires_rows_finish()

    def prove(self, wires, public_inputs, mem=0, sharded_columns=False, row_exchange=False, chunked=False):
        steps = self.prove_steps(wires, public_inputs, mem, sharded_columns, row_exchange, chunked)
        reply = None
        try:
            while True:
                req = steps.send(reply)
                if req[0] == "sum_host":
                    reply = self.comm.sum_host(req[1])
                elif req[0] == "all_to_all_device":
                    self.comm.all_to_all_device(req[1], req[2], req[3])
                    reply = None
                elif req[0] == "all_gather_async":  # a communicator without the asynchronous form gathers on the spot
                    start = getattr(self.comm, "all_gather_device_async", None)
                    reply = (start(req[1], req[2], req[3], source_ready=len(req) > 4 and req[4] == "later") if start
                             else self.comm.all_gather_device(req[1], req[2], req[3]))
                elif req[0] == "wait":
                    if req[1] is not None:
                        self.comm.wait(req[1])
                    reply = None
                else:
                    self.comm.all_gather_device(req[1], req[2], req[3])
                    reply = None
        except StopIteration:
            return self.proof

    def close(self):
        for buf in [self._vals, self._coeffs] + list(self._row_bufs or []) + list(getattr(self, "_chunk_bufs", None) or []):
            if buf:
                self.ctx.buffer_free(buf)
        self._vals = self._coeffs = self._row_bufs = self._chunk_bufs = None
        self.data.close()


class _DevicePtr:
    """zero-copy view of a raw device allocation for torch.as_tensor (CUDA array interface v2)"""

    def __init__(self, ptr, words):
        self.__cuda_array_interface__ = {"shape": (int(words),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


class TorchComm:
    """torch.distributed collectives for ShardedProver: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" on CPU.

    The all-gather runs IN PLACE on the library's own buffer (a zero-copy tensor view of it) so that every byte crosses the
    fabric once.  `LCP2_SHARDED_STAGED=1` (or `staged=True`) selects the conservative form instead: the rank's part is cloned
    into a torch tensor, gathered into a torch-owned buffer and copied back.  `self_check()` runs a small in-place gather with
    known contents first and falls back to the staged form if the result is not what every rank wrote - the in-place form over
    RCCL with more than one rank has not run on hardware yet."""

    # Message sizes beyond these go in pieces: collectives with more than 1 GiB per peer are the rare case in RCCL's use (and its
    # one-rank all-to-all demonstrably mishandles them, tools/rccl_a2a_probe.py); at 8 ranks and n = 2^22 nothing is split
    A2A_WORDS_PER_PAIR = 1 << 26     # 512 MiB per pair
    A2A_WORDS_PER_CALL = 1 << 27     # 1 GiB per call and rank: the one measured defect (profiles/r03_rccl_a2a_probe.log, world 1) is keyed on the
                                     # TOTAL size of a call; at n = 2^22 over 4 ranks an unsplit call would carry 4 x 285 MB.  Both bounds are guesses
                                     # about multi-rank RCCL (it has not run here): the proof's verification stays the acceptance signal
    GATHER_WORDS_PER_RANK = 1 << 27  # 1 GiB per rank

    def __init__(self, dist, device=None, ctx=None, staged=None):
        import os
        self.dist, self.device, self.ctx = dist, device, ctx
        self.staged = bool(int(os.environ.get("LCP2_SHARDED_STAGED", "0"))) if staged is None else bool(staged)
        self.bytes_gathered = 0  # received by this rank through all_gather_device / all_to_all_device (exchange accounting of the bench)
        self.row_exchange_ok = True  # self_check(): the all-to-all of the row exchange form reproduces a known answer
        self.seconds = {"all_gather": 0.0, "all_to_all": 0.0, "all_reduce": 0.0}  # wall time inside the (synchronous) collectives
        self._side = self._lib_stream = None  # all_gather_device_async: the communication stream and the library's stream, made at the first use
        self._side_ordered = False            # ... and whether the communication stream has been ordered behind the library's stream yet

    def sum_host(self, arr):
        import time
        import torch
        t0 = time.perf_counter()
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy())
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)  # two's-complement wrap-around = uint64 addition
        out = t.cpu().numpy().view(np.uint64).reshape(np.shape(arr))
        self.seconds["all_reduce"] += time.perf_counter() - t0
        return out

    def all_gather_tensor(self, out, rank):
        """all-gather on a 1-D tensor whose part r is out[r * k : (r + 1) * k].  In place over RCCL (it recognises the aliasing
        and moves every byte once); gloo and the staged form get a private copy of the input."""
        world = self.dist.get_world_size()
        k = out.numel() // world
        mine = out[rank * k:(rank + 1) * k]
        if world > 1 and k > self.GATHER_WORDS_PER_RANK:  # parts above 1 GiB in pieces, each through a staging tensor
            import torch
            for off in range(0, k, self.GATHER_WORDS_PER_RANK):
                end = min(off + self.GATHER_WORDS_PER_RANK, k)
                got = torch.empty(world * (end - off), dtype=out.dtype, device=out.device)
                self.dist.all_gather_into_tensor(got, mine[off:end].clone())
                for r in range(world):
                    out[r * k + off:r * k + end].copy_(got[r * (end - off):(r + 1) * (end - off)])
            return
        if self.staged:
            import torch
            tmp = torch.empty_like(out)
            self.dist.all_gather_into_tensor(tmp, mine.clone())
            out.copy_(tmp)
            return
        if self.dist.get_backend() != "nccl":
            mine = mine.clone()
        self.dist.all_gather_into_tensor(out, mine)

    def all_gather_device(self, ptr, total_words, words_per_rank):
        import time
        import torch
        world = self.dist.get_world_size()
        assert total_words == words_per_rank * world
        self.ctx.sync()  # the library's stream has written this rank's part
        t0 = time.perf_counter()
        out = torch.as_tensor(_DevicePtr(ptr, total_words), device=self.device)  # aliases the library's buffer: no staging copies
        self.all_gather_tensor(out, self.dist.get_rank())
        torch.cuda.synchronize(self.device)
        self.seconds["all_gather"] += time.perf_counter() - t0
        self.bytes_gathered += 8 * words_per_rank * (world - 1)

    def all_gather_device_async(self, ptr, total_words, words_per_rank, source_ready=False):
        """all_gather_device on the communicator's own stream: returns an event; the library's stream (wrapped by its handle,
        lcp2_ctx_stream) goes on with the chunk before.  What this rank contributes was written by work already queued on the library's
        stream: the side stream waits for it - unless the caller says it was complete when an EARLIER asynchronous gather was issued
        (source_ready: the side stream, being in order, is already behind that point), in which case the gather does not wait for whatever
        the library's stream has been given since.  (With TORCH_NCCL_BLOCKING_WAIT the HOST blocks inside the call until the collective
        is done - the kernels queued before it still overlap it.)  Other backends: the synchronous form."""
        import time
        import torch
        if self.dist.get_backend() != "nccl" or self.staged:
            self.all_gather_device(ptr, total_words, words_per_rank)
            return None
        world = self.dist.get_world_size()
        assert total_words == words_per_rank * world
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._lib_stream = torch.cuda.ExternalStream(self.ctx.stream_ptr(), device=self.device)  # the context's own stream, by handle
        t0 = time.perf_counter()
        if not (source_ready and self._side_ordered):
            ready = torch.cuda.Event()
            ready.record(self._lib_stream)
            self._side.wait_event(ready)
            self._side_ordered = True  # (a later source_ready gather relies on this wait having been queued on the side stream)
        with torch.cuda.stream(self._side):
            out = torch.as_tensor(_DevicePtr(ptr, total_words), device=self.device)
            self.all_gather_tensor(out, self.dist.get_rank())
            done = torch.cuda.Event()
            done.record(self._side)
        self.seconds["all_gather"] += time.perf_counter() - t0  # host time inside the call (not the transfer, which runs on)
        self.bytes_gathered += 8 * words_per_rank * (world - 1)
        return done

    def wait(self, done):
        """the library's stream waits for an asynchronous gather; the host does not"""
        self._lib_stream.wait_event(done)

    def all_to_all_tensor(self, recv, send):
        """part d of `send` goes to rank d, part s of `recv` comes from rank s (1-D tensors of world * k words)"""
        world = self.dist.get_world_size()
        k = send.numel() // world
        if world == 1:
            recv.copy_(send)  # (RCCL 2.26's one-rank all-to-all moves only half of a message above 1 GiB: tools/rccl_a2a_probe.py)
        elif self.staged:  # torch-owned tensors on both sides
            import torch
            tmp = torch.empty_like(recv)
            self.dist.all_to_all_single(tmp, send.clone())
            recv.copy_(tmp)
        elif k <= self.A2A_WORDS_PER_PAIR and world * k <= self.A2A_WORDS_PER_CALL:
            self.dist.all_to_all_single(recv, send)
        else:  # large messages in pieces, each through a packed staging pair; a piece is bounded per pair AND per call
            import torch
            step = max(1, min(self.A2A_WORDS_PER_PAIR, self.A2A_WORDS_PER_CALL // world))
            for off in range(0, k, step):
                end = min(off + step, k)
                packed = torch.cat([send[d * k + off:d * k + end] for d in range(world)])
                got = torch.empty_like(packed)
                self.dist.all_to_all_single(got, packed)
                for src in range(world):
                    recv[src * k + off:src * k + end].copy_(got[src * (end - off):(src + 1) * (end - off)])

    def all_to_all_device(self, send_ptr, recv_ptr, words_per_pair):
        """all_to_all_tensor on two library buffers"""
        import time
        import torch
        world = self.dist.get_world_size()
        self.ctx.sync()
        t0 = time.perf_counter()
        send = torch.as_tensor(_DevicePtr(send_ptr, world * words_per_pair), device=self.device)
        recv = torch.as_tensor(_DevicePtr(recv_ptr, world * words_per_pair), device=self.device)
        self.all_to_all_tensor(recv, send)
        torch.cuda.synchronize(self.device)
        self.seconds["all_to_all"] += time.perf_counter() - t0
        self.bytes_gathered += 8 * words_per_pair * (world - 1)

    def _check_all_to_all(self, words_per_pair):
        """known-answer all-to-all on library buffers; the verdict is the same on every rank"""
        world, rank = self.dist.get_world_size(), self.dist.get_rank()
        send, recv = self.ctx.buffer_alloc(world * words_per_pair), self.ctx.buffer_alloc(world * words_per_pair)
        word = lambda src, dst: np.arange(words_per_pair, dtype=np.uint64) * np.uint64(40503) + np.uint64(1000 * src + dst + 1)
        try:
            self.ctx.buffer_write(send, np.concatenate([word(rank, d) for d in range(world)]))
            self.ctx.buffer_write(recv, np.zeros(world * words_per_pair, dtype=np.uint64))
            before = self.bytes_gathered
            try:
                self.all_to_all_device(send, recv, words_per_pair)
                good = bool((self.ctx.buffer_read(recv, world * words_per_pair) == np.concatenate([word(s, rank) for s in range(world)])).all())
            except RuntimeError:
                good = False
            self.bytes_gathered = before
            return int(self.sum_host(np.array([0 if good else 1], dtype=np.uint64))[0]) == 0
        finally:
            self.ctx.buffer_free(send)
            self.ctx.buffer_free(recv)

    def self_check(self, words_per_rank=1 << 16):
        """Known-answer all-gather on a library buffer.  Returns "in-place" or "staged" (the form that will be used), raises if
        neither reproduces what the ranks wrote.  Also tries the all-to-all of the row exchange form: `row_exchange_ok` says
        whether ShardedProver.prove(..., row_exchange=True) can be used (the whole-column form needs all-gathers only)."""
        form = self._check_all_gather(words_per_rank)
        self.row_exchange_ok = self._check_all_to_all(max(words_per_rank // 16, 1))
        return form

    def _check_all_gather(self, words_per_rank):
        world, rank = self.dist.get_world_size(), self.dist.get_rank()
        buf = self.ctx.buffer_alloc(world * words_per_rank)
        want = np.concatenate([(np.arange(words_per_rank, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(r + 1)) for r in range(world)])
        try:
            for attempt in range(2):
                self.ctx.buffer_write(buf, np.zeros(world * words_per_rank, dtype=np.uint64))
                self.ctx.buffer_write(buf + 8 * rank * words_per_rank, want[rank * words_per_rank:(rank + 1) * words_per_rank])
                before = self.bytes_gathered
                try:
                    self.all_gather_device(buf, world * words_per_rank, words_per_rank)
                    good = bool((self.ctx.buffer_read(buf, world * words_per_rank) == want).all())
                except RuntimeError:
                    # an argument check of the framework (the same on every rank: the ranks stay in step); anything raised
                    # from inside the collective itself is not recoverable here and propagates from the staged attempt
                    if self.staged:
                        raise
                    good = False
                self.bytes_gathered = before
                # every rank must take the same decision: the collectives of the two forms differ
                agreed = int(self.sum_host(np.array([0 if good else 1], dtype=np.uint64))[0]) == 0
                if agreed:
                    return "staged" if self.staged else "in-place"
                if self.staged:
                    break
                self.staged = True
        finally:
            self.ctx.buffer_free(buf)
        raise RuntimeError("all-gather self-check failed in both the in-place and the staged form")
