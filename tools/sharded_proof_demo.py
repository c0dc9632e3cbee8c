#!/usr/bin/env python3
"""One light-client-sized proof sharded by LDE coset over the GPUs of one node (BASELINE configs[3]).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/sharded_proof_demo.py [--degree-bits 22]

Every rank builds the same synthetic circuit and witness (same seed), holds 8/N leaf blocks of every LDE / Merkle tree and
proves through eth-lc-plonky2_amd/parallel.py::ShardedProver: in-place all-gathers of the witness values, the coefficients and
the quotient planes, and small sum all-reduces of the shares (caps, openings, the proof array) over RCCL.  Rank 0 verifies the assembled proof and
prints the wall time per proof.  With N = 1 the same code path runs with one rank holding all 8 blocks (the only case a
one-GPU box can run; the multi-rank data flow is covered by tests/test_sharded_prover.py with the ranks in lockstep on
one GPU)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Solo:  # N = 1: the collectives are the identity
    def sum_host(self, a):
        return a

    def all_gather_device(self, ptr, total_words, words_per_rank):
        pass

    def all_to_all_device(self, send_ptr, recv_ptr, words_per_pair):
        raise RuntimeError("the row exchange form needs more than one rank (or --rehearse)")


class _Rehearsal(_Solo):
    """rank R of W alone on one GPU: the collectives are the identity except the first all-gather (the witness values), which is
    filled from the whole witness - a rank checks the witness it proves, so the values must be the real ones.  The time spent in
    here is not the rank's compute."""

    def __init__(self, m, ctx, d_wires, num_wires, n, world, rank=0, row_exchange=False, num_challenges=2, chunked=False):
        self.m, self.ctx, self.d_wires, self.n, self.calls, self.seconds = m, ctx, d_wires, n, 0, 0.0
        self.shards = m.parallel.column_shards(num_wires, world)
        self.chunk_cols = [m.parallel.chunk_columns(num_wires, r, world) for r in range(world)] if chunked else None
        self.world, self.rank, self.row_exchange, self.ch = world, rank, row_exchange, num_challenges

    def sum_host(self, a):
        # row exchange form: the table of block products must multiply to 1 (lcp2_perm_zs_rows_finish checks the copy constraints
        # with it); the other ranks' entries are made up so that it does
        if self.row_exchange and a.size == self.world * self.ch and a.ndim == 1 and self.world > 1:
            P = self.m.GOLDILOCKS_P
            out = a.copy()
            other = (self.rank + 1) % self.world
            for r in range(self.world):
                for k in range(self.ch):
                    if r != self.rank:
                        out[r * self.ch + k] = pow(int(a[self.rank * self.ch + k]), P - 2, P) if r == other else 1
            return out
        return a

    def all_to_all_device(self, send_ptr, recv_ptr, words_per_pair):
        """this rank's rows of EVERY column, out of the whole witness (the rank checks the gates on them)"""
        import torch
        torch.cuda.synchronize()
        t0 = time.time()
        rows = self.n // self.world
        if self.chunk_cols is not None:  # the chunked form: rank r's columns are chunk_columns(.., r, ..), in that order
            for r, cols in enumerate(self.chunk_cols):
                for i, col in enumerate(cols):
                    self.ctx.buffer_copy(recv_ptr + 8 * (r * words_per_pair + i * rows), self.d_wires.data_ptr() + 8 * (col * self.n + self.rank * rows), rows)
        else:
            for r, (s, e) in enumerate(self.shards):
                self.ctx.buffer_copy_2d(recv_ptr + 8 * r * words_per_pair, rows, self.d_wires.data_ptr() + 8 * (s * self.n + self.rank * rows), self.n, rows, e - s)
        torch.cuda.synchronize()
        self.seconds += time.time() - t0

    def all_gather_device(self, ptr, total_words, words_per_rank):
        import torch
        torch.cuda.synchronize()
        t0 = time.time()
        if self.calls == 0 and not self.row_exchange:
            for r, (s, e) in enumerate(self.shards):
                self.ctx.buffer_copy(ptr + 8 * r * words_per_rank, self.d_wires.data_ptr() + 8 * s * self.n, (e - s) * self.n)
        self.calls += 1
        torch.cuda.synchronize()
        self.seconds += time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree-bits", type=int, default=22)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sharded-columns", action="store_true", help="the witness arrives column-sharded (all-gather of values and coefficients)")
    ap.add_argument("--row-exchange", action="store_true", help="with --sharded-columns: the values cross the ranks as row blocks "
                    "(all-to-all), K5 and the gate check run on a rank's own rows (include/lcp2.h, the row exchange form)")
    ap.add_argument("--chunked", action="store_true", help="with --row-exchange: the coefficient exchange in chunks of 8 columns, overlapped with "
                    "the commitment (lcp2_commit_wires_chunk); the ranks bring the columns parallel.chunk_columns assigns to them")
    ap.add_argument("--rehearse", default="", help="R/W: time the compute of rank R of W on this one GPU (collectives replaced by "
                    "the identity, so the transcript is not the real one and the proof is not verified)")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import eth_lc_plonky2_amd as m
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    ctx = m.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    params = m.standard_params(a.degree_bits, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=4242)
    comm = m.parallel.TorchComm(dist, torch.device("cuda", local), ctx) if world > 1 else _Solo()
    rehearse = None
    if a.rehearse:
        rehearse = tuple(int(x) for x in a.rehearse.split("/"))
        rank, world = rehearse
    t0 = time.time()
    prover = m.parallel.ShardedProver(ctx, circ, rank, world, comm)
    prover.finish_build()
    build_s = time.time() - t0
    d_wires = torch.from_numpy(wires.view("int64")).cuda()  # witness resident in HBM, as in bench.py
    n = 1 << a.degree_bits
    if rehearse and a.sharded_columns:
        prover.comm = comm = _Rehearsal(m, ctx, d_wires, params.num_wires, n, world, rank, a.row_exchange, params.num_challenges, a.chunked)
    first, end = prover.column_shard()
    mine = d_wires[first:end].contiguous() if a.sharded_columns else d_wires
    if a.chunked:
        mine = d_wires[torch.tensor(m.parallel.chunk_columns(params.num_wires, rank, world), device=d_wires.device)].contiguous()
    times = []
    ctx.prof_enable(True)
    for _ in range(a.reps):
        if world > 1 and not rehearse:
            dist.barrier()
        if isinstance(comm, _Rehearsal):
            comm.calls, comm.seconds = 0, 0.0
        ctx.prof_reset()
        torch.cuda.synchronize()
        t0 = time.time()
        proof = prover.prove(mine.data_ptr(), pis, mem=m.MEM_DEVICE, sharded_columns=a.sharded_columns, row_exchange=a.row_exchange, chunked=a.chunked)
        torch.cuda.synchronize()
        times.append(time.time() - t0 - (comm.seconds if isinstance(comm, _Rehearsal) else 0.0))
    fam = ", ".join("%s %.1f" % (k, v["ms"]) for k, v in ctx.prof_get().items() if v["ms"] > 0.05)
    if rehearse:
        print("rehearsal of rank %d of %d, degree_bits %d%s: build %.2f s, per-rank compute of a sharded proof %s ms (exchanges excluded); "
              "kernel families of the last proof (ms): %s"
              % (rank, world, a.degree_bits, (", column-sharded witness" + (", row exchange" if a.row_exchange else "") + (", chunked coefficient exchange" if a.chunked else "")) if a.sharded_columns else "", build_s,
                 ", ".join("%.1f" % (1e3 * t) for t in times), fam))
        return
    if rank == 0:
        digest, _ = prover.data.digest()
        vd = m.CircuitData.verifier_only(circ, digest, prover.data.digest()[1])
        vd.verify(proof, pis)
        print("world %d degree_bits %d: build %.2f s, sharded proof %s ms (verified)" % (world, a.degree_bits, build_s,
              ", ".join("%.1f" % (1e3 * t) for t in times)))
    if world > 1 and not rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
