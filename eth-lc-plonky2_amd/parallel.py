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
import numpy as np


def column_shards(ncols, world_size):
    base, extra = divmod(ncols, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


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
    """Rank `rank` of `world` in a coset-sharded proof.  Every rank holds the whole witness and runs the same sequence of
    seams on its own leaf blocks; results are shares that a SUM all-reduce assembles (caps, the proof array; uint64 wrap-around,
    the shares are disjoint) plus one bulk all-reduce of the quotient values.  `comm` supplies the two collectives:
        comm.sum_host(numpy uint64 array) -> numpy uint64 array      (tiny: 512-byte caps, the proof array)
        comm.sum_device(device pointer, uint64 words)                 (in place; 2 * 8n words: RCCL over xGMI)
    so that the same orchestration runs over torch.distributed (TorchComm) and, in the single-GPU tests, over ranks that are
    stepped in lockstep inside one process."""

    def __init__(self, ctx, circ, rank, world, comm, constants_sigmas_ptr=None, mem=0):
        from . import binding as b
        self.b, self.circ, self.rank, self.world, self.comm = b, circ, rank, world, comm
        first, count = block_range(rank, world, circ.params.rate_bits)
        self.data = b.CircuitData.build_sharded(ctx, circ, first, count, constants_sigmas_ptr, mem)
        self.cap_share = self.data.digest()[1]

    def finish_build(self):
        self.data.set_constants_cap(self.comm.sum_host(self.cap_share))
        self.digest = self.data.digest()[0]

    # the proof as a generator of exchange points, so that a test can interleave several ranks in one process
    def prove_steps(self, wires, public_inputs, mem=0):
        b, d, p = self.b, self.data, self.circ.params
        capw = 4 << p.cap_height
        proof = np.zeros(d.proof_words, dtype=np.uint64)
        pis = np.asarray(public_inputs, dtype=np.uint64)
        ch = b.Challenger()
        ch.observe(self.digest)
        pi_hash = b.hash_no_pad(pis)
        ch.observe(pi_hash)
        share = d.commit_wires(wires, mem)
        proof[0:capw] = (yield ("sum_host", share)).ravel()
        ch.observe(proof[0:capw])
        betas, gammas = ch.get(p.num_challenges), ch.get(p.num_challenges)
        share = d.perm_zs(betas, gammas)
        proof[capw:2 * capw] = (yield ("sum_host", share)).ravel()
        ch.observe(proof[capw:2 * capw])
        alphas = ch.get(p.num_challenges)
        d.quotient_values(alphas, pi_hash)
        yield ("sum_device",) + d.quotient_buffer()
        share = d.quotient_commit()
        proof[2 * capw:3 * capw] = (yield ("sum_host", share)).ravel()
        ch.observe(proof[2 * capw:3 * capw])
        zeta = ch.get(2)
        caps = proof[:3 * capw].copy()
        d.fri_open(zeta, ch.state, proof)  # writes the words after the caps: this rank's share of them
        proof[:3 * capw] = 0
        proof = (yield ("sum_host", proof))
        proof[:3 * capw] = caps
        self.proof = proof
        return

    def prove(self, wires, public_inputs, mem=0):
        steps = self.prove_steps(wires, public_inputs, mem)
        reply = None
        try:
            while True:
                req = steps.send(reply)
                if req[0] == "sum_host":
                    reply = self.comm.sum_host(req[1])
                else:
                    self.comm.sum_device(req[1], req[2])
                    reply = None
        except StopIteration:
            return self.proof


class TorchComm:
    """torch.distributed collectives for ShardedProver: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" on CPU."""

    def __init__(self, dist, device=None, ctx=None):
        self.dist, self.device, self.ctx = dist, device, ctx

    def sum_host(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy())
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)  # two's-complement wrap-around = uint64 addition
        return t.cpu().numpy().view(np.uint64).reshape(np.shape(arr))

    def sum_device(self, ptr, words):
        import torch
        t = torch.empty(words, dtype=torch.int64, device=self.device)  # staging tensor: two device copies (< 1 ms at 537 MB)
        self.ctx.buffer_copy(t.data_ptr(), ptr, words)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        self.ctx.buffer_copy(ptr, t.data_ptr(), words)
