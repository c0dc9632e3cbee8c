"""Coset-sharded commitment: the sharding arithmetic and the exchange orchestration.
CPU: two gloo ranks with the oracle as the per-rank compute; GPU (1 device): the ranks of a 1/2/4/8-way split are run one
after the other through lcp2_commit_cosets and the assembled cap must equal the single-GPU commitment."""
import os
import socket

import numpy as np
import pytest

import oracle_lib
from oracle_lib import P, commit_reference, merkle_cap, merkle_verify, rand_field, vp


def test_shard_arithmetic():
    import eth_lc_plonky2_amd as m
    assert m.parallel.column_shards(135, 8) == [(0, 17), (17, 34), (34, 51), (51, 68), (68, 85), (85, 102), (102, 119), (119, 135)]
    assert [m.parallel.block_range(r, 4) for r in range(4)] == [(0, 2), (2, 2), (4, 2), (6, 2)]
    assert m.parallel.block_range(5, 8) == (5, 1)
    with pytest.raises(ValueError):
        m.parallel.block_range(0, 3)


class OracleOps:
    """per-rank compute on the CPU oracle (test stand-in for the HIP kernels)"""

    def __init__(self, L, n, rate_bits=3, cap_height=4):
        self.L, self.n, self.rate_bits, self.cap_height = L, n, rate_bits, cap_height

    def intt(self, shard):
        c = np.ascontiguousarray(shard, dtype=np.uint64) % np.uint64(P)
        if c.shape[0]:
            self.L.orc_ifft_batch(vp(c), c.shape[0], self.n)
        return c

    def gather_columns(self, shard, shards, rank, world, dist):
        if world == 1:
            return shard
        import torch
        most = max(e - s for s, e in shards)
        pad = torch.zeros((most, self.n), dtype=torch.int64)
        pad[:shard.shape[0]] = torch.from_numpy(shard.view(np.int64))
        bufs = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
        return np.concatenate([bufs[r][:e - s].numpy().view(np.uint64) for r, (s, e) in enumerate(shards)])

    def commit_blocks(self, coeffs, first, count):
        lde = oracle_lib.lde_leaf_order(self.L, coeffs, self.rate_bits)
        leaves = np.ascontiguousarray(lde[:, first * self.n:(first + count) * self.n].T)
        local_cap = self.cap_height - self.rate_bits + (count.bit_length() - 1)
        return leaves, merkle_cap(self.L, leaves, local_cap)

    def gather_caps(self, cap_part, rank, world, dist):
        if world == 1:
            return cap_part
        import torch
        t = torch.from_numpy(cap_part.view(np.int64).copy())
        bufs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(bufs, t)
        return np.concatenate([b.numpy().view(np.uint64) for b in bufs]).reshape(-1, 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import torch.distributed as dist
    import eth_lc_plonky2_amd as m
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = oracle_lib.load()
    lg, ncols = 6, 11
    n = 1 << lg
    values = rand_field(np.random.default_rng(5), (ncols, n))  # every rank derives the same trace, keeps only its columns
    s, e = m.parallel.column_shards(ncols, world)[rank]
    com = m.parallel.sharded_commit(values[s:e].copy(), ncols, lg, rank, world, OracleOps(L, n), dist=dist)
    want = commit_reference(L, values)[2]
    assert (com.cap == want).all()
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), com.cap)
    dist.destroy_process_group()


def test_sharded_commit_two_gloo_ranks(tmp_path):
    import torch.multiprocessing as mp
    oracle_lib.build()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (np.load(tmp_path / "ok0.npy") == np.load(tmp_path / "ok1.npy")).all()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("lg,ncols", [(10, 20), (14, 9)])
def test_commit_cosets_assembles_to_the_single_gpu_cap(gpu_ctx, oracle, world, lg, ncols):
    import eth_lc_plonky2_amd as m
    n = 1 << lg
    values = rand_field(np.random.default_rng(lg * 10 + world), (ncols, n), canonical=False)
    full = gpu_ctx.commit_values(values)
    coeffs, lde = full.read()
    # polynomial-parallel side: every "rank" inverts its own column shard
    shards = m.parallel.column_shards(ncols, world)
    gathered = np.concatenate([gpu_ctx.ntt_batch(values[s:e], inverse=True) if e > s else np.zeros((0, n), dtype=np.uint64) for s, e in shards])
    assert (gathered == coeffs).all()
    caps = []
    for rank in range(world):
        first, count = m.parallel.block_range(rank, world)
        o = gpu_ctx.commit_cosets(gathered, first, count)
        caps.append(o.cap)
        _, part = o.read(coeffs=False)
        assert (part == lde[:, first * n:(first + count) * n]).all()
        # a local opening verifies against the GLOBAL cap at the global leaf index
        local_idx = np.array([0, count * n - 1, (count * n) // 3], dtype=np.uint64)
        leaves, sib = o.open(local_idx)
        for q, li in enumerate(local_idx):
            assert merkle_verify(oracle, leaves[q], first * n + int(li), sib[q], full.cap)
        o.close()
    assert (np.concatenate(caps) == full.cap).all()
    full.close()
