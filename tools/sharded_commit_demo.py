#!/usr/bin/env python3
"""Coset-sharded commitment of a light-client-sized wire matrix over the GPUs of one node (BASELINE configs[3]).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/sharded_commit_demo.py [--degree-bits 22] [--cols 135]

Each rank owns a column shard, inverts it, all-gathers the coefficients over RCCL and commits its leaf blocks
(eth-lc-plonky2_amd/parallel.py).  Prints the per-rank time split and checks that every rank assembled the same cap.
With N = 1 it degenerates to the single-GPU commitment (the only case that can be run on a one-GPU box)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree-bits", type=int, default=22)
    ap.add_argument("--cols", type=int, default=135)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import eth_lc_plonky2_amd as m
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    ctx = m.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    n = 1 << a.degree_bits
    s, e = m.parallel.column_shards(a.cols, world)[rank]
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
    shard0 = torch.randint(0, 2 ** 32, (e - s, n), generator=g, device="cuda", dtype=torch.int64)
    ops = m.parallel.GpuOps(ctx, n)
    for rep in range(a.reps):
        shard = shard0.clone()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        com = m.parallel.sharded_commit(shard, a.cols, a.degree_bits, rank, world, ops, dist=dist if world > 1 else None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        com.local.close()
        dt = m.batch.max_over_ranks(dt, dist if world > 1 else None, device="cuda")
        if rank == 0:
            print(f"rep {rep}: sharded commit of {a.cols} x 2^{a.degree_bits} over {world} GPU(s): {dt * 1e3:.1f} ms (slowest rank)")
    cap = torch.from_numpy(com.cap.view(np.int64).copy()).cuda()
    if world > 1:
        caps = [torch.zeros_like(cap) for _ in range(world)]
        dist.all_gather(caps, cap)
        assert all(bool((c == cap).all()) for c in caps), "ranks disagree on the cap"
        dist.destroy_process_group()
    if rank == 0:
        print("cap[0] =", [hex(int(x)) for x in com.cap[0]])


if __name__ == "__main__":
    main()
