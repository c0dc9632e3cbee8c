"""Batch proving across ranks (BASELINE configs[4]: a batch of consecutive light-client updates,
data-parallel over the GPUs of one node).  The updates are independent proofs, so the path shards by
update with NO data-path collective; the only communication is gathering the finished proofs (a few
hundred KB each) on rank 0 and the max-reduce of the elapsed time that the benchmark contract asks for.

One process per GPU under torch.distributed (backend "nccl" = RCCL on the MI355X node, "gloo" in the CPU tests).
"""
import numpy as np


def assign_updates(num_updates, world_size):
    """Contiguous, balanced blocks: rank r proves updates [start_r, end_r).  32 updates over 8 ranks -> 4 each."""
    base, extra = divmod(num_updates, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append(range(start, start + n))
        start += n
    return out


def prove_batch(prove_fn, witnesses, rank=0, world_size=1, dist=None, device=None):
    """prove_fn(witness) -> proof (numpy uint64 array, fixed length).  Every rank proves its block of `witnesses`
    (a sequence indexed by update number, or a callable update -> witness).  Returns the list of all proofs in
    update order on rank 0 and None elsewhere."""
    mine = assign_updates(len(witnesses) if not callable(witnesses) else witnesses.num_updates, world_size)[rank]
    get = witnesses if callable(witnesses) else witnesses.__getitem__
    local = [np.ascontiguousarray(prove_fn(get(u)), dtype=np.uint64) for u in mine]
    if world_size == 1 or dist is None:
        return local
    import torch
    words = local[0].size if local else 0
    wt = torch.tensor([words], dtype=torch.int64, device=device)
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    words = int(wt.item())
    blocks = assign_updates(len(witnesses) if not callable(witnesses) else witnesses.num_updates, world_size)
    most = max(len(b) for b in blocks)
    buf = torch.zeros((most, words), dtype=torch.int64, device=device)
    for i, p in enumerate(local):
        buf[i] = torch.from_numpy(p.view(np.int64)).to(buf.device)
    gathered = [torch.zeros_like(buf) for _ in range(world_size)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if rank != 0:
        return None
    proofs = []
    for r, blk in enumerate(blocks):
        g = gathered[r].cpu().numpy().view(np.uint64)
        proofs += [g[i].copy() for i in range(len(blk))]
    return proofs


def max_over_ranks(seconds, dist=None, device=None):
    """elapsed time of the slowest rank (the contract's timing rule)"""
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
