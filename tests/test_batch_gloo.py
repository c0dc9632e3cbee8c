"""The N > 1 path on CPU: two gloo ranks shard a batch of light-client updates (independent proofs, no data-path
collective), rank 0 gathers every proof in update order.  The per-rank prover is the oracle here (no GPU in this
container); on the MI355X node the same code runs with backend nccl (= RCCL) and the HIP prover (bench.py)."""
import os
import socket

import numpy as np
import pytest


def test_assign_updates_covers_every_update_once():
    import eth_lc_plonky2_amd as m
    for n in (0, 1, 5, 32, 33):
        for w in (1, 2, 3, 8):
            blocks = m.batch.assign_updates(n, w)
            assert len(blocks) == w
            flat = [u for b in blocks for u in b]
            assert flat == list(range(n))
            assert max(len(b) for b in blocks) - min(len(b) for b in blocks) <= 1
    assert [len(b) for b in m.batch.assign_updates(32, 8)] == [4] * 8  # BASELINE configs[4]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_updates, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import torch.distributed as dist
    import eth_lc_plonky2_amd as m
    import oracle_lib
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = oracle_lib.load()
    params = m.standard_params(5, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=11)
    oc = oracle_lib.OracleCircuit(L, circ)

    def witness(u):  # update u: the same circuit, a different witness (free cells of the padding row tagged with u)
        w = wires.copy()
        m.circuit.tag_witness(w, u + 1)
        return w

    witness.num_updates = num_updates
    proofs = m.batch.prove_batch(lambda w: oc.prove(w, pis), witness, rank, world, dist=dist)
    t = m.batch.max_over_ranks(float(rank + 1), dist=dist)
    assert t == float(world)
    if rank == 0:
        assert len(proofs) == num_updates
        for u, p in enumerate(proofs):
            assert oc.verify(p, pis) == 0
            assert (p == oc.prove(witness(u), pis)).all()  # update order preserved
        np.save(os.path.join(out_dir, "ok.npy"), np.array([len(proofs)]))
    else:
        assert proofs is None
    dist.destroy_process_group()


@pytest.mark.parametrize("num_updates", [3, 4])
def test_two_rank_batch_over_gloo(tmp_path, num_updates):
    import torch.multiprocessing as mp
    import oracle_lib
    oracle_lib.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, num_updates, str(tmp_path)), nprocs=2, join=True)
    assert int(np.load(tmp_path / "ok.npy")[0]) == num_updates


def test_four_rank_batch_with_an_idle_rank_over_gloo(tmp_path):
    """6 updates over 4 ranks (2, 2, 1, 1: padded gather blocks) and 3 updates over 4 ranks (the last rank proves nothing and still takes
    part in the collectives)"""
    import torch.multiprocessing as mp
    import oracle_lib
    oracle_lib.build()
    for num_updates in (6, 3):
        mp.spawn(_worker, args=(4, _free_port(), num_updates, str(tmp_path)), nprocs=4, join=True)
        assert int(np.load(tmp_path / "ok.npy")[0]) == num_updates
