#!/usr/bin/env python3
"""One-off robustness run (GPU): many seeds and sizes of GPU-vs-oracle parity beyond what the test-suite pins down:
whole proofs (synthetic circuits, random and small-valued witnesses), 2^20 random Poseidon permutations, LDE of random columns.
    python tests/checks/parity_soak.py [rounds]"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("OMP_NUM_THREADS", "16")  # the oracle's OpenMP regions: a GPU box shows more hardware threads than the job may use

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import eth_lc_plonky2_amd as m  # noqa: E402
import oracle_lib  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = oracle_lib.load()
ctx = m.Context(0)
t0 = time.time()
bad = 0
for r in range(rounds):
    db = 5 + r % 8
    params = m.standard_params(db, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=7000 + r, small_values=bool(r & 1))
    oc = oracle_lib.OracleCircuit(L, circ)
    want = oc.prove(wires, pis)
    data = m.CircuitData.build(ctx, circ)
    for rep in range(2):  # the second proof reuses the workspace
        got = data.prove(wires, pis)
        if not (got == want).all():
            bad += 1
            print("MISMATCH proof", r, db, rep, int(np.nonzero(got != want)[0][0]))
    data.close()
    oc.close()
    print("round %d: degree_bits %d ok (%.1f s so far)" % (r, db, time.time() - t0), flush=True)
print("proofs: %d rounds, %d mismatches, %.1f s" % (rounds, bad, time.time() - t0))
rng = np.random.default_rng(99)
P = m.GOLDILOCKS_P
for rep in range(4):
    n = 1 << 18
    st = rng.integers(0, 1 << 63, size=(n, 12), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 12), dtype=np.uint64)  # any u64, incl. >= p
    got = ctx.poseidon_permute_batch(st)
    want = np.ascontiguousarray(st % np.uint64(P))
    L.orc_poseidon_permute_batch(oracle_lib.vp(want), oracle_lib.vp(want), n)
    if not (got == want).all():
        bad += 1
        print("MISMATCH poseidon batch", rep)
print("poseidon: 4 x 2^18 permutations of arbitrary u64 states compared", flush=True)
for lg in (10, 13, 14, 16, 18, 20):
    cols = rng.integers(0, P, size=(3, 1 << lg), dtype=np.uint64)
    got = ctx.lde_batch(cols, 3)
    want = oracle_lib.lde_leaf_order(L, cols, 3, 7)
    if not (np.asarray(got) == np.asarray(want)).all():
        bad += 1
        print("MISMATCH lde", lg)
    print("lde 2^%d ok" % lg, flush=True)
print("lde: sizes 2^10..2^20 compared")
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
