#!/usr/bin/env python3
"""bench.py's cpu_baseline scales a 2^18-row oracle proof to 2^22 rows with a model (transforms as n log n, the rest linearly).
This measures the oracle at 2^18 AND at 2^20 (about 2.5 minutes on the box's 32 threads: too long for the default bench run), applies
the model to the 2^18 sample and writes both next to each other: profiles/r04_cpu_baseline_scaling.json.  The GPU proves both samples
too and the proofs are compared word for word (`gpu_proof_equal`): whole-proof parity at 2^20 rows, which the default bench run stops
short of (2^18).
    python3 tests/checks/cpu_baseline_scaling.py [big_bits=20] > gpurun_out/cpu_scaling.json"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import eth_lc_plonky2_amd as m  # noqa: E402
import oracle_lib  # noqa: E402

big = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = oracle_lib.load()
omp = ctypes.CDLL("libgomp.so.1")
threads = min(32, omp.omp_get_max_threads())
omp.omp_set_num_threads(threads)
out = {"threads": threads, "samples": {}}
ctx = m.Context(0)
for bits in (18, big):
    params = m.standard_params(bits, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=1, small_values=True)
    oc = oracle_lib.OracleCircuit(L, circ)
    t0 = time.perf_counter()
    proof = oc.prove(wires, pis)
    dt = time.perf_counter() - t0
    assert oc.verify(proof, pis) == 0
    oc.close()
    data = m.CircuitData.build(ctx, circ)
    t0 = time.perf_counter()
    gpu_proof = data.prove(wires, pis)
    gpu_dt = time.perf_counter() - t0
    equal = bool((gpu_proof == proof).all())
    data.close()
    del gpu_proof, proof
    n = 1 << bits
    cols = np.ascontiguousarray(wires % np.uint64(m.GOLDILOCKS_P))
    lde = np.zeros((cols.shape[0], n << params.rate_bits), dtype=np.uint64)
    t0 = time.perf_counter()
    L.orc_lde_batch(oracle_lib.vp(cols), cols.shape[0], n, params.rate_bits, 7, oracle_lib.vp(lde))
    t_ntt = min(1.4 * (time.perf_counter() - t0), 0.9 * dt)
    del lde, cols, wires, circ
    out["samples"][str(bits)] = {"prove_s": dt, "transforms_s": t_ntt, "gpu_proof_equal": equal, "gpu_prove_from_host_witness_s": gpu_dt}
    print("2^%d rows: %.2f s (transforms ~%.2f s); GPU proof %s" % (bits, dt, t_ntt, "EQUAL" if equal else "DIFFERS"), file=sys.stderr, flush=True)
    assert equal, "GPU proof differs from the oracle proof at 2^%d rows" % bits
s = out["samples"]["18"]
rows = float(1 << (big - 18))
model = rows * ((s["prove_s"] - s["transforms_s"]) + s["transforms_s"] * (big + 3) / (18 + 3))
out["model_from_2p18_for_2p%d_s" % big] = model
out["measured_2p%d_s" % big] = out["samples"][str(big)]["prove_s"]
out["model_over_measured"] = model / out["samples"][str(big)]["prove_s"]
rows22 = float(1 << (22 - big))
sb = out["samples"][str(big)]
out["extrapolated_2p22_from_2p%d_s" % big] = rows22 * ((sb["prove_s"] - sb["transforms_s"]) + sb["transforms_s"] * 25 / (big + 3))
out["extrapolated_2p22_from_2p18_s"] = 16.0 * ((s["prove_s"] - s["transforms_s"]) + s["transforms_s"] * 25 / 21)
print(json.dumps(out))
