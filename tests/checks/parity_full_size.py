#!/usr/bin/env python3
"""Whole-proof parity at the headline size: the CPU oracle (oracle/, 32 OpenMP threads, ~10 minutes of prove() at 2^22 rows plus its
build()) and the GPU prove the same 2^22-row circuit of plonky2's gate set from the same witness, and the two proofs are compared word
for word.  Too long for the test-suite and for bench.py (whose cpu_baseline leg does the same at 2^18); run once per round on the GPU box:
    python3 tests/checks/parity_full_size.py [bits=22] [plonky2|reference-mix] > gpurun_out/parity_full_size.json
`reference-mix`: the reference's own gate set (u32_gates.reference_mix_circuit: plonky2_u32 / comparison gates on the GENERATED native
evaluators of the device, interpreted gate programs in the oracle).
A heartbeat line goes to stderr every minute (the box takes a silent command for hung)."""
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import eth_lc_plonky2_amd as m  # noqa: E402
import oracle_lib  # noqa: E402

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 22
workload = sys.argv[2] if len(sys.argv) > 2 else "plonky2"
phase = {"name": "start", "t0": time.perf_counter()}


def heartbeat():
    while True:
        time.sleep(60)
        print("[%6.0f s] %s" % (time.perf_counter() - phase["t0"], phase["name"]), file=sys.stderr, flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
L = oracle_lib.load()
omp = ctypes.CDLL("libgomp.so.1")
threads = min(32, omp.omp_get_max_threads())
omp.omp_set_num_threads(threads)
out = {"degree_bits": bits, "threads": threads, "workload": workload}
phase["name"] = "circuit description"
t0 = time.perf_counter()
if workload == "reference-mix":
    params = m.standard_params(bits, 5)
    circ, wires, pis = m.u32_gates.reference_mix_circuit(params, seed=1, native=True, small_values=True)
else:
    params = m.standard_params(bits, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=1, small_values=True)
out["description_s"] = time.perf_counter() - t0
# the GPU first (seconds): if the oracle run is cut short the GPU side is still on record
phase["name"] = "GPU build + prove"
ctx = m.Context(0)
data = m.CircuitData.build(ctx, circ)
t0 = time.perf_counter()
gpu_proof = data.prove(wires, pis)
out["gpu_prove_from_host_witness_s"] = time.perf_counter() - t0
data.close()
phase["name"] = "oracle build (constants and sigmas commitment)"
t0 = time.perf_counter()
oc = oracle_lib.OracleCircuit(L, circ)
out["oracle_build_s"] = time.perf_counter() - t0
print("oracle build %.1f s" % out["oracle_build_s"], file=sys.stderr, flush=True)
phase["name"] = "oracle prove"
t0 = time.perf_counter()
proof = oc.prove(wires, pis)
out["oracle_prove_s"] = time.perf_counter() - t0
print("oracle prove %.1f s" % out["oracle_prove_s"], file=sys.stderr, flush=True)
phase["name"] = "compare + verify"
equal = bool((gpu_proof == proof).all())
out["proof_words"] = int(proof.size)
out["gpu_proof_equal"] = equal
out["differing_words"] = int((gpu_proof != proof).sum())
out["oracle_verifies_gpu_proof"] = oc.verify(gpu_proof, pis) == 0
oc.close()
print(json.dumps(out))
sys.exit(0 if equal else 1)
