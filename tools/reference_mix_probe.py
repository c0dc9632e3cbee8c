#!/usr/bin/env python3
"""The reference's gate set as a device workload: `reps` proofs of u32_gates.reference_mix_circuit (U32AddMany / U32Arithmetic /
U32RangeCheck / U32Subtraction / Comparison rows next to BaseSum, Arithmetic, Constant, PublicInput and PoseidonGate rows, the public
inputs hashed in-circuit) at 2^bits rows, witness resident in HBM; prints ONE JSON line with the per-kernel-family times of the last
`reps` proofs.  bench.py imports measure() for config.reference_gate_set_2p22; under rocprofv3 the same script gives the kernel
stats / FETCH_SIZE passes of profiles/r04_reference_mix_*.
    python3 tools/reference_mix_probe.py 22 3 [--interpreted]
(the comparison with the oracle's proof of this circuit, at 2^20 and 2^22 rows, is tests/checks/parity_full_size.py N reference-mix)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def generated_gate_registers():
    """VGPRs of the generated gate kernels as compiled (code-object metadata of liblcp2.so via tools/kernel_regs.py), by gate name"""
    import re
    import subprocess
    import eth_lc_plonky2_amd as m
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_regs.py"), "k_q_gen"], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return None
    regs = {}
    for k, check, v, spill in re.findall(r"k_q_gen<(\d+)u, (false|true)>\s+vgpr\s+(\d+) .* spill v(\d+)", out):
        if check == "false":
            regs[m.circuit.GENERATED_GATE_NAMES[int(k)]] = {"vgprs": int(v), "spilled": int(spill)}
    return regs or None


def measure(ctx, degree_bits=22, reps=3, native=True, seed=3):
    import numpy as np
    import torch
    import eth_lc_plonky2_amd as m
    from eth_lc_plonky2_amd import u32_gates as ug
    params = m.standard_params(degree_bits, 5)
    t0 = time.perf_counter()
    circ, wires, pis = ug.reference_mix_circuit(params, seed=seed, native=native, small_values=True)
    t_desc = time.perf_counter() - t0
    gs = circ.gateset
    counts = {name: int((circ.constants_sigmas[gs.gates[k].selector_index] == np.uint64(k)).sum()) for k, name in enumerate(gs.names)}
    t0 = time.perf_counter()
    data = m.CircuitData.build(ctx, circ)
    t_build = time.perf_counter() - t0
    w = torch.from_numpy(wires.view(np.int64)).cuda()
    torch.cuda.synchronize()
    data.prove(w.data_ptr(), pis, mem=m.MEM_DEVICE)  # warm-up
    was = ctx.prof_enable(True) if hasattr(ctx, "prof_enable") else None
    p0 = ctx.prof_get()
    times = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        proof = data.prove(w.data_ptr(), pis, mem=m.MEM_DEVICE)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    p1 = ctx.prof_get()
    data.verify(proof, pis)
    out = {"workload": "the reference's gate set at 2^%d rows: %s; real copy constraints, public inputs hashed in-circuit, witness resident in HBM, proof verified"
                       % (degree_bits, ", ".join("%s x%d" % (k, v) for k, v in counts.items() if v)),
           "degree_bits": degree_bits, "gate_evaluators": "generated straight-line (LCP2_GATE_NATIVE_GENERATED) + native plonky2 gates" if native else "interpreted",
           "ms_per_proof": 1e3 * min(times), "ms_per_proof_mean": 1e3 * sum(times) / len(times),
           "kernel_ms_per_proof": {k: round((p1[k]["ms"] - p0[k]["ms"]) / reps, 3) for k in p1 if p1[k]["launches"] - p0[k]["launches"]},
           "quotient_ms": round((p1["quotient"]["ms"] - p0["quotient"]["ms"]) / reps, 3),
           "description_s": round(t_desc, 1), "build_s": round(t_build, 1)}
    data.close()
    del w
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import torch
    import eth_lc_plonky2_amd as m
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    bits = int(args[0]) if args else 22
    reps = int(args[1]) if len(args) > 1 else 3
    torch.cuda.set_device(0)
    ctx = m.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    out = measure(ctx, bits, reps, native="--interpreted" not in sys.argv)
    if "--interpreted" not in sys.argv and "--no-regs" not in sys.argv:  # (no child process under rocprofv3)
        out["generated_gate_kernels"] = generated_gate_registers()
    print(json.dumps(out))
