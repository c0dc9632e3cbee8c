#!/usr/bin/env python3
"""Small whole-proof workload for rocprofv3 passes (kernel trace, --pmc): `reps` proofs of the synthetic circuit (plonky2's gate
set: Noop / Constant / PublicInput / BaseSum / Arithmetic / Poseidon) at 2^bits rows, witness resident in HBM.
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/prof_prove.py 20 2"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import eth_lc_plonky2_amd as m

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.cuda.set_device(0)
ctx = m.Context(0, stream=torch.cuda.current_stream().cuda_stream)
params = m.standard_params(bits, 4)
circ, wires, pis = m.circuit.synthetic_circuit(params, seed=3, small_values=True)
data = m.CircuitData.build(ctx, circ)
w = torch.from_numpy(wires.view(np.int64)).cuda()
torch.cuda.synchronize()
for _ in range(reps):
    proof = data.prove(w.data_ptr(), pis, mem=m.MEM_DEVICE)
data.verify(proof, pis)
print("done: %d proofs at 2^%d rows" % (reps, bits))
