#!/usr/bin/env python3
"""Small workload for rocprofv3 --pmc passes: PolynomialBatch::from_values of `cols` columns at 2^bits rows, `reps` times
(the same kernels as the proof's commitments, short enough for counter collection)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import eth_lc_plonky2_amd as m

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 22
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
torch.cuda.set_device(0)
ctx = m.Context(0, stream=torch.cuda.current_stream().cuda_stream)
t = torch.randint(0, 2 ** 32, (cols, 1 << bits), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for _ in range(reps):
    o = ctx.commit_values(t.data_ptr(), mem=m.MEM_DEVICE, shape=(cols, 1 << bits))
    o.close()
print("done")
