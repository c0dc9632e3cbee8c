#!/usr/bin/env python3
"""LDE of NCOLS columns of 2^LG coefficients (rate 8) through the C ABI, a few times: run under
`rocprofv3 --kernel-trace --stats` to get the duration of each NTT pass (strided pass first, contiguous pass second)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eth_lc_plonky2_amd as m  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = m.Context(0)
rng = np.random.default_rng(1)
coeffs = rng.integers(0, m.GOLDILOCKS_P, size=(ncols, 1 << lg), dtype=np.uint64)
o = ctx.commit_coeffs(coeffs, rate_bits=3, cap_height=4)
ctx.sync()
ctx.prof_enable(True)
for _ in range(3):
    o2 = ctx.commit_coeffs(coeffs, rate_bits=3, cap_height=4)
    o2.close()
ctx.sync()
print(ctx.prof_get())
