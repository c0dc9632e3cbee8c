#!/usr/bin/env python3
"""LDE of NCOLS columns of 2^LG coefficients (rate 8) and the inverse transform of NCOLS value columns, device to device
through the C ABI (lcp2_lde_batch, lcp2_ntt_batch), REPS times each: the NTT kernels alone, for `rocprofv3 --pmc` runs.
    python3 tools/lde_probe.py [LG=22] [NCOLS=32] [REPS=3]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eth_lc_plonky2_amd as m  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = m.Context(0)
n = 1 << lg
rng = np.random.default_rng(1)
src = ctx.buffer_alloc(ncols * n)
dst = ctx.buffer_alloc(ncols * n * 8)
for c in range(ncols):
    ctx.buffer_write(src + 8 * c * n, rng.integers(0, m.GOLDILOCKS_P, size=n, dtype=np.uint64))
for name, call in (("lde", lambda: ctx.lib.lcp2_lde_batch(ctx.handle, ctypes.c_void_p(src), ctypes.c_void_p(dst), ncols, lg, 3, m.MEM_DEVICE)),
                   ("intt", lambda: ctx.lib.lcp2_ntt_batch(ctx.handle, ctypes.c_void_p(src), ncols, lg, 1, 1, m.MEM_DEVICE))):
    ctx._check(call())
    ctx.sync()
    t0 = time.time()
    for _ in range(reps):
        ctx._check(call())
    ctx.sync()
    ms = 1e3 * (time.time() - t0) / reps
    print("%s of %d columns of 2^%d: %.3f ms (%.1f ns per column element)" % (name, ncols, lg, ms, 1e6 * ms / (ncols * n)))
