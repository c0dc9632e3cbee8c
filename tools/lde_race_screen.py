#!/usr/bin/env python3
"""Race screen for the prefetching NTT kernel (raw s_barrier + lgkmcnt-only waits, loads in flight across barriers): the same LDE
(2^LG coefficients x NCOLS columns, rate 8) REPS times; every output must have the checksums of the first one.  A data race shows
as a run-to-run difference long before a parity test catches it.
    python3 tools/lde_race_screen.py [LG=22] [NCOLS=16] [REPS=40]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eth_lc_plonky2_amd as m  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
torch.cuda.set_device(0)
ctx = m.Context(0, stream=torch.cuda.current_stream().cuda_stream)
n = 1 << lg
rng = np.random.default_rng(7)
src = torch.from_numpy(rng.integers(0, m.GOLDILOCKS_P, size=(ncols, n), dtype=np.uint64).view(np.int64)).cuda()
dst = torch.empty((ncols, n * 8), dtype=torch.int64, device="cuda")
weights = torch.arange(1, 2 * n * 8, 2, dtype=torch.int64, device="cuda")  # odd multipliers: a permuted or shifted output changes the sum


def run():
    dst.zero_()
    torch.cuda.synchronize()  # torch's default stream is stream 0: the library then runs on a stream of its own
    ctx._check(ctx.lib.lcp2_lde_batch(ctx.handle, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), ncols, lg, 3, m.MEM_DEVICE))
    torch.cuda.synchronize()
    return [int((dst[c] * weights).sum().item()) for c in range(ncols)] + [int(dst.sum().item())]


first = run()
bad = 0
for r in range(1, reps):
    got = run()
    if got != first:
        bad += 1
        diff = [c for c in range(len(first)) if got[c] != first[c]]
        print("run %d differs from run 0 in checksums %s" % (r, diff[:8]), flush=True)
print("lde race screen: 2^%d x %d columns, %d runs, %d differing" % (lg, ncols, reps, bad))
sys.exit(1 if bad else 0)
