#!/usr/bin/env python3
"""Do two kernel families of a proof gain anything from running CONCURRENTLY on one MI355X?  Two library contexts (a HIP stream each) on
the same device, one host thread per context (the C ABI's device-to-device calls synchronise their own stream; ctypes releases the GIL):
    A  lcp2_lde_batch            coset LDE of NCOLS columns of 2^LG coefficients (rate 8): VALU-bound with ~35 % of its cycles waiting
       lcp2_ntt_batch (inverse)  the iNTT of the same columns: its three passes are memory-bound
    B  lcp2_poseidon_permute_batch  2^PERMS_LG permutations, one per lane: integer-VALU bound like the leaf hashing (K4a)
Each pair is timed back to back on one thread and then side by side; `gain` = (sequential - concurrent) / sequential.  A measurement for
DESIGN.md section 6 ("left"): whether a commitment that extends chunk j + 1 while chunk j is absorbed could hide LDE time under hashing.
    python3 tools/overlap_probe.py [LG=22] [NCOLS=32] [PERMS_LG=25] [REPS=4]"""
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eth_lc_plonky2_amd as m  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 32
perms_lg = int(sys.argv[3]) if len(sys.argv) > 3 else 25
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
n, perms = 1 << lg, 1 << perms_lg
A, B = m.Context(0), m.Context(0)
assert A.stream_ptr() != B.stream_ptr(), "the two contexts share a stream"
rng = np.random.default_rng(1)
src, dst = A.buffer_alloc(ncols * n), A.buffer_alloc(ncols * n * 8)
col = rng.integers(0, m.GOLDILOCKS_P, size=n, dtype=np.uint64)
for c in range(ncols):
    A.buffer_write(src + 8 * c * n, col + np.uint64(c))
sin, sout = B.buffer_alloc(12 * perms), B.buffer_alloc(12 * perms)
block = rng.integers(0, m.GOLDILOCKS_P, size=12 << 20, dtype=np.uint64)
for off in range(0, 12 * perms, block.size):
    B.buffer_write(sin + 8 * off, block[:min(block.size, 12 * perms - off)])
V = ctypes.c_void_p
calls = {
    "lde": lambda: A._check(A.lib.lcp2_lde_batch(A.handle, V(src), V(dst), ncols, lg, 3, m.MEM_DEVICE)),
    "intt": lambda: A._check(A.lib.lcp2_ntt_batch(A.handle, V(src), ncols, lg, 1, 1, m.MEM_DEVICE)),
    "poseidon": lambda: B._check(B.lib.lcp2_poseidon_permute_batch(B.handle, V(sin), V(sout), perms, m.MEM_DEVICE)),
}


def run(name, count):
    for _ in range(count):
        calls[name]()


def wall(jobs):
    """jobs: [(name, count)], one thread each; wall time until all are done"""
    threads = [threading.Thread(target=run, args=j) for j in jobs]
    A.sync(), B.sync()
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    A.sync(), B.sync()
    return time.perf_counter() - t0


out = {"log_n": lg, "columns": ncols, "permutations": perms, "reps": reps, "pairs": {}}
for name in calls:
    run(name, 1)  # warm-up (tables, code objects)
alone = {name: min(wall([(name, reps)]) for _ in range(3)) / reps for name in calls}
out["alone_ms"] = {k: round(1e3 * v, 3) for k, v in alone.items()}
for a in ("lde", "intt"):
    # the same total time on both sides, so that the two streams overlap from start to end
    ra = max(1, round(reps * alone["poseidon"] / alone[a]))
    seq = ra * alone[a] + reps * alone["poseidon"]
    conc = min(wall([(a, ra), ("poseidon", reps)]) for _ in range(3))
    out["pairs"][a + "+poseidon"] = {"calls": [ra, reps], "sequential_ms": round(1e3 * seq, 3), "concurrent_ms": round(1e3 * conc, 3),
                                     "gain": round((seq - conc) / seq, 4)}
print(json.dumps(out))
