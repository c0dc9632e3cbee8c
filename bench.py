#!/usr/bin/env python3
"""Headline benchmark: light-client proof throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the prover hot path over one synthetic witness trace of the
BASELINE workload (configs[2]: full light-client circuit shape, n = 2^22 rows,
135 wires, standard_recursion_config) with the trace already resident in HBM.
N > 1: one rank per GPU, every rank proves its own independent update (BASELINE
configs[4], "replicas": no data-path collective) -> weak scaling; the only
collective is the timing barrier / max-reduce the contract asks for.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (Poseidon
leaf hashing, K4a): algorithmic bytes / HIP-event time measured inside this run.
`cpu_baseline` is the oracle (oracle/, "port") timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--degree-bits", type=int, default=22)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-bits", type=int, default=15, help="log2 rows of the oracle's bounded sample")
    return ap.parse_args()


def synth_trace(torch, dev, ncols, n, seed):
    """SURVEY 8(d) config 2/3 trace: byte-valued and u32-valued columns mixed 50/50 (SHA-256 traces are small-valued)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    t = torch.empty((ncols, n), dtype=torch.int64, device=dev)
    half = ncols // 2
    t[:half] = torch.randint(0, 256, (half, n), generator=g, device=dev, dtype=torch.int64)
    t[half:] = torch.randint(0, 2 ** 32, (ncols - half, n), generator=g, device=dev, dtype=torch.int64)
    return t


def cpu_baseline(sample_bits, widths):
    """Oracle (CPU port) timed on a bounded sample: the same three commitments at 2^sample_bits rows."""
    import numpy as np
    import oracle_lib
    L = oracle_lib.load()
    rng = np.random.default_rng(0)
    n = 1 << sample_bits
    t0 = time.perf_counter()
    for w in widths:
        vals = rng.integers(0, 2 ** 32, size=(w, n), dtype=np.uint64)
        oracle_lib.commit_reference(L, vals)
    dt = time.perf_counter() - t0
    return dt, os.cpu_count()


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import eth_lc_plonky2_amd as m

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    stream = torch.cuda.current_stream(dev)
    ctx = m.Context(local, stream=stream.cuda_stream)

    n = 1 << a.degree_bits
    widths = [135, 20, 16]  # wires, Z + partial products, quotient chunks
    traces = [synth_trace(torch, dev, w, n, 1000 + rank * 10 + i) for i, w in enumerate(widths)]

    def step():
        for t, w in zip(traces, widths):
            o = ctx.commit_values(t.data_ptr(), mem=m.MEM_DEVICE, shape=(w, n))
            o.close()

    for _ in range(a.warmup):
        step()
    ctx.prof_reset()
    ctx.prof_enable(True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ctx.prof_enable(False)
    prof = ctx.prof_get()

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = world * a.steps / dt * 3600.0
        lh = prof["leaf_hash"]
        avg_ms = lh["ms"] / max(lh["launches"], 1)
        achieved = (lh["bytes"] / max(lh["launches"], 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "lc_proofs_per_hour", "value": value, "unit": "proofs/hr", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "PARTIAL (round-1 bring-up): wires/Z/quotient commitments (K1-K4) of the n=2^%d, W=135 light-client proof" % a.degree_bits,
                       "degree_bits": a.degree_bits, "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "k_hash_leaves", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms},
            "kernels": {k: v for k, v in prof.items() if v["launches"]},
        }
        if not a.no_cpu_baseline and world == 1:
            cdt, cores = cpu_baseline(a.cpu_sample_bits, widths)
            scale = (1 << a.degree_bits) / (1 << a.cpu_sample_bits)
            out["cpu_baseline"] = {"value": 3600.0 / (cdt * scale), "unit": "proofs/hr", "cores": cores, "kind": "port",
                                   "sample": "oracle commitments of 135/20/16 columns at 2^%d rows (%.1f s), scaled linearly to 2^%d" % (a.cpu_sample_bits, cdt, a.degree_bits)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
