#!/usr/bin/env python3
"""Headline benchmark: light-client proof wall-time / proofs per hour on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one `data.prove(witness)` (the region the reference times, eth-lc-plonky2/src/main.rs:229-232)
over the BASELINE workload configs[2]: a light-client-sized circuit, n = 2^22 rows, 135 wires (80 routed),
standard_recursion_config (rate 1/8, cap height 4, 2 challenges, quotient degree factor 8, 5 arity-16 FRI
layers, 16 PoW bits, 28 queries).  The circuit is the synthetic satisfiable one of
eth-lc-plonky2_amd/circuit.py over plonky2's own gate set (NoopGate, ConstantGate, PublicInputGate,
BaseSumGate<2>, ArithmeticGate, PoseidonGate as gate programs; the public inputs hashed in-circuit as
circuit_builder.rs::build does; real copy constraints); the witness is resident in HBM when the timed region
starts and the proof produced in the last step is checked by the verifier.  Side fields carry the same
prover on circuits built from the reference's own gadgets by the C++ host layer, device-side witness
generation included: `config.sync_committee_ssz` (BASELINE configs[1]: the SyncCommitteeSSZ gadget alone),
`config.real_lc_step` (updates 633 -> 634, 2^19 rows in the own SHA-256 layout), `config.real_lc_step_recursive`
(the same with the recursive verification of an inner proof that has the BLS proof's public inputs) and
`config.real_gadget_circuit_2p22` (the step plus six more SyncCommitteeSSZ gadgets: 2.24 M gates, 2^22 rows).

`--gpus N` without WORLD_SIZE in the environment: this process only LAUNCHES the N ranks (child processes of this script with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; it never imports torch or touches HIP), relays rank 0's line and passes a failing
rank's exit code on.  With WORLD_SIZE set (torch.distributed.run) the process is a rank.

N > 1: one rank per GPU, every rank proves its own witness of the same circuit (BASELINE configs[4],
independent light-client updates: "replicas", no data-path collective) -> weak scaling; the only
collectives are the timing barrier and the max-reduce of the elapsed time.  After that timed region the same
ranks prove ONE proof together, sharded by LDE coset with the witness arriving column-sharded (BASELINE
configs[3]; eth-lc-plonky2_amd/parallel.py::ShardedProver over RCCL: all-gathers of the witness, its
coefficients and the quotient planes, sum all-reduces of caps and proof shares); its wall time is reported
beside the headline value as `config.sharded_proof` (`value` stays the replica throughput; a failure of the sharded proof is a
non-zero exit code).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (Poseidon leaf hashing, K4a):
algorithmic bytes per launch / HIP-event time per launch measured inside this run.  `cpu_baseline` is the
oracle (oracle/, kind "port") timed on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, a full-rate wave64 VALU instruction issues in 2 cycles, 2.4 GHz -> 78.6 T lane-instructions/s
VALU_PEAK_LANE_INSTR = 256 * 4 * 32 * 2.4e9
CLOCK_HZ = 2.4e9
CENSUS = os.path.join(ROOT, "profiles", "r03_poseidon_census.json")  # tools/poseidon_census.py: VALU instructions per permutation as compiled


def valu_roofline(perms_per_s):
    """Poseidon is integer-VALU bound.  One denominator: the full-rate VALU peak of the guide (every wave64 instruction at 2 cycles,
    78.6 T lane-instructions/s).  No stream of 64-bit integer multiplies can reach it on this ISA - v_mad_u64_u32 and the carry ops
    issue at 4.2-4.4 cycles alone (profiles/r02_ubench_int_rates.txt) - so the fraction is a lower bound on how busy the VALU is; the
    measured cycles per instruction say the rest.  (Round 2 also printed a fraction of a "solo-class issue floor"; the kernel beat that
    floor by 9 %, so it was a mis-model and is gone.)"""
    try:
        instr = json.load(open(CENSUS))["valu_instructions"]
    except (OSError, KeyError, ValueError):
        return {"permutations_per_s": perms_per_s}
    lane_instr = perms_per_s * instr
    return {"permutations_per_s": perms_per_s, "valu_instructions_per_permutation": instr,
            "achieved_lane_instr_per_s": lane_instr, "full_rate_peak_lane_instr_per_s": VALU_PEAK_LANE_INSTR,
            "frac_of_full_rate_peak": lane_instr / VALU_PEAK_LANE_INSTR,
            "cycles_per_valu_instruction_per_simd": 256 * 4 * 64 * CLOCK_HZ / lane_instr}


HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")  # tools/pmc_traffic.py on the --pmc passes of this bench (tools/collect_profiles.sh)


def pmc_traffic_bytes(kernel, algorithmic_bytes_per_launch):
    """HBM bytes per launch of `kernel` from the committed PMC passes (counters cannot be read from inside the process):
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE) / algorithmic bytes, both per launch of the PMC pass,
    applied to the algorithmic bytes per launch of THIS run (the passes ran 1 proof, a default run averages 7)."""
    try:
        k = json.load(open(PMC_TRAFFIC))["kernels"][kernel]
        return k["traffic_over_algorithmic"] * algorithmic_bytes_per_launch
    except (OSError, KeyError, ValueError):
        return None


def real_lc_step(extra_committees=0, recursive=False, sync_committee_only=False):
    """Not the headline number: the reference's own update pair 633 -> 634 through examples/lc_prover (the C++ host layer's
    light-client circuit in its own SHA-256 layout), if the binary has been built.  The proof time includes the device-side witness
    generation (K10).  extra_committees = 6 adds six more SyncCommitteeSSZ gadgets: 7 207 two_to_one_sha256, 2.24 M gates, 2^22 rows -
    the reference's scale (README.md:71) made of real gadgets.  recursive: the circuit also verifies, as the reference's does, a proof
    with the BLS proof's 25 216 public inputs (of the stand-in statement circuit: the BLS12-381 verifier itself is out of scope)."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "examples", "lc_prover")
    golden = os.path.join(ROOT, "tests", "golden", "lc_updates.json")
    if not (os.path.exists(exe) and os.path.exists(golden)):
        return None
    try:
        lc = json.load(open(golden))
        with tempfile.TemporaryDirectory() as d:
            paths = []
            for tag in ("633", "634"):
                paths.append(os.path.join(d, "u%s.json" % tag))
                json.dump(lc[tag], open(paths[-1], "w"))
            env = dict(os.environ, LCP2_PROF="1")
            r = subprocess.run([exe] + paths + ["--repeat", "3", "--extra-committees", str(extra_committees)] + (["--bls-proof-stand-in"] if recursive else [])
                               + (["--sync-committee-only"] if sync_committee_only else []),
                               capture_output=True, text=True, timeout=600, env=env)
        ms = [float(x) for x in re.findall(r"proved in ([0-9.]+) ms", r.stdout)]
        bits = re.search(r"degree_bits (\d+)", r.stdout)
        gates = re.search(r"(\d+) gates", r.stdout)
        kern = {k: float(v) for k, v in re.findall(r"^\s+([a-z0-9_]+)\s+([0-9.]+) ms", r.stdout, re.M)}
        if r.returncode != 0 or len(ms) < 3 or not bits:
            return None
        what = "light-client step for updates 633 -> 634 (examples/lc_prover)"
        if sync_committee_only:
            what = ("configs[1]: the SyncCommitteeSSZ gadget alone (512 pubkeys + aggregate key of update 634's next_sync_committee -> SSZ root, 1 025 "
                    "two_to_one_sha256; the reference's test_ssz_sync_committee), root checked against the native SSZ root")
        inner = re.search(r"inner proof .*: 2\^(\d+) rows, (\d+) public inputs, build ([0-9.]+) ms, inner prove ([0-9.]+) ms", r.stdout)
        if recursive and inner:
            what += (" with the recursive verification of a 2^%s-row inner proof that has the BLS proof's %s public inputs (stand-in statement circuit; "
                     "inner proof %s ms, not in ms_per_proof)" % (inner.group(1), inner.group(2), inner.group(4)))
        if extra_committees:
            what += " + %d more SyncCommitteeSSZ gadgets: %s gates" % (extra_committees, gates.group(1) if gates else "?")
        return {"workload": what + ", device witness generation included, proof verified", "degree_bits": int(bits.group(1)),
                "ms_per_proof": min(ms[1:]), "kernel_ms_last_proof": kern}
    except Exception:  # a side measurement must never take the bench line down
        return None


def sharded_proof(m, ctx, circ, cs_ptr, w_dev, pis, rank, world, dist, dev, reps=3):
    """BASELINE configs[3]: one proof of the same circuit sharded over the `world` GPUs by LDE coset.  Every rank brings only its
    column shard of the witness (a slice of the resident tensor stands in for the column-sharded host upload)."""
    import numpy as np
    import torch
    comm = m.parallel.TorchComm(dist, dev, ctx)
    form = comm.self_check()  # known-answer all-gather on a library buffer: "in-place", or "staged" if the aliased form misbehaves
    prover = m.parallel.ShardedProver(ctx, circ, rank, world, comm, constants_sigmas_ptr=cs_ptr, mem=m.MEM_DEVICE)
    prover.finish_build()
    n = 1 << circ.params.degree_bits
    first, end = prover.column_shard()
    shard_ptr = w_dev.data_ptr() + 8 * first * n
    rows = bool(comm.row_exchange_ok) and os.environ.get("LCP2_SHARDED_WHOLE_COLUMNS", "0") != "1"
    times = []
    for it in range(reps + 1):
        torch.cuda.synchronize()
        dist.barrier()
        comm.bytes_gathered = 0
        comm.seconds = dict.fromkeys(comm.seconds, 0.0)
        t0 = time.perf_counter()
        proof = prover.prove(shard_ptr, pis, mem=m.MEM_DEVICE, sharded_columns=True, row_exchange=rows)
        torch.cuda.synchronize()
        dist.barrier()
        if it:  # the first proof warms RCCL's channels up
            times.append(time.perf_counter() - t0)
    tt = torch.tensor([min(times)], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ok = None
    if rank == 0:
        digest, cap = prover.data.digest()
        m.CircuitData.verifier_only(circ, digest, cap).verify(proof, pis)  # raises if the assembled proof is not accepted
        ok = True
    prover.close()
    n_words = 1 << (circ.params.degree_bits + circ.params.rate_bits)
    return {"workload": "configs[3]: the same n=2^%d proof sharded by LDE coset over %d GPUs, witness arriving column-sharded" % (circ.params.degree_bits, world),
            "ms_per_proof": float(tt.item()) * 1e3, "world": world, "rccl_world_size": dist.get_world_size(), "proof_verified": ok,
            "all_gather_form": form, "witness_values_exchange": "all-to-all of row blocks" if rows else "all-gather of whole columns",
            "exchange_bytes_received_per_rank": int(comm.bytes_gathered),
            "exchange_ms_rank0_last_proof": {k: round(1e3 * v, 3) for k, v in comm.seconds.items()},  # wall time inside the synchronous collectives
            "exchange": ("RCCL all_gather_into_tensor (in place): witness coefficients, %s, %d planes of per-coset quotient interpolants; %s"
                         "all_reduce(SUM) of 3 caps and the proof array")
                        % ("Z / partial-product rows" if rows else "witness values", circ.params.num_challenges,
                           "all_to_all_single: witness values as row blocks; " if rows else "")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--degree-bits", type=int, default=22)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="also run the sharded-proof side measurement with one rank "
                    "(RCCL process group of size 1: exercises the N > 1 code path on a one-GPU box)")
    ap.add_argument("--cpu-sample-bits", type=int, default=18, help="log2 rows of the oracle's bounded sample")
    ap.add_argument("--no-real-gadgets", action="store_true", help="skip the examples/lc_prover side measurements")
    ap.add_argument("--no-sharded", action="store_true", help="N > 1: skip the sharded-proof measurement (configs[3]) after the replica run")
    ap.add_argument("--launch-dry-run", action="store_true", help="start the rank processes, let each report the environment it was "
                    "given and exit before anything touches torch or the GPU (CPU test of the launcher)")
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(a):
    """`python bench.py --gpus N` without WORLD_SIZE in the environment: this process is only the launcher.  It starts N rank
    processes of this same script (plain child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set: what
    torch.distributed.run would have set), waits for them and relays rank 0's JSON line.  It never imports torch and never makes
    a HIP call (a process that has initialised the GPU must not exec or be replaced; children are started, not exec'ed into).
    A rank that fails ends the run with its exit code: the others are given a moment to fall out of their collectives, then
    terminated by PID."""
    import subprocess
    assert "torch" not in sys.modules
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    args = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, LCP2_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it across processes)
        # rank 0 (and, in a dry run, every rank) writes its line into a pipe; the other ranks' stdout joins stderr
        out = subprocess.PIPE if (r == 0 or a.launch_dry_run) else sys.stderr
        procs.append(subprocess.Popen(args, env=env, stdout=out))
    import threading
    captured = {}

    def drain(r, pipe):  # a pipe nobody reads would block its writer once the buffer is full
        captured[r] = pipe.read().decode()

    readers = [threading.Thread(target=drain, args=(r, p.stdout), daemon=True) for r, p in enumerate(procs) if p.stdout is not None]
    for t in readers:
        t.start()
    rc, failed_at, terminated = 0, None, False
    live = set(range(a.gpus))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc, failed_at = code, time.monotonic()
                print("bench.py launcher: rank %d exited with code %d" % (r, code), file=sys.stderr)
        if failed_at is not None and live and not terminated and time.monotonic() - failed_at > 30.0:
            for r in live:
                procs[r].terminate()  # the exact PIDs this launcher started
            terminated = True
        if live:
            time.sleep(0.2)
    for t in readers:
        t.join()
    lines = [ln for r in sorted(captured) for ln in captured[r].splitlines() if ln.strip()]
    if a.launch_dry_run:
        kids = [json.loads(ln) for ln in lines]
        print(json.dumps({"launch_dry_run": True, "gpus": a.gpus, "parent_imported_torch": "torch" in sys.modules,
                          "children": sorted(kids, key=lambda k: k["rank"])}))
    else:
        for ln in lines:
            print(ln)
        if not lines and rc == 0:
            rc = 4  # rank 0 printed nothing
    sys.stdout.flush()
    return rc


def cpu_baseline(sample_bits, degree_bits):
    """Oracle prove() on a 2^sample_bits-row circuit of the same gate set, on min(32, cores) OpenMP threads (the reference's
    published figure is for 32 vCPU, README.md:71), scaled to 2^degree_bits: the transforms (measured separately on the sample: the
    oracle's LDE of the 135 wire columns, x 1.4 for the Z / quotient / FRI columns) as n log n, everything else linearly in the row
    count.  A 2^20-row sample would take ~2.5 minutes on the box, too long for the default run; 2^18 takes ~35 s."""
    import ctypes
    import numpy as np
    import eth_lc_plonky2_amd as m
    import oracle_lib
    L = oracle_lib.load()
    params = m.standard_params(sample_bits, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=1, small_values=True)
    threads = 1
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        threads = min(32, omp.omp_get_max_threads())
        omp.omp_set_num_threads(threads)
    except OSError:
        pass
    oc = oracle_lib.OracleCircuit(L, circ)
    t0 = time.perf_counter()
    proof = oc.prove(wires, pis)
    dt = time.perf_counter() - t0
    assert oc.verify(proof, pis) == 0
    oc.close()
    n = 1 << sample_bits
    cols = np.ascontiguousarray(wires[:, :] % np.uint64(m.GOLDILOCKS_P))
    lde = np.zeros((cols.shape[0], n << params.rate_bits), dtype=np.uint64)
    t0 = time.perf_counter()
    L.orc_lde_batch(oracle_lib.vp(cols), cols.shape[0], n, params.rate_bits, 7, oracle_lib.vp(lde))
    t_ntt = min(1.4 * (time.perf_counter() - t0), 0.9 * dt)
    del lde
    lg_s, lg_d = sample_bits + params.rate_bits, degree_bits + params.rate_bits
    rows = float(1 << (degree_bits - sample_bits))
    est = rows * ((dt - t_ntt) + t_ntt * lg_d / lg_s)
    return {"value": 3600.0 / est, "unit": "proofs/hr", "cores": threads, "kind": "port",
            "sample": "oracle prove() of the same gate set at 2^%d rows: %.2f s on %d OpenMP threads, of which transforms ~%.2f s; scaled to 2^%d rows "
                      "(transforms x%d x %d/%d for n log n, the rest x%d): %.0f s per proof"
                      % (sample_bits, dt, threads, t_ntt, degree_bits, int(rows), lg_d, lg_s, int(rows), est)}


def main():
    a = parse()
    if a.launch_dry_run and "WORLD_SIZE" in os.environ:  # a child of the dry run: report and leave, nothing imported
        print(json.dumps({k.lower(): os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                         | {"rank": int(os.environ["RANK"]), "gpus_flag": a.gpus, "imported_torch": "torch" in sys.modules}))
        return 0
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.launch_dry_run):
        return launch(a)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%s in the environment: the launcher's world size wins" % (a.gpus, os.environ["WORLD_SIZE"]),
              file=sys.stderr)
    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner when its first communicator comes
    # up) write to file descriptor 1 directly, so fd 1 is pointed at stderr for the whole run and the line goes out through a
    # saved duplicate of the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    import eth_lc_plonky2_amd as m

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or a.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # a rank that fails inside the sharded side measurement must not take the bench line with it: collectives raise after the
        # timeout instead of the watchdog aborting every rank (the replica measurement itself uses one barrier and one all-reduce)
        os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=300))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    stream = torch.cuda.current_stream(dev)
    ctx = m.Context(local, stream=stream.cuda_stream)
    ctx.prof_enable(True)  # HIP-event timing of every kernel family from the first launch (same population as rocprofv3)

    # ---- build(): circuit description on the host, preprocessed polynomials committed on the GPU (untimed)
    params = m.standard_params(a.degree_bits, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=3, small_values=True)
    cs_dev = torch.from_numpy(circ.constants_sigmas.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    data = m.CircuitData.build(ctx, circ, constants_sigmas_ptr=cs_dev.data_ptr(), mem=m.MEM_DEVICE)
    if world == 1 and not a.force_sharded:
        del cs_dev  # N > 1 keeps it for the sharded circuit built after the timed region
    # each rank proves its own witness: the free cells of the padding row carry the (rank, update) tag
    m.circuit.tag_witness(wires, rank + 1)
    w_dev = torch.from_numpy(wires.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    del wires  # 4.5 GB of host memory per rank; the witness lives in HBM from here on
    circ.constants_sigmas = np.zeros((1, 1), dtype=np.uint64)  # 2.8 GB: the preprocessed columns are on the device too
    torch.cuda.empty_cache()

    def step():
        return data.prove(w_dev.data_ptr(), pis, mem=m.MEM_DEVICE)

    for _ in range(a.warmup):
        step()
    prof0 = ctx.prof_get()  # build() + warmup launches; subtracted for the per-proof table

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    proof = None
    for _ in range(a.steps):
        proof = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ctx.prof_enable(False)
    prof_all = ctx.prof_get()
    prof = {k: {f: prof_all[k][f] - prof0[k][f] for f in ("ms", "launches", "bytes")} for k in prof_all}
    data.verify(proof, pis)  # raises if the GPU proof is not accepted
    sharded, rc = None, 0
    rccl_world = dist.get_world_size() if (world > 1 or a.force_sharded) else 1  # as the RCCL process group reports it
    reported = []  # the JSON line goes out exactly once: normally after the sharded side measurement, or from its watchdog

    def report(sharded):
        if rank != 0 or reported:
            return
        reported.append(True)
        nonlocal w_dev
        ms_per_step = dt / a.steps * 1e3
        value = world * a.steps / dt * 3600.0
        lh = prof_all["leaf_hash"]  # every k_hash_leaves launch of this process, as rocprofv3 --stats averages them
        avg_ms = lh["ms"] / max(lh["launches"], 1)
        achieved = (lh["bytes"] / max(lh["launches"], 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # the kernel is integer-VALU bound: permutations/s x instructions per permutation against the VALU issue peak
        perms = (-(-params.num_wires // 8) + 3 + 2) * float(1 << (a.degree_bits + 3)) * a.steps  # wires, 20 Z columns, 16 quotient chunks
        perms_per_s = perms / max(prof["leaf_hash"]["ms"] * 1e-3, 1e-12)
        kern = {}
        for k, v in prof.items():
            if v["launches"]:
                per = v["ms"] / a.steps
                kern[k] = {"ms_per_proof": round(per, 3), "scopes_per_proof": v["launches"] / a.steps,
                           "algorithmic_GB_per_proof": round(v["bytes"] / a.steps / 1e9, 3),
                           "achieved_GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)}
        out = {
            "metric": "lc_proofs_per_hour", "value": value, "unit": "proofs/hr", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[2]: full light-client-sized proof, n=2^%d rows x 135 wires, standard_recursion_config, "
                                   "synthetic satisfiable circuit over plonky2's own gate set (Noop, Constant, PublicInput, BaseSum, Arithmetic, Poseidon; public inputs hashed in-circuit), recursive BLS verifier stubbed" % a.degree_bits,
                       "degree_bits": a.degree_bits, "proof_wall_time_s": ms_per_step / 1e3, "proof_verified": True,
                       "parallelism": "replicas x%d (one independent proof per GPU)" % world,
                       "rccl_world_size": rccl_world, "replica_proofs_per_hour": value},
            "roofline": {"bound": "hbm", "kernel": "k_hash_leaves", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes("k_hash_leaves", lh["bytes"] / max(lh["launches"], 1)),
                         "algorithmic_bytes_per_launch": lh["bytes"] / max(lh["launches"], 1), "avg_launch_ms": avg_ms, "launches": lh["launches"],
                         "note": "integer-VALU bound (Poseidon), so the HBM fraction is legitimately low: see `valu` and DESIGN.md section 3",
                         "valu": valu_roofline(perms_per_s)},
            "kernels": kern,
        }
        if sharded:
            out["config"]["sharded_proof"] = sharded
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(min(a.cpu_sample_bits, a.degree_bits), a.degree_bits)
        if world == 1 and not a.no_real_gadgets:
            data.close()  # the side measurements run in a child process: give the 92 GB workspace and the witness back first
            w_dev = None
            torch.cuda.empty_cache()
            step_633 = real_lc_step()
            if step_633:
                out["config"]["real_lc_step"] = step_633
            ssz = real_lc_step(sync_committee_only=True)
            if ssz:
                out["config"]["sync_committee_ssz"] = ssz
            rec = real_lc_step(recursive=True)
            if rec:
                out["config"]["real_lc_step_recursive"] = rec
            big = real_lc_step(extra_committees=6)
            if big:
                out["config"]["real_gadget_circuit_2p22"] = big
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    if (world > 1 or a.force_sharded) and not a.no_sharded and (world & (world - 1)) == 0 and world <= (1 << params.rate_bits):
        data.close()  # the replica's 92 GB workspace makes room for the sharded handle
        torch.cuda.empty_cache()
        # A collective that never completes (this exchange has not run over RCCL with more than one rank yet) must not take the
        # replica result down with it: after LCP2_SHARDED_TIMEOUT_S the line goes out with the error and the rank exits non-zero.
        import threading

        def give_up():
            report({"error": "the sharded proof did not finish within %s s (a collective hangs?)" % limit})
            os._exit(3)
        limit = float(os.environ.get("LCP2_SHARDED_TIMEOUT_S", "240"))
        watchdog = threading.Timer(limit if rank == 0 else limit + 5.0, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            sharded = sharded_proof(m, ctx, circ, cs_dev.data_ptr(), w_dev, pis, rank, world, dist, dev)
        except Exception as e:  # the replica line still goes out, but the run fails: configs[3] is a first-class result for N > 1
            import traceback
            traceback.print_exc()
            sharded = {"error": "%s: %s" % (type(e).__name__, e)}
            rc = 3
        finally:
            watchdog.cancel()

    report(sharded)
    if world > 1 or a.force_sharded:
        try:
            dist.destroy_process_group()
        except Exception:  # after a failed collective the group may not shut down cleanly; the exit code already says so
            rc = rc or 3
    return rc


if __name__ == "__main__":
    sys.exit(main())
