#!/usr/bin/env python3
"""Headline benchmark: light-client proof wall-time / proofs per hour on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one `data.prove(pw)` of eth-lc-plonky2/src/main.rs:229-232 - the region the reference times, witness generation
(`generate_partial_witness`) INSIDE it - on a circuit built from the reference's own gadgets by the C++ host layer
(eth-lc-plonky2_amd/host, through host/lc_capi.h in this process): the light-client step for the reference's update pair 633 -> 634
(add_virtual_proof_target, src/targets.rs:391-683; 16 public inputs cur_state / new_state) plus six more SyncCommitteeSSZ gadgets,
7 207 two_to_one_sha256, 2 240 740 gates = 2^22 rows of 135 wires under standard_recursion_config - the reference's scale
(README.md:71, "~2.98 M gates"; this library's SHA-256 layout needs 310 rows per hash where plonky2_crypto needs ~2 800, so
the bare step is 2^19 rows) made of real gadgets.  The recursive BLS verifier is stubbed (BASELINE configs[2]).  The PartialWitness
is a host object (a few thousand values); the SHA-256 rows are generated on the device (K10), the proof is verified and its public
inputs compared with the natively computed contract states.  `data`: the two real mainnet updates of the reference repository
(tests/golden/lc_updates.json), nothing synthetic.

Side fields (`config.*`, N = 1 only, each with the HIP-event time per kernel family and a roofline block of its own):
  real_lc_step / sync_committee_ssz / real_lc_step_recursive   the step at its true size (2^19 rows), BASELINE configs[1] (the
      SyncCommitteeSSZ gadget alone), and the step with the recursive verification of a stand-in inner proof;
  synthetic_plonky2_gate_set_2p22     the headline of rounds 1-3: a synthetic satisfiable circuit over plonky2's own gate set
      (Noop, Constant, PublicInput, BaseSum, Arithmetic, Poseidon) at 2^22 rows, witness resident in HBM (lcp2_prove alone);
  reference_gate_set_2p22             the same with the reference's real gate set (plonky2_u32 U32AddMany / U32Arithmetic / U32RangeCheck /
      U32Subtraction, ComparisonGate next to BaseSum / Arithmetic / Poseidon) on generated native evaluators: quotient time, VGPRs per
      gate kernel, K6 FETCH_SIZE ratio of the committed PMC pass;
  host_witness_2p22                   lcp2_prove with the witness in HOST memory (what a fork with its own generators hands over):
      pinned, double-buffered upload of witness i+1 on a second stream while proof i runs.

`--gpus N` without WORLD_SIZE in the environment: this process only LAUNCHES the N ranks (child processes of this script with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; it never imports torch or touches HIP), relays rank 0's line and passes a failing
rank's exit code on.  With WORLD_SIZE set (torch.distributed.run) the process is a rank.

N > 1: one rank per GPU, every rank proves the same circuit for its own update (BASELINE configs[4], independent light-client
updates: "replicas", no data-path collective) -> weak scaling; the only collectives are the timing barrier and the max-reduce of the
elapsed time.  After that timed region the ranks (a) prove a batch of 32 updates data-parallel through batch.prove_batch (configs[4]
as worded: 32 / N per rank, proofs gathered on rank 0) and (b) prove ONE proof together, sharded by LDE coset (BASELINE configs[3];
eth-lc-plonky2_amd/parallel.py::ShardedProver over RCCL) - reported as `config.batch_of_32` and `config.sharded_proof`; `value` stays
the replica throughput; a failure or a hang of one of these side measurements is reported inside the line (`.error`) and the exit code stays 0
unless --strict-sharded is given.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (Poseidon leaf hashing, K4a, integer-VALU bound):
algorithmic bytes per launch / HIP-event time per launch measured inside this run.  `cpu_baseline` is the oracle (oracle/, kind
"port") timed on a bounded sample; the GPU proves the same sample and the two proofs must be equal word for word.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, a full-rate wave64 VALU instruction issues in 2 cycles, 2.4 GHz -> 78.6 T lane-instructions/s
VALU_PEAK_LANE_INSTR = 256 * 4 * 32 * 2.4e9
CLOCK_HZ = 2.4e9
CENSUS = os.path.join(ROOT, "profiles", "r04_poseidon_census.json")  # tools/poseidon_census.py: VALU instructions per permutation as compiled
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")  # tools/pmc_traffic.py on the --pmc passes of the bench (tools/collect_profiles.sh)
REFERENCE_MIX_PMC = os.path.join(ROOT, "profiles", "r04_reference_mix_k6.json")  # tools/k6_profile_summary.py: K6 kernel times and FETCH_SIZE of the reference gate set


def valu_roofline(perms_per_s):
    """Poseidon is integer-VALU bound.  One denominator: the full-rate VALU peak of the guide (every wave64 instruction at 2 cycles,
    78.6 T lane-instructions/s).  No stream of 64-bit integer multiplies can reach it on this ISA - v_mad_u64_u32 and the carry ops
    issue at 4.2-4.4 cycles alone (profiles/r02_ubench_int_rates.txt) - so the fraction is a lower bound on how busy the VALU is; the
    measured cycles per instruction say the rest."""
    try:
        instr = json.load(open(CENSUS))["valu_instructions"]
    except (OSError, KeyError, ValueError):
        return {"permutations_per_s": perms_per_s}
    lane_instr = perms_per_s * instr
    return {"permutations_per_s": perms_per_s, "valu_instructions_per_permutation": instr,
            "achieved_lane_instr_per_s": lane_instr, "full_rate_peak_lane_instr_per_s": VALU_PEAK_LANE_INSTR,
            "frac_of_full_rate_peak": lane_instr / VALU_PEAK_LANE_INSTR,
            "cycles_per_valu_instruction_per_simd": 256 * 4 * 64 * CLOCK_HZ / lane_instr}


def pmc_traffic_bytes(kernel, algorithmic_bytes_per_launch):
    """HBM bytes per launch of `kernel` from the COMMITTED PMC passes (counters cannot be read from inside the process):
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE) / algorithmic bytes, both per launch of the PMC pass,
    applied to the algorithmic bytes per launch of THIS run."""
    try:
        k = json.load(open(PMC_TRAFFIC))["kernels"][kernel]
        return k["traffic_over_algorithmic"] * algorithmic_bytes_per_launch
    except (OSError, KeyError, ValueError):
        return None


def leaf_perms(degree_bits, proofs, builds=0, rate_bits=3):
    """Poseidon permutations of the leaf hashing of `proofs` proofs (135 wires, 20 Z / partial-product columns, 16 quotient chunks: 17 + 3 + 2
    per leaf) and `builds` build()s (84 or 85 constant and sigma columns: 11 per leaf) at 2^degree_bits rows"""
    return float(1 << (degree_bits + rate_bits)) * (22 * proofs + 11 * builds)


def leaf_hash_roofline(lh, perms):
    """roofline block of k_hash_leaves (K4a, the dominant kernel of every workload here) from the HIP-event totals `lh` of its kernel
    family {ms, launches, bytes} and the permutations those launches computed: algorithmic bytes per launch / average launch time against
    the HBM peak (legitimately low: the kernel is integer-VALU bound), and the VALU fraction next to it."""
    launches = max(lh["launches"], 1)
    avg_ms = lh["ms"] / launches
    per_launch = lh["bytes"] / launches
    achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    perms_per_s = perms / max(lh["ms"] * 1e-3, 1e-12)
    traffic = pmc_traffic_bytes("k_hash_leaves", per_launch)
    return {"bound": "valu", "kernel": "k_hash_leaves", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_from_committed_pmc": True, "traffic_source": "profiles/r04_pmc_traffic.json (ratio of the PMC pass applied to this run's algorithmic bytes)",
            "algorithmic_bytes_per_launch": per_launch, "avg_launch_ms": avg_ms, "launches": lh["launches"],
            "note": "integer-VALU bound (Poseidon): `frac` is the HBM fraction the contract asks for and is legitimately low; `valu` is the bound that binds (DESIGN.md section 3)",
            "valu": valu_roofline(perms_per_s)}


def kernel_table(prof, steps):
    out = {}
    for k, v in prof.items():
        if v["launches"]:
            out[k] = {"ms_per_proof": round(v["ms"] / steps, 3), "scopes_per_proof": v["launches"] / steps,
                      "algorithmic_GB_per_proof": round(v["bytes"] / steps / 1e9, 3), "achieved_GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)}
    return out


def prof_delta(p1, p0):
    return {k: {f: p1[k][f] - p0[k][f] for f in ("ms", "launches", "bytes")} for k in p1}


# ---------------------------------------------------------------- workloads on the C++ host layer (in-process, host/lc_capi.h)
def light_client_workload(m, ctx, sync, flags=0, extra_committees=0, steps=3, warmup=1):
    """`steps` timed data.prove(pw) of the light-client circuit for updates 633 -> 634 (+ extra SyncCommitteeSSZ gadgets); the last
    proof is verified and its public inputs compared with the natively computed ones.  sync(): barrier + device synchronisation.
    Returns (seconds for `steps` proofs, per-family profile delta, info, description)."""
    import numpy as np
    prev, cur = m.light_client.reference_updates()
    step = m.light_client.LightClientStep(ctx, prev, cur, flags=flags, extra_committees=extra_committees)
    for _ in range(warmup):
        step.prove()
    p0 = ctx.prof_get()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        proof, pis = step.prove()
    sync()
    dt = time.perf_counter() - t0
    prof = prof_delta(ctx.prof_get(), p0)
    step.verify(proof, pis)  # raises if the GPU proof is not accepted
    assert (np.asarray(pis) == step.expected_public_inputs).all(), "the proved public inputs are not the natively computed ones"
    info = step.info
    what = "light-client step for the reference's updates 633 -> 634 (add_virtual_proof_target + set_proof_target, 16 public inputs)"
    if flags & m.light_client.SYNC_COMMITTEE_ONLY:
        what = ("configs[1]: the SyncCommitteeSSZ gadget alone (512 pubkeys + aggregate key of update 634's next_sync_committee -> SSZ root, 1 025 two_to_one_sha256; "
                "the reference's test_ssz_sync_committee), root = the native SSZ root")
    if flags & m.light_client.BLS_PROOF_STAND_IN:
        what += (" with the recursive verification of a 2^%d-row inner proof that has the BLS proof's %d public inputs (stand-in statement circuit: no statement about the "
                 "signature; inner proof %.1f ms, not in ms_per_proof)" % (info.inner_degree_bits, info.inner_public_inputs, info.inner_prove_ms))
    if extra_committees:
        what += " + %d more SyncCommitteeSSZ gadgets" % extra_committees
    what += ": %d gates, 2^%d rows; data.prove(pw) with witness generation inside (SHA-256 rows generated on the device), proof verified" % (info.num_gates, info.degree_bits)
    out = (dt, prof, {"degree_bits": info.degree_bits, "gates": int(info.num_gates), "build_ms": round(info.build_ms, 1), "attach_ms": round(info.attach_ms, 1)}, what)
    step.close()
    return out


def side_light_client(m, ctx, sync, **kw):
    try:
        dt, prof, info, what = light_client_workload(m, ctx, sync, steps=3, warmup=1, **kw)
        return {"workload": what, "degree_bits": info["degree_bits"], "gates": info["gates"], "ms_per_proof": dt / 3 * 1e3,
                "kernel_ms_per_proof": {k: round(v["ms"] / 3, 3) for k, v in prof.items() if v["launches"]},
                "roofline": leaf_hash_roofline(prof["leaf_hash"], leaf_perms(info["degree_bits"], 3)), "host_build_ms": info["build_ms"], "device_build_ms": info["attach_ms"]}
    except Exception as e:  # a side measurement must never take the bench line down
        return {"error": "%s: %s" % (type(e).__name__, e)}


# ---------------------------------------------------------------- the synthetic circuits (lcp2_prove alone, witness resident in HBM)
def synthetic_workload(m, ctx, sync, degree_bits, steps=3, host_witness=False):
    """rounds 1-3's headline: `steps` lcp2_prove calls on the synthetic circuit over plonky2's own gate set, witness resident in HBM.
    host_witness: additionally the same proofs with the witness in host memory (pinned double-buffered upload)."""
    import numpy as np
    import torch
    params = m.standard_params(degree_bits, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=3, small_values=True)
    data = m.CircuitData.build(ctx, circ)
    w_dev = torch.from_numpy(wires.view(np.int64)).cuda()
    torch.cuda.synchronize()
    data.prove(w_dev.data_ptr(), pis, mem=m.MEM_DEVICE)
    p0 = ctx.prof_get()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        proof = data.prove(w_dev.data_ptr(), pis, mem=m.MEM_DEVICE)
    sync()
    dt = time.perf_counter() - t0
    prof = prof_delta(ctx.prof_get(), p0)
    data.verify(proof, pis)
    out = {"workload": "synthetic satisfiable circuit over plonky2's own gate set (Noop, Constant, PublicInput, BaseSum, Arithmetic, Poseidon; public inputs hashed "
                       "in-circuit, real copy constraints), n=2^%d rows x 135 wires, witness resident in HBM: lcp2_prove alone (no witness generation), proof verified" % degree_bits,
           "degree_bits": degree_bits, "ms_per_proof": dt / steps * 1e3, "kernels": kernel_table(prof, steps), "roofline": leaf_hash_roofline(prof["leaf_hash"], leaf_perms(degree_bits, steps))}
    hw = None
    if host_witness and hasattr(data, "host_witness_benchmark"):
        try:
            del w_dev
            torch.cuda.empty_cache()
            hw = data.host_witness_benchmark(wires, pis, steps=steps + 1, reference_proof=proof)
            hw["device_resident_ms_per_proof"] = out["ms_per_proof"]
            hw["host_over_device"] = hw["ms_per_proof_steady_state"] / out["ms_per_proof"]
        except Exception as e:
            hw = {"error": "%s: %s" % (type(e).__name__, e)}
    data.close()
    torch.cuda.empty_cache()
    return out, hw


def reference_gate_set(m, ctx, degree_bits):
    try:
        import reference_mix_probe as probe
        out = probe.measure(ctx, degree_bits, reps=3)
        out["generated_gate_kernels"] = probe.generated_gate_registers()
        try:
            pmc = json.load(open(REFERENCE_MIX_PMC))
            out["k6_fetch_from_committed_pmc"] = {"source": "profiles/r04_reference_mix_k6.json (rocprofv3 --pmc FETCH_SIZE of tools/reference_mix_probe.py 22 1, doubled per the gfx950 correction)",
                                                  "k6_read_GB_per_proof": pmc["k6_read_GB_per_proof"], "k6_algorithmic_GB_per_proof": pmc["k6_algorithmic_GB_per_proof"],
                                                  "fetch_over_algorithmic": pmc["k6_read_GB_per_proof"] / pmc["k6_algorithmic_GB_per_proof"],
                                                  "kernels": pmc["kernels"]}
        except (OSError, KeyError, ValueError):
            pass
        return out
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def sharded_proof(m, ctx, circ, cs_ptr, w_dev, pis, rank, world, dist, dev, reps=3):
    """BASELINE configs[3]: one proof of the same circuit sharded over the `world` GPUs by LDE coset.  Every rank brings only its
    column shard of the witness (a slice of the resident tensor stands in for the column-sharded host upload)."""
    import torch
    comm = m.parallel.TorchComm(dist, dev, ctx)
    form = comm.self_check()  # known-answer all-gather on a library buffer: "in-place", or "staged" if the aliased form misbehaves
    prover = m.parallel.ShardedProver(ctx, circ, rank, world, comm, constants_sigmas_ptr=cs_ptr, mem=m.MEM_DEVICE)
    prover.finish_build()
    n = 1 << circ.params.degree_bits
    first, end = prover.column_shard()
    shard_ptr = w_dev.data_ptr() + 8 * first * n
    rows = bool(comm.row_exchange_ok) and os.environ.get("LCP2_SHARDED_WHOLE_COLUMNS", "0") != "1"
    # the coefficient exchange in chunks of 16 columns, gathered on a side stream while the chunk before is extended and absorbed
    chunked = rows and os.environ.get("LCP2_SHARDED_UNCHUNKED", "0") != "1"
    if chunked:  # this rank's columns of the chunked assignment, gathered out of the resident tensor (stands in for the column-sharded upload)
        cols = torch.tensor(m.parallel.chunk_columns(circ.params.num_wires, rank, world), device=dev)
        mine = w_dev.view(circ.params.num_wires, n)[cols].contiguous()
        shard_ptr = mine.data_ptr()
        torch.cuda.synchronize()
    times = []
    for it in range(reps + 1):
        torch.cuda.synchronize()
        dist.barrier()
        comm.bytes_gathered = 0
        comm.seconds = dict.fromkeys(comm.seconds, 0.0)
        t0 = time.perf_counter()
        proof = prover.prove(shard_ptr, pis, mem=m.MEM_DEVICE, sharded_columns=True, row_exchange=rows, chunked=chunked)
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
    return {"workload": "configs[3]: a n=2^%d proof (synthetic circuit over plonky2's gate set) sharded by LDE coset over %d GPUs, witness arriving column-sharded" % (circ.params.degree_bits, world),
            "ms_per_proof": float(tt.item()) * 1e3, "world": world, "rccl_world_size": dist.get_world_size(), "proof_verified": ok,
            "all_gather_form": form, "witness_values_exchange": "all-to-all of row blocks" if rows else "all-gather of whole columns",
            "coefficient_exchange": ("chunks of %d columns gathered on a side stream, overlapped with their coset LDE and leaf absorption" % m.parallel.CHUNK_COLS) if chunked
                                    else "one all-gather before the commitment",
            "exchange_bytes_received_per_rank": int(comm.bytes_gathered),
            "exchange_ms_rank0_last_proof": {k: round(1e3 * v, 3) for k, v in comm.seconds.items()},  # wall time inside the synchronous collectives
            "exchange": ("RCCL all_gather_into_tensor (in place): witness coefficients, %s, %d planes of per-coset quotient interpolants; %s"
                         "all_reduce(SUM) of 3 caps and the proof array")
                        % ("Z / partial-product rows" if rows else "witness values", circ.params.num_challenges,
                           "all_to_all_single: witness values as row blocks; " if rows else "")}


def guarded(work, limit_s, on_timeout, exit_fn=os._exit, exit_code=3):
    """work() under a watchdog: if it has not returned after limit_s seconds, on_timeout(message) runs on the timer thread and the
    process exits with `exit_code` (a collective that never completes cannot be interrupted from Python).  The decision "finished" /
    "timed out" is taken once, under a lock: a timer that fires while work() is returning finds the work done and does nothing, and
    the caller does not go on before the timer thread has either exited the process or returned."""
    lock = threading.Lock()
    state = {"done": False}

    def give_up():
        with lock:
            if state["done"]:
                return
            state["done"] = True  # from here on the outcome is the timeout
        on_timeout("did not finish within %s s (a collective hangs?)" % limit_s)
        exit_fn(exit_code)
    timer = threading.Timer(limit_s, give_up)
    timer.daemon = True
    timer.start()
    try:
        return work()
    finally:
        with lock:
            finished_first = not state["done"]
            state["done"] = True
        timer.cancel()
        if not finished_first:
            timer.join()  # the timeout won: its thread reports and exits the process (with a test's exit_fn it returns, and so do we)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--extra-committees", type=int, default=6, help="SyncCommitteeSSZ gadgets added to the light-client step of the headline: 6 -> 2.24 M gates, 2^22 rows")
    ap.add_argument("--degree-bits", type=int, default=22, help="log2 rows of the synthetic side workloads (and of the sharded proof for N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="also run the sharded-proof side measurement with one rank "
                    "(RCCL process group of size 1: exercises the N > 1 code path on a one-GPU box)")
    ap.add_argument("--cpu-sample-bits", type=int, default=18, help="log2 rows of the oracle's bounded sample")
    ap.add_argument("--no-real-gadgets", action="store_true", help="skip the light-client side workloads (2^19-row step, configs[1], recursive step)")
    ap.add_argument("--no-synthetic", action="store_true", help="skip the synthetic side workloads (plonky2 gate set, reference gate set, host witness)")
    ap.add_argument("--no-sharded", action="store_true", help="N > 1: skip the sharded-proof measurement (configs[3]) after the replica run")
    ap.add_argument("--strict-sharded", action="store_true", help="N > 1: exit non-zero when the sharded proof or the batch raises (default: the error "
                    "is reported in the line - config.sharded_proof.error / config.batch_of_32.error - and on stderr, and the replica result stands)")
    ap.add_argument("--no-batch", action="store_true", help="N > 1: skip the 32-update batch (configs[4] as worded)")
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
    A rank that fails ends the run with its exit code: the others are given 30 s to fall out of their collectives, then terminated
    by PID, then - 20 s later - killed; LCP2_LAUNCH_DEADLINE_S (default 3000) bounds the whole wait, so the launcher always exits."""
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
    captured = {}

    def drain(r, pipe):  # a pipe nobody reads would block its writer once the buffer is full
        captured[r] = pipe.read().decode()

    readers = [threading.Thread(target=drain, args=(r, p.stdout), daemon=True) for r, p in enumerate(procs) if p.stdout is not None]
    for t in readers:
        t.start()
    rc, failed_at, terminated_at, killed = 0, None, None, False
    deadline = time.monotonic() + float(os.environ.get("LCP2_LAUNCH_DEADLINE_S", "3000"))
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
        now = time.monotonic()
        if live and now > deadline and failed_at is None:
            rc, failed_at = 5, now - 30.0
            print("bench.py launcher: deadline reached with ranks %s still running" % sorted(live), file=sys.stderr)
        if failed_at is not None and live and terminated_at is None and now - failed_at > 30.0:
            for r in live:
                procs[r].terminate()  # the exact PIDs this launcher started
            terminated_at = now
        if terminated_at is not None and live and not killed and now - terminated_at > 20.0:
            for r in live:
                procs[r].kill()  # a rank stuck in an uninterruptible driver call ignores SIGTERM
            killed = True
        if killed and live and now - terminated_at > 40.0:
            print("bench.py launcher: ranks %s do not exit; giving up on them" % sorted(live), file=sys.stderr)
            break
        if live:
            time.sleep(0.2)
    for t in readers:
        t.join(timeout=5.0)
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


def cpu_baseline(m, ctx, sample_bits, degree_bits):
    """Oracle prove() on a 2^sample_bits-row circuit of plonky2's gate set, on min(32, cores) OpenMP threads (the reference's
    published figure is for 32 vCPU, README.md:71), scaled to 2^degree_bits: the transforms (measured separately on the sample: the
    oracle's LDE of the 135 wire columns, x 1.4 for the Z / quotient / FRI columns) as n log n, everything else linearly in the row
    count (profiles/r04_cpu_baseline_scaling.json holds a measured 2^20 point next to this model).  The GPU proves the SAME sample
    circuit and the two proofs must be equal word for word: whole-proof parity at 2^18 rows inside every default run."""
    import ctypes
    import numpy as np
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
    data = m.CircuitData.build(ctx, circ)
    gpu_proof = data.prove(wires, pis)
    equal = bool((gpu_proof == proof).all())
    data.close()
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
    full = None
    try:  # the measured full-size points (separate runs on the GPU box's host, committed: not of this run)
        f = json.load(open(os.path.join(ROOT, "profiles", "r04_parity_full_size.json")))
        full = {"plonky2_gate_set": {
            "source": "profiles/r04_parity_full_size.json (tests/checks/parity_full_size.py: the oracle's prove() at 2^%d rows on %d threads, the GPU proof of the same "
                      "witness compared word for word)" % (f["degree_bits"], f["threads"]),
            "oracle_prove_s": f["oracle_prove_s"], "proofs_per_hour": 3600.0 / f["oracle_prove_s"], "gpu_proof_equal": f["gpu_proof_equal"]}}
        g = json.load(open(os.path.join(ROOT, "profiles", "r04_real_gadget_parity.json")))["headline_circuit_2p22"]
        full["headline_circuit"] = {
            "source": "profiles/r04_real_gadget_parity.json (LCP2_ORACLE_PROVE_ALL=1 tests/cpp/test_gadgets gpu %s: the circuit of this line's headline, 2^%d rows, oracle on %d "
                      "threads, GPU proof compared word for word)" % (g["test"], g["degree_bits"], g["threads"]),
            "oracle_prove_s": g["oracle_prove_s"], "proofs_per_hour": 3600.0 / g["oracle_prove_s"], "gpu_proof_equal": g["gpu_proof_equal"]}
    except (OSError, KeyError, ValueError):
        pass
    return {"value": 3600.0 / est, "unit": "proofs/hr", "cores": threads, "kind": "port", "gpu_proof_equal": equal, "measured_at_full_size": full,
            "sample": "oracle prove() of plonky2's gate set at 2^%d rows: %.2f s on %d OpenMP threads, of which transforms ~%.2f s; scaled to 2^%d rows "
                      "(transforms x%d x %d/%d for n log n, the rest x%d): %.0f s per proof; the GPU proved the same sample: proofs %s word for word"
                      % (sample_bits, dt, threads, t_ntt, degree_bits, int(rows), lg_d, lg_s, int(rows), est, "EQUAL" if equal else "DIFFER")}


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
    ctx = m.Context(local, stream=stream.cuda_stream)  # raises without a GPU: there is no CPU fallback
    ctx.prof_enable(True)  # HIP-event timing of every kernel family from the first launch (same population as rocprofv3)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the headline: data.prove(pw) of the 2^22-row circuit of real gadgets, witness generation inside the timed region
    dt, prof, info, what = light_client_workload(m, ctx, barrier, flags=0, extra_committees=a.extra_committees, steps=a.steps, warmup=a.warmup)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    rccl_world = dist.get_world_size() if (world > 1 or a.force_sharded) else 1  # as the RCCL process group reports it
    prof_all = ctx.prof_get()  # every launch of the process so far (build, warm-up, timed steps): the population rocprofv3 --stats averages
    lock = threading.Lock()
    state = {"reported": False}  # the JSON line goes out exactly once: after the side measurements, or from the watchdog

    def report(sharded, batch, from_watchdog=False):
        with lock:
            if state["reported"]:
                return
            state["reported"] = True
        if rank != 0:
            return
        ms_per_step = dt / a.steps * 1e3
        value = world * a.steps / dt * 3600.0
        out = {
            "metric": "lc_proofs_per_hour", "value": value, "unit": "proofs/hr", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64",
            "data": "the reference's two real light-client updates (periods 633 -> 634, tests/golden/lc_updates.json); nothing synthetic in the headline",
            "config": {"workload": "configs[2]: " + what + "; recursive BLS verifier stubbed",
                       "degree_bits": info["degree_bits"], "gates": info["gates"], "proof_wall_time_s": ms_per_step / 1e3, "proof_verified": True,
                       "witness_generation_in_timed_region": True, "host_build_ms": info["build_ms"], "device_build_ms": info["attach_ms"],
                       "parallelism": "replicas x%d (one independent proof per GPU)" % world,
                       "rccl_world_size": rccl_world, "replica_proofs_per_hour": value},
            "roofline": leaf_hash_roofline(prof_all["leaf_hash"], leaf_perms(info["degree_bits"], a.warmup + a.steps, builds=1)),
            "kernels": kernel_table(prof, a.steps),
        }
        out["roofline"]["population"] = ("every k_hash_leaves launch of this process up to the end of the timed region (build, warm-up, steps): what `rocprofv3 --kernel-trace "
                                         "--stats -- python3 bench.py --no-cpu-baseline --no-real-gadgets --no-synthetic` averages (profiles/r04_full_proof_kernel_stats.csv)")
        if sharded:
            out["config"]["sharded_proof"] = sharded
        if batch:
            out["config"]["batch_of_32"] = batch
        if world == 1 and not from_watchdog:
            sync = barrier
            if not a.no_real_gadgets:
                out["config"]["real_lc_step"] = side_light_client(m, ctx, sync)
                out["config"]["sync_committee_ssz"] = side_light_client(m, ctx, sync, flags=m.light_client.SYNC_COMMITTEE_ONLY)
                out["config"]["real_lc_step_recursive"] = side_light_client(m, ctx, sync, flags=m.light_client.BLS_PROOF_STAND_IN)
            if not a.no_synthetic:
                try:
                    syn, hw = synthetic_workload(m, ctx, sync, a.degree_bits, host_witness=True)
                    out["config"]["synthetic_plonky2_gate_set_2p%d" % a.degree_bits] = syn
                    if hw:
                        out["config"]["host_witness_2p%d" % a.degree_bits] = hw
                except Exception as e:
                    out["config"]["synthetic_plonky2_gate_set_2p%d" % a.degree_bits] = {"error": "%s: %s" % (type(e).__name__, e)}
                out["config"]["reference_gate_set_2p%d" % a.degree_bits] = reference_gate_set(m, ctx, a.degree_bits)
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(m, ctx, min(a.cpu_sample_bits, a.degree_bits), a.degree_bits)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    sharded, batch, rc = None, None, 0
    multi = (world > 1 or a.force_sharded)
    if multi and not a.no_batch:
        # BASELINE configs[4] as worded: a batch of 32 consecutive light-client updates data-parallel over the GPUs (32 / N per rank,
        # no data-path collective, the finished proofs gathered on rank 0).  The reference ships ONE consecutive pair, so every update
        # of the batch is that pair (the proofs are independent either way).
        try:
            prev, cur = m.light_client.reference_updates()
            step = m.light_client.LightClientStep(ctx, prev, cur, flags=0, extra_committees=a.extra_committees)
            step.prove()

            class Updates:
                num_updates = 32

                def __call__(self, u):
                    return u
            barrier()
            t0 = time.perf_counter()
            proofs = m.batch.prove_batch(lambda u: step.prove()[0], Updates(), rank, world, dist if world > 1 else None, dev)
            barrier()
            bt = m.batch.max_over_ranks(time.perf_counter() - t0, dist if world > 1 else None, dev)
            ok = None
            if rank == 0:
                pis = step.expected_public_inputs
                for p in (proofs[0], proofs[-1]):
                    step.verify(p, pis)
                ok = len(proofs) == 32
            step.close()
            batch = {"workload": "configs[4]: 32 light-client updates (each the reference's pair 633 -> 634 on the 2^%d-row circuit), %d per GPU over %d GPUs, "
                                 "proofs gathered on rank 0" % (info["degree_bits"], -(-32 // world), world),
                     "seconds": bt, "proofs_per_hour": 32 / bt * 3600.0, "all_32_proofs_gathered_first_and_last_verified": ok}
        except Exception as e:
            import traceback
            traceback.print_exc()
            batch = {"error": "%s: %s" % (type(e).__name__, e)}
            rc = 3 if a.strict_sharded else 0
    if multi and not a.no_sharded and (world & (world - 1)) == 0 and world <= 8:
        # A collective that never completes (this exchange has not run over RCCL with more than one rank yet) must not take the
        # replica result down with it: after LCP2_SHARDED_TIMEOUT_S the line goes out with the error and the rank exits non-zero.
        limit = float(os.environ.get("LCP2_SHARDED_TIMEOUT_S", "300")) + (0.0 if rank == 0 else 5.0)

        def work():
            params = m.standard_params(a.degree_bits, 4)
            circ, wires, pis = m.circuit.synthetic_circuit(params, seed=3, small_values=True)
            cs_dev = torch.from_numpy(circ.constants_sigmas.view(np.int64)).to(dev)
            m.circuit.tag_witness(wires, rank + 1)
            w_dev = torch.from_numpy(wires.view(np.int64)).to(dev)
            torch.cuda.synchronize()
            del wires
            return sharded_proof(m, ctx, circ, cs_dev.data_ptr(), w_dev, pis, rank, world, dist, dev)
        try:
            # a hang of this SIDE measurement is reported in the line like an exception of it: the replica result stands and the exit code
            # stays 0 (a non-zero rank would make the launcher - ours or torch.distributed.run - fail the whole run) unless --strict-sharded
            sharded = guarded(work, limit, lambda why: report({"error": "the sharded proof " + why}, batch, from_watchdog=True),
                              exit_code=3 if a.strict_sharded else 0)
        except Exception as e:  # the replica line still goes out and carries the error; --strict-sharded also fails the run
            import traceback
            traceback.print_exc()
            sharded = {"error": "%s: %s" % (type(e).__name__, e)}
            rc = 3 if a.strict_sharded else 0
        torch.cuda.empty_cache()

    report(sharded, batch)
    if world > 1 or a.force_sharded:
        try:
            dist.destroy_process_group()
        except Exception:  # after a failed collective the group may not shut down cleanly; the exit code already says so
            rc = rc or 3
    return rc


if __name__ == "__main__":
    sys.exit(main())
