"""GPU parity of the whole prover path (build -> prove -> verify) through the C ABI:
the proof produced on the MI355X must equal the oracle's proof word for word (caps, openings,
FRI commit-phase caps, final polynomial, minimum proof-of-work witness, query rounds), the product's
host verifier and the oracle's verifier must both accept it."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def _sections(m, params):
    """(name, start) pairs of the flat proof for readable mismatch reports"""
    import ctypes
    lib = m.load_library()
    total = lib.lcp2_proof_words(ctypes.byref(params))
    capw = 4 << params.cap_height
    names = [("wires_cap", 0), ("zs_cap", capw), ("quotient_cap", 2 * capw), ("openings", 3 * capw)]
    return names, total


def _first_mismatch(m, params, a, b):
    names, total = _sections(m, params)
    bad = np.nonzero(a != b)[0]
    if bad.size == 0:
        return None
    pos = int(bad[0])
    sec = [n for n, s in names if s <= pos][-1]
    return f"first mismatch at word {pos} of {total} (in or after section {sec}); {bad.size} words differ"


# 14-16: the LDE (2^17 .. 2^19 points) and the quotient iNTT run as two passes (strided + contiguous), the FRI schedule has
# 3 layers, the scans span several blocks: the multi-pass paths of the headline size inside a whole proof


def test_other_coset_shifts_take_the_general_path(gpu_ctx, oracle):
    """plonky2's coset shifts are k_j = 7^j and the permutation pass of K6 carries beta x k_j from wire to wire with a multiply by 7
    (QuotientArgs.kis_pow7); a description with other shifts (49^j here) must take the general multiply and still give the oracle's proof"""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(10, 4)
    for base in (49, 7):
        circ, wires, pis = m.circuit.synthetic_circuit(params, seed=77, coset_shift_base=base)
        assert int(circ.k_is[1]) == base
        oc = oracle_lib.OracleCircuit(oracle, circ)
        assert oc.check_witness(wires, pis)[0] == 0
        want = oc.prove(wires, pis)
        data = m.CircuitData.build(gpu_ctx, circ)
        got = data.prove(wires, pis)
        assert _first_mismatch(m, params, got, want) is None, (base, _first_mismatch(m, params, got, want))
        data.verify(got, pis)
        data.close()
        oc.close()


@pytest.mark.parametrize("degree_bits", [5, 6, 8, 10, 12, 13, 14, 15, 16])
def test_proof_equals_oracle(gpu_ctx, oracle, degree_bits):
    import eth_lc_plonky2_amd as m
    params = m.standard_params(degree_bits, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=100 + degree_bits)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    assert oc.check_witness(wires, pis)[0] == 0
    want = oc.prove(wires, pis)
    data = m.CircuitData.build(gpu_ctx, circ)
    d_gpu, cap_gpu = data.digest()
    d_orc, cap_orc = oc.digest()
    assert (cap_gpu == cap_orc).all() and (d_gpu == d_orc).all()
    got = data.prove(wires, pis)
    ch_g, ch_o = data.last_challenges(), oc.challenges()
    assert list(ch_g["betas"][:2]) == list(ch_o.betas)[:2], "betas differ: wires commitment mismatch"
    assert list(ch_g["alphas"][:2]) == list(ch_o.alphas)[:2], "alphas differ: Z / partial-product commitment mismatch"
    assert list(ch_g["zeta"]) == list(ch_o.zeta), "zeta differs: quotient commitment mismatch"
    assert list(ch_g["fri_alpha"]) == list(ch_o.fri_alpha), "fri alpha differs: openings mismatch"
    assert ch_g["pow_witness"] == ch_o.pow_witness, "proof-of-work witness is not the minimum"
    assert _first_mismatch(m, params, got, want) is None, _first_mismatch(m, params, got, want)
    data.verify(got, pis)
    assert oc.verify(got, pis) == 0
    bad = got.copy()
    bad[len(bad) // 3] ^= np.uint64(1)
    with pytest.raises(m.ProofRejected):
        data.verify(bad, pis)
    data.close()
    oc.close()


def test_proof_equals_oracle_at_2p18(gpu_ctx, oracle):
    """whole-proof parity at 2^18 rows (the sample bench.py's cpu_baseline leg proves on both sides: ~35 s of oracle time on the box's
    32 threads): three-pass quotient iNTT, four FRI layers, every kernel family at a size where a proof has 2^21 leaves"""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(18, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=1, small_values=True)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    want = oc.prove(wires, pis)
    data = m.CircuitData.build(gpu_ctx, circ)
    got = data.prove(wires, pis)
    assert _first_mismatch(m, params, got, want) is None, _first_mismatch(m, params, got, want)
    assert oc.verify(got, pis) == 0
    data.close()
    oc.close()


def test_light_client_step_in_process(gpu_ctx):
    """host/lc_capi.h through eth_lc_plonky2_amd.light_client: the reference's main() flow in this process (what bench.py times) - build
    the light-client circuit for updates 633 -> 634, data.prove(pw) with the witness generated inside, verify, public inputs = the
    natively computed cur_state / new_state; the SyncCommitteeSSZ gadget alone gives the reference's KAT root 0x27afac05..."""
    import eth_lc_plonky2_amd as m
    prev, cur = m.light_client.reference_updates()
    step = m.light_client.LightClientStep(gpu_ctx, prev, cur)
    assert step.info.degree_bits == 19 and step.info.num_public_inputs == 16
    proof, pis = step.prove()
    step.verify(proof, pis)
    assert (pis == step.expected_public_inputs).all()
    proof2, _ = step.prove()
    assert (proof2 == proof).all()  # deterministic (minimum proof-of-work witness)
    bad = proof.copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    with pytest.raises(m.ProofRejected):
        step.verify(bad, pis)
    step.close()
    ssz = m.light_client.LightClientStep(gpu_ctx, prev, cur, flags=m.light_client.SYNC_COMMITTEE_ONLY)
    proof, pis = ssz.prove()
    ssz.verify(proof, pis)
    root = b"".join(int(w).to_bytes(4, "big") for w in pis)
    assert root.hex() == "27afac05d6c340dd44d9b0515c667df438f4d5b3ac8fe7389649c6009eebaca9"  # src/sync_committee_pubkeys.rs:622
    ssz.close()


def test_host_generator_lanes_give_the_same_proof(gpu_ctx):
    """LCP2_HOST_LANES > 1 spreads the host generators of a witness generation (here: the recursive verifier's PoseidonGate chains of the
    light-client circuit with the stand-in inner proof) over threads; value slots are claimed by compare-and-swap (host/builder.cpp
    Values::set).  The proof must be the one a single lane gives, word for word, proof after proof."""
    import os
    import eth_lc_plonky2_amd as m
    prev, cur = m.light_client.reference_updates()
    proofs = {}
    old = os.environ.get("LCP2_HOST_LANES")
    try:
        for lanes in ("1", "4"):
            os.environ["LCP2_HOST_LANES"] = lanes  # read when the first proof of a circuit plans its lanes
            step = m.light_client.LightClientStep(gpu_ctx, prev, cur, flags=m.light_client.BLS_PROOF_STAND_IN)
            got = [step.prove() for _ in range(3)]
            for proof, pis in got:
                step.verify(proof, pis)
                assert (proof == got[0][0]).all() and (pis == step.expected_public_inputs).all()
            proofs[lanes] = got[0][0]
            step.close()
    finally:
        if old is None:
            os.environ.pop("LCP2_HOST_LANES", None)
        else:
            os.environ["LCP2_HOST_LANES"] = old
    assert (proofs["1"] == proofs["4"]).all()


def test_staged_host_witnesses(gpu_ctx, oracle):
    """lcp2_witness_stage / lcp2_prove_staged: two host witnesses of one circuit uploaded on the copy stream into the two staging slots
    (one from pinned, one from pageable memory) while proofs run; every proof equals the oracle's proof of the same witness; a slot
    that was not staged is refused"""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(12, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=77)
    other = m.circuit.tag_witness(wires.copy(), 5)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    want = [oc.prove(wires, pis), oc.prove(other, pis)]
    assert not (want[0] == want[1]).all()
    data = m.CircuitData.build(gpu_ctx, circ)
    with pytest.raises(m.Lcp2Error):
        data.prove_staged(0, pis)
    lib = gpu_ctx.lib
    import ctypes
    assert lib.lcp2_host_register(gpu_ctx.handle, wires.ctypes.data_as(ctypes.c_void_p), wires.nbytes) == 0
    try:
        data.stage_witness(wires, 0)
        for i in range(4):
            nxt = (i + 1) % 2
            data.stage_witness(other if nxt else wires, nxt)   # the next witness goes up while this proof runs
            got = data.prove_staged(i % 2, pis)
            assert (got == want[i % 2]).all(), i
        got = data.prove_staged(0, pis)  # staged in the last iteration
        assert (got == want[0]).all()
        with pytest.raises(m.Lcp2Error):
            data.prove_staged(0, pis)    # a slot is consumed by its proof
    finally:
        gpu_ctx.sync()
        lib.lcp2_host_unregister(gpu_ctx.handle, wires.ctypes.data_as(ctypes.c_void_p))
    data.close()
    oc.close()


def test_workspace_reuse_and_device_resident_witness(gpu_ctx, oracle):
    import torch
    import eth_lc_plonky2_amd as m
    params = m.standard_params(9, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=7)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    data = m.CircuitData.build(gpu_ctx, circ)
    want = oc.prove(wires, pis)
    assert (data.prove(wires, pis) == want).all()
    # second witness on the same circuit: the free cells of the NoopGate padding row
    w2 = m.circuit.tag_witness(wires.copy(), 0xC0FFEE)
    assert oc.check_witness(w2, pis)[0] == 0
    want2 = oc.prove(w2, pis)
    assert (want2 != want).any()
    t = torch.from_numpy(w2.view(np.int64)).cuda()
    torch.cuda.synchronize()
    got2 = data.prove(t.data_ptr(), pis, mem=m.MEM_DEVICE)
    assert (got2 == want2).all()
    assert (data.prove(wires, pis) == want).all()  # and back again: no state leaks between proofs
    data.close()
    oc.close()


@pytest.mark.parametrize("device_witness", [False, True])
def test_non_canonical_witness_proves_like_its_canonical_form(gpu_ctx, oracle, device_witness):
    """lcp2.h accepts any u64 as a field element.  A witness with p added to a few hundred cells (values in [p, 2^64)) - gate
    outputs, routed cells, PoseidonGate state - proves to the SAME proof as its canonical form, from host and from device memory
    (there the library keeps the caller's pointer: the iNTT's bit-reversal reports the non-canonical values and the witness check
    and K5 continue from a canonical copy); with a real violation added it is still LCP2_E_UNSAT, not a false accept."""
    import torch
    import eth_lc_plonky2_amd as m
    params = m.standard_params(8, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=21, small_values=True)
    data = m.CircuitData.build(gpu_ctx, circ)
    want = data.prove(wires, pis)
    rng = np.random.default_rng(5)
    P = np.uint64(m.GOLDILOCKS_P)
    lifted = wires.copy()
    room = lifted < np.uint64(2 ** 32 - 1)          # x + p < 2^64  <=>  x < 2^32 - 1
    idx = np.argwhere(room)
    for r, c in idx[rng.choice(len(idx), size=600, replace=False)]:
        lifted[r, c] += P
    assert (lifted >= P).sum() == 600 and ((lifted % P) == wires).all()

    def prove(w):
        if not device_witness:
            return data.prove(w, pis)
        t = torch.from_numpy(np.ascontiguousarray(w).view(np.int64)).cuda()
        torch.cuda.synchronize()
        out = data.prove(t.data_ptr(), pis, mem=m.MEM_DEVICE)
        assert (t.cpu().numpy().view(np.uint64) == w).all()  # the caller's buffer is not written
        return out

    assert (prove(lifted) == want).all()
    assert (prove(wires) == want).all()
    arith = int(np.nonzero(circ.constants_sigmas[0] == circ.gateset.index("ArithmeticGate"))[0][2])
    bad = lifted.copy()
    bad[7, arith] = (bad[7, arith] % P) ^ np.uint64(1)
    with pytest.raises(m.Lcp2Error) as e:
        prove(bad)
    assert e.value.status == -5
    assert (prove(lifted) == want).all()  # and the handle is fine afterwards
    data.close()


def test_unsatisfied_witness_is_an_error(gpu_ctx, oracle):
    """prove() of an unsatisfiable witness is an Err in plonky2 (the reference's #[should_panic] tests): lcp2_prove returns
    LCP2_E_UNSAT for a violated gate constraint (checked over the rows of H on the device) and for a broken copy constraint
    (the permutation product does not return to 1); the same witness unbroken proves."""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(7, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=8)
    data = m.CircuitData.build(gpu_ctx, circ)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    arith = int(np.nonzero(circ.constants_sigmas[0] == circ.gateset.index("ArithmeticGate"))[0][3])
    bad = wires.copy()
    bad[7, arith] ^= np.uint64(1)          # an ArithmeticGate output
    assert oc.check_witness(bad, pis)[0] > 0
    with pytest.raises(m.Lcp2Error) as e:
        data.prove(bad, pis)
    assert e.value.status == -5 and "gate constraint" in str(e.value) and ("row %d" % arith) in str(e.value)
    # a copy constraint: the first arithmetic row's copy of public input 0 changed together with the product it feeds, so
    # that every gate constraint still holds
    first = int(np.nonzero(circ.constants_sigmas[0] == circ.gateset.index("ArithmeticGate"))[0][0])
    bad = wires.copy()
    bad[1, first] = np.uint64(12345)
    c0, c1 = (int(circ.constants_sigmas[2 + k, first]) for k in (0, 1))
    P = m.GOLDILOCKS_P
    bad[3, first] = np.uint64((c0 * int(bad[0, first]) * 12345 + c1 * int(bad[2, first])) % P)
    second = int(np.nonzero(circ.constants_sigmas[0] == circ.gateset.index("ArithmeticGate"))[0][1])
    c0s, c1s = (int(circ.constants_sigmas[2 + k, second]) for k in (0, 1))
    bad[0, second] = bad[3, first]
    bad[3, second] = np.uint64((c0s * int(bad[0, second]) * int(bad[1, second]) + c1s * int(bad[2, second])) % P)
    assert oc.check_witness(bad, pis)[0] == 0  # gate constraints hold
    with pytest.raises(m.Lcp2Error) as e:
        data.prove(bad, pis)
    assert e.value.status == -5 and "copy constraint" in str(e.value)
    got = data.prove(wires, pis)           # the handle is still usable
    assert (got == oc.prove(wires, pis)).all()
    data.close()
    oc.close()


def test_native_gate_claim_is_checked_at_build(gpu_ctx, oracle):
    """LCP2_GATE_NATIVE_POSEIDON is a claim of the caller: build() runs the gate's program and the native evaluator on random
    points and refuses a program that is not plonky2's PoseidonGate (here: one round constant off, one wire index off); with
    the flag cleared the same description is interpreted, and both forms give the oracle's proof."""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(6, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=21)
    gs = circ.gateset
    pg = gs.gates[gs.index("PoseidonGate")]
    assert pg.flags & m.circuit.GATE_NATIVE_POSEIDON
    want = oracle_lib.OracleCircuit(oracle, circ).prove(wires, pis)
    data = m.CircuitData.build(gpu_ctx, circ)
    assert (data.prove(wires, pis) == want).all()
    data.close()
    # the same gate interpreted instruction by instruction
    pg.flags &= ~m.circuit.GATE_NATIVE_POSEIDON
    data = m.CircuitData.build(gpu_ctx, circ)
    assert (data.prove(wires, pis) == want).all()
    data.close()
    pg.flags |= m.circuit.GATE_NATIVE_POSEIDON
    # a program that is not PoseidonGate must not get the native path
    k = int(np.nonzero(gs.imm == np.uint64(m.poseidon_py.round_constants()[200]))[0][0])
    gs.imm[k] ^= np.uint64(1)
    with pytest.raises(m.Lcp2Error) as e:
        m.CircuitData.build(gpu_ctx, circ)
    assert e.value.status == -1 and "NATIVE" in str(e.value)
    gs.imm[k] ^= np.uint64(1)
    code = gs.code
    pc = pg.code_offset + pg.code_len - 2   # the last SUB: state[11] - wire_output(11)
    assert code[2 * pc] & 0xF == m.circuit.OP_SUB
    code[2 * pc + 1] ^= np.uint32(1 << 16)  # compare with a neighbouring wire instead
    with pytest.raises(m.Lcp2Error):
        m.CircuitData.build(gpu_ctx, circ)
    code[2 * pc + 1] ^= np.uint32(1 << 16)
    m.CircuitData.build(gpu_ctx, circ).close()


@pytest.mark.parametrize("round_", range(8))
def test_parity_soak_seeds(gpu_ctx, oracle, round_):
    """tests/checks/parity_soak.py folded into the suite: further seeds and sizes, random and small-valued witnesses, and a second
    proof on the same handle (workspace reuse)."""
    import eth_lc_plonky2_amd as m
    db = 5 + (round_ * 3) % 9
    params = m.standard_params(db, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=7000 + round_, small_values=bool(round_ & 1))
    oc = oracle_lib.OracleCircuit(oracle, circ)
    want = oc.prove(wires, pis)
    data = m.CircuitData.build(gpu_ctx, circ)
    for rep in range(2):
        got = data.prove(wires, pis)
        assert _first_mismatch(m, params, got, want) is None, "rep %d: %s" % (rep, _first_mismatch(m, params, got, want))
    data.close()
    oc.close()


def test_headline_size_proof_verifies(gpu_ctx):
    """What bench.py times, as a test: build -> prove -> verify at n = 2^22 rows x 135 wires (BASELINE configs[2]); the proof is
    accepted, a flipped public input and a flipped opening are rejected, and a second proof of the same witness is identical
    (no state leaks through the 92 GB workspace)."""
    import torch
    import eth_lc_plonky2_amd as m
    params = m.standard_params(22, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=3, small_values=True)
    dev = torch.device("cuda", 0)
    cs_dev = torch.from_numpy(circ.constants_sigmas.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    data = m.CircuitData.build(gpu_ctx, circ, constants_sigmas_ptr=cs_dev.data_ptr(), mem=m.MEM_DEVICE)
    del cs_dev
    w_dev = torch.from_numpy(wires.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    del wires
    proof = data.prove(w_dev.data_ptr(), pis, mem=m.MEM_DEVICE)
    data.verify(proof, pis)
    again = data.prove(w_dev.data_ptr(), pis, mem=m.MEM_DEVICE)
    assert (again == proof).all()
    wrong = pis.copy()
    wrong[1] ^= np.uint64(1)
    with pytest.raises(m.ProofRejected):
        data.verify(proof, wrong)
    bad = proof.copy()
    bad[3 * (4 << params.cap_height) + 7] ^= np.uint64(1)  # an opening
    with pytest.raises(m.ProofRejected):
        data.verify(bad, pis)
    # the independent check at headline size (the reference ends with stock plonky2's data.verify, src/main.rs:233): the ORACLE's
    # verifier, built from the digest and the constants/sigmas cap alone (no oracle build(), which would take minutes here),
    # accepts the GPU proof and rejects it after a one-word change
    import oracle_lib
    digest, cap = data.digest()
    ov = oracle_lib.OracleCircuit.verifier_only(oracle_lib.load(), circ, digest, cap)
    assert ov.verify(proof, pis) == 0
    assert ov.verify(bad, pis) != 0          # an opening: the vanishing identity
    assert ov.verify(proof, wrong) != 0      # a public input: the transcript
    lay = m.proof_layout(params)
    bad2 = proof.copy()
    bad2[lay.queries + lay.q_init_off[1] + 2] ^= np.uint64(1)  # a wires leaf word of the first query: the Merkle path
    assert ov.verify(bad2, pis) != 0
    ov.close()
    data.close()
    del w_dev
    torch.cuda.empty_cache()


def test_large_proof_verifies(gpu_ctx):
    # size-independent property at a size the oracle would take minutes for: a 2^18-row proof verifies
    import eth_lc_plonky2_amd as m
    params = m.standard_params(18, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=18, small_values=True)
    data = m.CircuitData.build(gpu_ctx, circ)
    proof = data.prove(wires, pis)
    data.verify(proof, pis)
    wrong = pis.copy()
    wrong[0] ^= np.uint64(1)
    with pytest.raises(m.ProofRejected):
        data.verify(proof, wrong)
    data.close()


class _Challenger:
    """Challenger<F, PoseidonHash> on the oracle's permutation: the transcript a plonky2 fork keeps on its own side
    when it binds the seams one by one."""
    P = (1 << 64) - (1 << 32) + 1

    def __init__(self, oracle):
        self.o, self.s, self.inp, self.out = oracle, np.zeros(12, dtype=np.uint64), [], []

    def observe(self, xs):
        for x in np.asarray(xs, dtype=np.uint64).ravel():
            self.out = []
            self.inp.append(int(x) % self.P)
            if len(self.inp) == 8:
                self._duplex()

    def _duplex(self):
        for i, v in enumerate(self.inp):
            self.s[i] = v
        self.inp = []
        self.o.orc_poseidon_permute(oracle_lib.vp(self.s))
        self.out = [int(v) for v in self.s[:8]]

    def get(self, k=1):
        r = []
        for _ in range(k):
            if self.inp or not self.out:
                self._duplex()
            r.append(self.out.pop())
        return np.array(r, dtype=np.uint64)

    def state(self, m):
        st = m.binding.ChallengerState()
        for i in range(12):
            st.sponge[i] = int(self.s[i])
        for i, v in enumerate(self.inp):
            st.input[i] = v
        for i, v in enumerate(self.out):
            st.output[i] = v
        st.input_len, st.output_len = len(self.inp), len(self.out)
        return st


def test_staged_seams_compose_to_prove(gpu_ctx, oracle):
    """lcp2_commit_wires -> lcp2_perm_zs -> lcp2_quotient -> lcp2_fri_open, with the transcript run by the
    caller, give the proof lcp2_prove gives; calling a seam before its predecessor is LCP2_E_INVALID."""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(10, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=77)
    data = m.CircuitData.build(gpu_ctx, circ)
    want = data.prove(wires, pis)
    ch = data.last_challenges()
    with pytest.raises(m.Lcp2Error):  # nothing committed yet on a fresh handle
        m.CircuitData.build(gpu_ctx, circ).perm_zs([1, 2], [3, 4])

    capw = 4 << params.cap_height
    proof = np.zeros(data.proof_words, dtype=np.uint64)
    digest, _ = data.digest()
    pi_hash = np.zeros(4, dtype=np.uint64)
    p = np.asarray(pis, dtype=np.uint64)
    oracle.orc_hash_no_pad(oracle_lib.vp(p), len(p), oracle_lib.vp(pi_hash))
    t = _Challenger(oracle)
    t.observe(digest)
    t.observe(pi_hash)
    proof[0:capw] = data.commit_wires(wires).ravel()
    t.observe(proof[0:capw])
    betas, gammas = t.get(2), t.get(2)
    assert list(betas) == list(ch["betas"][:2]) and list(gammas) == list(ch["gammas"][:2])
    proof[capw:2 * capw] = data.perm_zs(betas, gammas).ravel()
    t.observe(proof[capw:2 * capw])
    alphas = t.get(2)
    proof[2 * capw:3 * capw] = data.quotient(alphas, pi_hash).ravel()
    t.observe(proof[2 * capw:3 * capw])
    zeta = t.get(2)
    assert list(zeta) == list(ch["zeta"])
    st = t.state(m)
    data.fri_open(zeta, st, proof)
    assert _first_mismatch(m, params, proof, want) is None, _first_mismatch(m, params, proof, want)
    data.verify(proof, pis)
    # the returned transcript state is the one after the last query index was drawn: its next output differs from the
    # state that went in, and the seam is deterministic
    st2 = t.state(m)
    proof2 = np.zeros_like(proof)
    proof2[:3 * capw] = proof[:3 * capw]
    data.fri_open(zeta, st2, proof2)
    assert (proof2 == proof).all() and list(st2.sponge) == list(st.sponge) and list(st.sponge) != [int(v) for v in t.s]
    # the same stage phase by phase (the exchange points of a sharded proof): identical words, identical final state;
    # a phase out of order is refused
    st3 = t.state(m)
    proof3 = np.zeros_like(proof)
    proof3[:3 * capw] = proof[:3 * capw]
    with pytest.raises(m.Lcp2Error):
        data.fri_open_commit(proof3)
    data.fri_open_begin(zeta, st3, proof3)
    lo, cnt = data.proof_section(m.binding.SECTION_OPENINGS)
    assert (proof3[lo:lo + cnt] == proof[lo:lo + cnt]).all() and (proof3[lo + cnt:] == 0).all()
    with pytest.raises(m.Lcp2Error):
        data.fri_open_finish(proof3)
    data.fri_open_commit(proof3)
    lo, cnt = data.proof_section(m.binding.SECTION_FRI_CAP0)
    assert cnt == capw and (proof3[lo:lo + cnt] == proof[lo:lo + cnt]).all()
    data.fri_open_finish(proof3, st3)
    assert (proof3 == proof).all() and list(st3.sponge) == list(st.sponge)
    assert data.proof_section(m.binding.SECTION_AFTER_CAPS) == (3 * capw, data.proof_words - 3 * capw)
    data.close()


def _variant(m, degree_bits, **kw):
    p = m.standard_params(degree_bits, 4)
    arities = kw.pop("fri_arity_bits", None)
    for k, v in kw.items():
        setattr(p, k, v)
    if arities is not None:
        p.num_fri_layers = len(arities)
        for i in range(8):
            p.fri_arity_bits[i] = arities[i] if i < len(arities) else 0
    return p


@pytest.mark.parametrize("name,degree_bits,kw", [
    ("cap_height_0", 8, dict(cap_height=0)),
    ("cap_height_1_one_challenge", 9, dict(cap_height=1, num_challenges=1)),
    ("cap_at_fri_limit", 7, dict(cap_height=5, fri_arity_bits=[3, 2])),
    ("mixed_arities", 10, dict(fri_arity_bits=[3, 1, 2, 2], num_query_rounds=9, proof_of_work_bits=7)),
    ("arity_32_no_layers_after", 9, dict(fri_arity_bits=[5], num_query_rounds=5)),
    ("no_fri_layers", 5, dict(fri_arity_bits=[], num_query_rounds=3, proof_of_work_bits=3)),
    ("rate_bits_4", 8, dict(rate_bits=4, quotient_degree_factor=16, fri_arity_bits=[4, 2])),
    ("wide_trace", 8, dict(num_wires=150)),
    ("max_queries", 8, dict(num_query_rounds=64)),
])
def test_nonstandard_params_equal_oracle(gpu_ctx, oracle, name, degree_bits, kw):
    """n, W, rate_bits, cap_height, quotient_degree_factor, the FRI schedule, query count and PoW bits are run-time
    parameters of every kernel: the GPU proof equals the oracle's for each variant.  (rate_bits < 3 is not in
    the list: the synthetic gate set's selector groups are laid out for constraint degree 9 = quotient_degree_factor 8.)"""
    import eth_lc_plonky2_amd as m
    params = _variant(m, degree_bits, **dict(kw))
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=500 + degree_bits)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    assert oc.check_witness(wires, pis)[0] == 0
    want = oc.prove(wires, pis)
    assert oc.verify(want, pis) == 0
    data = m.CircuitData.build(gpu_ctx, circ)
    got = data.prove(wires, pis)
    assert _first_mismatch(m, params, got, want) is None, name + ": " + _first_mismatch(m, params, got, want)
    data.verify(got, pis)
    data.close()
    oc.close()


@pytest.mark.parametrize("degree_bits,native", [(5, True), (8, True), (8, False), (11, True)])
def test_recursion_gate_programs_prove_on_the_gpu(gpu_ctx, oracle, degree_bits, native):
    """a circuit of plonky2's recursion gates (recursion_gates.py: extension arithmetic, the reducing gates, RandomAccessGate,
    ExponentiationGate, PoseidonMdsGate; native: the generated straight-line evaluators, whose claims build() checks against the
    programs, else interpreted by K6): the GPU proof equals the oracle's word for word and verifies"""
    import eth_lc_plonky2_amd as m
    from eth_lc_plonky2_amd import recursion_gates as rg
    params = m.standard_params(degree_bits, 4)
    circ, wires, pis = rg.recursion_gates_circuit(params, seed=40 + degree_bits, native=native)
    assert all(bool(g.flags & 0x8000) == (native and g.num_constraints > 0) for g in circ.gateset.gates)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    want = oc.prove(wires, pis)
    data = m.CircuitData.build(gpu_ctx, circ)
    got = data.prove(wires, pis)
    assert _first_mismatch(m, params, got, want) is None, _first_mismatch(m, params, got, want)
    data.verify(got, pis)
    w2 = wires.copy()
    w2[67, 3] ^= np.uint64(1)  # the output of the ExponentiationGate on row 3
    with pytest.raises(m.Lcp2Error) as e:
        data.prove(w2, pis)
    assert e.value.status == -5  # LCP2_E_UNSAT
    data.close()
    oc.close()


@pytest.mark.parametrize("degree_bits,native", [(6, True), (9, True), (9, False)])
def test_reference_gate_programs_prove_on_the_gpu(gpu_ctx, oracle, degree_bits, native):
    """a circuit of the gates the reference's own circuit is made of (u32_gates.py: plonky2_u32's U32 arithmetic / add-many /
    subtraction / range-check gates, ComparisonGate, plonky2's CosetInterpolationGate - degree-4 and degree-8 constraints, three
    selector groups; native: generated straight-line evaluators, else interpreted by K6): the GPU proof equals the oracle's word
    for word and verifies; a broken row is LCP2_E_UNSAT"""
    import eth_lc_plonky2_amd as m
    from eth_lc_plonky2_amd import u32_gates as ug
    params = m.standard_params(degree_bits, 5)
    circ, wires, pis = ug.reference_gates_circuit(params, seed=60 + degree_bits, native=native)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    want = oc.prove(wires, pis)
    data = m.CircuitData.build(gpu_ctx, circ)
    got = data.prove(wires, pis)
    assert _first_mismatch(m, params, got, want) is None, _first_mismatch(m, params, got, want)
    data.verify(got, pis)
    assert oc.verify(got, pis) == 0
    kinds = list(ug.ROW_GENERATORS)
    for kind, wire in (("U32ArithmeticGate", 3), ("ComparisonGate", 2), ("CosetInterpolationGate", 35), ("U32SubtractionGate", 4)):
        w2 = wires.copy()
        w2[wire, kinds.index(kind)] ^= np.uint64(1)
        with pytest.raises(m.Lcp2Error) as e:
            data.prove(w2, pis)
        assert e.value.status == -5, kind  # LCP2_E_UNSAT
    data.close()
    oc.close()


@pytest.mark.parametrize("degree_bits", [10, 14])
def test_reference_gate_mix_proof_equals_oracle(gpu_ctx, oracle, degree_bits):
    """The gate mix of a circuit built from the reference's own gadgets (u32_gates.ReferenceMix: U32AddMany / U32Arithmetic /
    U32RangeCheck / U32Subtraction / Comparison rows next to BaseSum, Arithmetic, Constant, PublicInput and PoseidonGate rows, real
    copy constraints, the public inputs hashed in-circuit; 11 gate types in three selector groups) - what `bench.py` times at 2^22 rows
    as config.reference_gate_set_2p22.  With the generated native evaluators and interpreted the GPU gives the same proof, and it is
    the oracle's proof word for word; both verifiers accept it; a broken u32 row is LCP2_E_UNSAT."""
    import eth_lc_plonky2_amd as m
    from eth_lc_plonky2_amd import u32_gates as ug
    params = m.standard_params(degree_bits, 5)
    circ, wires, pis = ug.reference_mix_circuit(params, seed=300 + degree_bits)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    assert oc.check_witness(wires, pis)[0] == 0
    want = oc.prove(wires, pis)
    data = m.CircuitData.build(gpu_ctx, circ)
    d_gpu, cap_gpu = data.digest()
    d_orc, cap_orc = oc.digest()
    assert (cap_gpu == cap_orc).all() and (d_gpu == d_orc).all()
    got = data.prove(wires, pis)
    assert _first_mismatch(m, params, got, want) is None, _first_mismatch(m, params, got, want)
    data.verify(got, pis)
    assert oc.verify(got, pis) == 0
    gs = circ.gateset
    rows = {name: int(np.nonzero(circ.constants_sigmas[gs.gates[gs.index(name)].selector_index] == np.uint64(gs.index(name)))[0][3])
            for name in ("U32AddManyGate", "U32ArithmeticGate", "U32RangeCheckGate", "U32SubtractionGate", "ComparisonGate")}
    for name, wire in (("U32AddManyGate", 4), ("U32ArithmeticGate", 3), ("U32RangeCheckGate", 7 + 16 * 3 + 2), ("U32SubtractionGate", 3), ("ComparisonGate", 2)):
        w2 = wires.copy()
        w2[wire, rows[name]] ^= np.uint64(1)
        with pytest.raises(m.Lcp2Error) as e:
            data.prove(w2, pis)
        assert e.value.status == -5, name  # LCP2_E_UNSAT
    data.close()
    circ_i, wires_i, pis_i = ug.reference_mix_circuit(params, seed=300 + degree_bits, native=False)
    assert (wires_i == wires).all() and not any(g.flags & 0x8000 for g in circ_i.gateset.gates)
    data = m.CircuitData.build(gpu_ctx, circ_i)
    assert (data.prove(wires_i, pis_i) == want).all(), "the interpreted form gives another proof"
    data.close()
    oc.close()


def test_generated_gate_claim_is_checked_at_build(gpu_ctx):
    """LCP2_GATE_NATIVE_GENERATED(k) is a claim like the plonky2-gate ones: a program that is not program k of csrc/generated_gates*.hpp
    (one immediate changed, one wire index changed, or the claim of another gate) is refused by build(); unclaimed it is interpreted."""
    import eth_lc_plonky2_amd as m
    from eth_lc_plonky2_amd import u32_gates as ug
    params = m.standard_params(6, 5)
    circ, wires, pis = ug.reference_gates_circuit(params, seed=3)
    gs = circ.gateset
    m.CircuitData.build(gpu_ctx, circ).close()
    g = gs.gates[gs.index("U32SubtractionGate")]
    pc = g.code_offset + 3     # SUB t, t, borrow (wire 2 of operation 0)
    assert gs.code[2 * pc] & 0xF == m.circuit.OP_SUB
    gs.code[2 * pc + 1] ^= np.uint32(1 << 16)
    with pytest.raises(m.Lcp2Error) as e:
        m.CircuitData.build(gpu_ctx, circ)
    assert e.value.status == -1 and "NATIVE" in str(e.value)
    gs.code[2 * pc + 1] ^= np.uint32(1 << 16)
    claim = g.flags
    g.flags = (g.flags & ~m.circuit.GATE_NATIVE_MASK) | m.circuit.gate_native_generated(m.circuit.GENERATED_GATE_INDEX["U32RangeCheckGate"])
    with pytest.raises(m.Lcp2Error):
        m.CircuitData.build(gpu_ctx, circ)
    g.flags = claim & ~m.circuit.GATE_NATIVE_MASK   # unclaimed: interpreted
    m.CircuitData.build(gpu_ctx, circ).close()
    g.flags = claim
    k = int(np.nonzero(gs.imm == np.uint64(1 << 32))[0][0])
    gs.imm[k] ^= np.uint64(2)
    with pytest.raises(m.Lcp2Error):
        m.CircuitData.build(gpu_ctx, circ)
