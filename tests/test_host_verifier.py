"""Host logic without a GPU: the product's verifier (lcp2_verify, host-only C++) must accept proofs
made by the oracle prover and reject tampered ones; the proof layout of both sides must agree."""
import numpy as np
import pytest

import oracle_lib


@pytest.mark.parametrize("degree_bits", [5, 6, 9])
def test_product_verifier_accepts_oracle_proofs(oracle, degree_bits):
    import eth_lc_plonky2_amd as m
    params = m.standard_params(degree_bits, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=degree_bits)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    assert oc.check_witness(wires, pis)[0] == 0
    proof = oc.prove(wires, pis)
    assert oc.verify(proof, pis) == 0
    digest, cap = oc.digest()
    vd = m.CircuitData.verifier_only(circ, digest, cap)
    assert vd.proof_words == oc.proof_words
    vd.verify(proof, pis)
    for pos in (0, 70, vd.proof_words // 2, vd.proof_words - 1, vd.proof_words - 20):
        bad = proof.copy()
        bad[pos] ^= np.uint64(1)
        with pytest.raises(m.ProofRejected) as e:
            vd.verify(bad, pis)
        assert e.value.check == oc.verify(bad, pis) != 0
    bad_pis = pis.copy()
    bad_pis[1] ^= np.uint64(2)
    with pytest.raises(m.ProofRejected):
        vd.verify(proof, bad_pis)
    with pytest.raises(m.Lcp2Error) as e:
        vd.prove(wires, pis)  # verifier-only circuits have no device: no CPU proving path exists
    assert e.value.status == -2
    oc.close()
    vd.close()


def test_oracle_verifier_only_circuit_agrees_with_the_built_one(oracle):
    """orc_verifier_new: the oracle's verifier from digest + cap alone (what the GPU tests use at 2^19 / 2^22 rows, where the
    oracle's build() would take minutes) gives the same verdict, check for check, as the oracle circuit that built the
    commitment itself; it refuses to prove."""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(7, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=77)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    proof = oc.prove(wires, pis)
    digest, cap = oc.digest()
    ov = oracle_lib.OracleCircuit.verifier_only(oracle, circ, digest, cap)
    assert ov.verify(proof, pis) == 0
    lay = m.proof_layout(params)
    spots = [0, lay.zs_cap + 1, lay.quot_cap + 2, lay.op_wires + 3, lay.op_quotient + 1, lay.fri_caps + 5, lay.final_poly, lay.pow_witness,
             lay.queries + lay.q_init_off[0] + 1, lay.queries + lay.q_init_off[1] + lay.q_init_cols[1] + 2, lay.queries + lay.q_step_off[0] + 3,
             lay.queries + 5 * lay.query_words + lay.q_init_off[3]]
    for pos in spots:
        bad = proof.copy()
        bad[pos] ^= np.uint64(1)
        assert ov.verify(bad, pis) == oc.verify(bad, pis) != 0, pos
    wrong_digest = digest.copy()
    wrong_digest[2] ^= np.uint64(1)
    ov2 = oracle_lib.OracleCircuit.verifier_only(oracle, circ, wrong_digest, cap)
    assert ov2.verify(proof, pis) != 0
    wrong_cap = cap.copy()
    wrong_cap[3, 1] ^= np.uint64(1)
    ov3 = oracle_lib.OracleCircuit.verifier_only(oracle, circ, digest, wrong_cap)
    assert ov3.verify(proof, pis) != 0  # the constants/sigmas Merkle paths no longer end in the cap (unless no query lands under entry 3)
    out = np.zeros(ov.proof_words, dtype=np.uint64)
    assert oracle.orc_prove(ov.h, oracle_lib.vp(wires), oracle_lib.vp(pis), oracle_lib.vp(out)) != 0
    for x in (oc, ov, ov2, ov3):
        x.close()


def test_unsatisfied_witness_is_rejected(oracle):
    import eth_lc_plonky2_amd as m
    params = m.standard_params(6, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=1)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    wires[7, 11] ^= np.uint64(1)  # break one arithmetic-gate input
    assert oc.check_witness(wires, pis)[0] > 0
    proof = oc.prove(wires, pis)
    vd = m.CircuitData.verifier_only(circ, *oc.digest())
    with pytest.raises(m.ProofRejected) as e:
        vd.verify(proof, pis)
    assert e.value.check == 3  # vanishing polynomial identity


def test_small_value_witness(oracle):
    import eth_lc_plonky2_amd as m
    params = m.standard_params(7, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=3, small_values=True)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    assert oc.check_witness(wires, pis)[0] == 0
    assert oc.verify(oc.prove(wires, pis), pis) == 0


def test_verifier_create_validates_gate_programs(oracle):
    """lcp2_verifier_create checks every instruction (opcode, register, operand ranges) like lcp2_circuit_create does,
    so that a hostile circuit description cannot make the host interpreter index out of range."""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(5, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=3)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    digest, cap = oc.digest()
    code = circ.gateset.code
    first = next(g for g in circ.gateset.gates if g.code_len).code_offset
    saved = (int(code[2 * first]), int(code[2 * first + 1]))
    try:
        for w0, w1 in [((saved[0] & ~0xF) | 15, saved[1]),                   # unknown opcode
                       ((saved[0] & ~0xFF00) | (200 << 8), saved[1]),         # destination register out of range
                       ((saved[0] & ~0xF0000) | (1 << 16), 0xFFFF)]:          # wire operand out of range
            code[2 * first], code[2 * first + 1] = w0, w1
            with pytest.raises(m.Lcp2Error) as e:
                m.CircuitData.verifier_only(circ, digest, cap)
            assert e.value.status == -1
    finally:
        code[2 * first], code[2 * first + 1] = saved
    m.CircuitData.verifier_only(circ, digest, cap).close()
    oc.close()


def _variant(m, base_bits, **kw):
    p = m.standard_params(base_bits, 4)
    for k, v in kw.items():
        if k == "fri_arity_bits":
            p.num_fri_layers = len(v)
            for i in range(8):
                p.fri_arity_bits[i] = v[i] if i < len(v) else 0
        else:
            setattr(p, k, v)
    return p


@pytest.mark.parametrize("kw", [
    dict(fri_arity_bits=[6]),            # arity 64: the verifier's interpolation arrays hold 32 points
    dict(cap_height=9),                  # above the LDE tree of a 2^5-row circuit: sibling counts would underflow
    dict(quotient_degree_factor=0),
    dict(proof_of_work_bits=0),          # a shift by 64
    dict(degree_bits=40),
    dict(num_query_rounds=65),
    dict(num_challenges=3),
    dict(num_routed_wires=200),
])
def test_verifier_create_checks_the_parameters(oracle, kw):
    """lcp2_verifier_create runs the same shape checks as lcp2_circuit_create: the verifier indexes fixed-size arrays with
    them and trusts nothing else about an untrusted description."""
    import eth_lc_plonky2_amd as m
    params = m.standard_params(5, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=3)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    digest, cap = oc.digest()
    oc.close()
    good = circ.params
    try:
        circ.params = _variant(m, 5, **kw)
        with pytest.raises(m.Lcp2Error) as e:
            m.CircuitData.verifier_only(circ, digest, cap)
        assert e.value.status in (-1, -6)
    finally:
        circ.params = good


@pytest.mark.parametrize("arities", [[31], [4, 4, 4], [5, 5], [0], [4, 4]])
def test_layout_entry_points_refuse_a_fri_schedule_that_does_not_fit(arities):
    """ProofLayout subtracts the arities from the LDE height in unsigned arithmetic: a single arity of 31, or a schedule whose sum
    exceeds degree_bits (7 here), would underflow into an undefined shift and a walk over ~2^32 sibling words.  Every entry point
    that computes a layout runs the full shape check of build() first (no abort, no hang across the ABI)."""
    import ctypes
    import eth_lc_plonky2_amd as m
    lib = m.load_library()
    p = _variant(m, 7, fri_arity_bits=arities)
    assert lib.lcp2_proof_words(ctypes.byref(p)) == 0
    with pytest.raises(m.Lcp2Error):
        m.proof_layout(p)
    lib.lcp2_proof_bytes.restype = ctypes.c_size_t
    assert lib.lcp2_proof_bytes(ctypes.byref(p), ctypes.c_size_t(4), ctypes.c_uint32(1)) == 0
    with pytest.raises(m.Lcp2Error):
        m.proof_to_bytes(p, np.zeros(16, dtype=np.uint64), np.zeros(4, dtype=np.uint64))
    with pytest.raises(m.Lcp2Error):
        m.proof_from_bytes(p, b"\0" * 64, 4)
    # a schedule that does fit still works
    ok = _variant(m, 7, fri_arity_bits=[2, 1])
    assert lib.lcp2_proof_words(ctypes.byref(ok)) == m.proof_layout(ok).total > 0


def test_verify_refuses_a_proof_of_the_wrong_length(oracle):
    import eth_lc_plonky2_amd as m
    params = m.standard_params(5, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=4)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    proof = oc.prove(wires, pis)
    vd = m.CircuitData.verifier_only(circ, *oc.digest())
    vd.verify(proof, pis)
    for bad in (proof[:-1], proof[:100], np.concatenate([proof, proof[:1]])):
        with pytest.raises(m.Lcp2Error) as e:
            vd.verify(bad, pis)
        assert e.value.status == -1 and not isinstance(e.value, m.ProofRejected)
    with pytest.raises(m.Lcp2Error) as e:
        vd.verify(proof, pis[:-1])
    assert e.value.status == -1
    oc.close()
    vd.close()


def test_challenger_rejects_a_full_input_buffer():
    """a Challenger never holds 8 buffered inputs (it duplexes at the eighth): a state that claims so is corrupt"""
    import ctypes
    import eth_lc_plonky2_amd as m
    ch = m.binding.Challenger()
    ch.observe([1, 2, 3])
    ch.state.input_len = 8
    with pytest.raises(m.Lcp2Error):
        ch.observe([4])
    with pytest.raises(m.Lcp2Error):
        ch.get(1)
    ch.state.input_len = 3
    ch.observe([4])
    assert ch.get(1).size == 1


def test_plonky2_gate_programs_hold_on_their_witness_rows(oracle):
    """The programs of plonky2's gates (circuit.py) vanish on rows filled by the matching generators: a PoseidonGate row
    from poseidon_py.gate_row (swap 0 and 1), a BaseSumGate row of bits, ConstantGate / ArithmeticGate / PublicInputGate rows."""
    import eth_lc_plonky2_amd as m
    from eth_lc_plonky2_amd import poseidon_py as pos
    params = m.standard_params(6, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=9, npi=11)  # 11 inputs: two sponge blocks
    gs = circ.gateset
    assert gs.names == ["NoopGate", "ConstantGate", "PublicInputGate", "BaseSumGate", "ArithmeticGate", "PoseidonGate"]
    assert [g.num_constraints for g in gs.gates] == [0, 2, 4, 64, 20, 123] and gs.num_selectors == 2
    oc = oracle_lib.OracleCircuit(oracle, circ)
    assert oc.check_witness(wires, pis)[0] == 0
    # the in-circuit hash is the transcript's hash
    want = np.zeros(4, dtype=np.uint64)
    oracle.orc_hash_no_pad(oracle_lib.vp(np.ascontiguousarray(pis)), len(pis), oracle_lib.vp(want))
    assert list(wires[:4, 0]) == list(want) == pos.hash_no_pad(pis)
    # a swapped PoseidonGate row satisfies the gate too: put one on a Poseidon row's place
    prow = int(np.nonzero(circ.constants_sigmas[1] == gs.index("PoseidonGate"))[0][0])
    w2 = wires.copy()
    w2[:135, prow] = np.array(pos.gate_row(list(range(1, 13)), 1), dtype=np.uint64)
    bad, first = oc.check_witness(w2, pis)
    assert bad == 0  # check_witness looks at gate constraints only (copy constraints are the permutation argument's job)
    w2[pos.W_DELTA, prow] ^= np.uint64(1)
    assert oc.check_witness(w2, pis)[0] > 0
    oc.close()


def test_recursion_gate_programs(oracle):
    """plonky2's ArithmeticExtensionGate, MulExtensionGate, ReducingGate, ReducingExtensionGate, RandomAccessGate, ExponentiationGate
    and PoseidonMdsGate as programs (recursion_gates.py): constraint counts as plonky2's num_constraints(), zero on rows filled by
    the matching generators, non-zero when a wire the gate constrains is changed, and a circuit made of them proves (oracle) and
    verifies (product)."""
    import eth_lc_plonky2_amd as m
    from eth_lc_plonky2_amd import recursion_gates as rg
    params = m.standard_params(6, 4)
    circ, wires, pis = rg.recursion_gates_circuit(params, seed=5)
    gs = circ.gateset
    assert gs.names == ["NoopGate", "PoseidonMdsGate", "ReducingExtensionGate", "ReducingGate", "ArithmeticExtensionGate", "MulExtensionGate",
                        "ExponentiationGate", "RandomAccessGate"]
    assert [g.num_constraints for g in gs.gates] == [0, 2 * 12, 2 * 32, 2 * 43, 2 * 10, 2 * 13, 66 + 1, 4 * (4 + 2) + 2] and gs.num_selectors == 2
    oc = oracle_lib.OracleCircuit(oracle, circ)
    assert oc.check_witness(wires, pis)[0] == 0
    # rows 0..6 hold one gate of each kind in the order Reducing, ArithmeticExtension, MulExtension, Exponentiation, RandomAccess,
    # ReducingExtension, PoseidonMds; per kind: wires whose change must break the row
    routed_ra = (2 + 16) * 4 + 2
    for row, wire in ((0, 3), (0, 6 + 42), (0, 6 + 43 + 5), (1, 0), (1, 77), (2, 4), (2, 6 * 12 + 5), (3, 0), (3, 1 + 65), (3, 67), (3, 2 + 66 + 30),
                      (4, 0), (4, 1), (4, 18 + 1), (4, routed_ra + 2), (4, 72), (5, 2), (5, 6 + 2 * 31 + 1), (5, 6 + 64 + 9), (6, 5), (6, 24 + 23)):
        w2 = wires.copy()
        w2[wire, row] = np.uint64((int(w2[wire, row]) + 1) % m.GOLDILOCKS_P)
        bad, first = oc.check_witness(w2, pis)
        assert bad > 0 and first[0] == row, (row, wire)
    # an unselected list item of a RandomAccessGate row is free
    idx = int(wires[0, 4])
    w2 = wires.copy()
    w2[2 + (idx + 1) % 16, 4] ^= np.uint64(5)
    assert oc.check_witness(w2, pis)[0] == 0
    proof = oc.prove(wires, pis)
    assert oc.verify(proof, pis) == 0
    digest, cap = oc.digest()
    vd = m.CircuitData.verifier_only(circ, digest, cap)
    vd.verify(proof, pis)
    bad = proof.copy()
    bad[3 * 64 + 9] ^= np.uint64(1)  # an opening
    with pytest.raises(m.ProofRejected):
        vd.verify(bad, pis)
    oc.close()
    vd.close()


def test_reference_gate_programs(oracle):
    """The gates the reference's circuit is really made of (u32_gates.py: plonky2_u32's U32ArithmeticGate, U32AddManyGate,
    U32SubtractionGate, U32RangeCheckGate, ComparisonGate and plonky2's CosetInterpolationGate; [RECALL], parity unpinned) as
    programs: constraint counts as the gates' num_constraints() formulas, zero on rows filled by the matching generators (incl. the
    corner cases: equal inputs of a comparison, the largest product of an arithmetic row), non-zero when a wire the gate constrains
    is changed, and a circuit made of them proves (oracle) and verifies (product)."""
    import eth_lc_plonky2_amd as m
    from eth_lc_plonky2_amd import u32_gates as ug
    params = m.standard_params(6, 5)
    circ, wires, pis = ug.reference_gates_circuit(params, seed=9)
    gs = circ.gateset
    assert gs.names == ["NoopGate", "ComparisonGate", "U32AddManyGate", "U32ArithmeticGate", "U32RangeCheckGate", "U32SubtractionGate",
                        "CosetInterpolationGate"]
    assert gs.num_selectors == 3
    want = {"ComparisonGate": 6 + 5 * 16 + 2, "U32AddManyGate": 5 * (3 + 18), "U32ArithmeticGate": 3 * (4 + 32), "U32RangeCheckGate": 7 * 17,
            "U32SubtractionGate": 6 * (3 + 16), "CosetInterpolationGate": 2 + 4 * 2 + 2}
    for name, count in want.items():
        assert gs.gates[gs.index(name)].num_constraints == count, name
    # corner-case rows: a comparison of equal inputs, an arithmetic row with the largest possible product
    rng = np.random.default_rng(1)
    kinds = list(ug.ROW_GENERATORS)
    wires[:, kinds.index("ComparisonGate")] = np.array(ug.row_comparison(rng, 135, equal=True), dtype=np.uint64)
    wires[:, kinds.index("U32ArithmeticGate")] = np.array(ug.row_u32_arithmetic(rng, 135, force_high_max=True), dtype=np.uint64)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    bad, first = oc.check_witness(wires, pis)
    assert bad == 0, first
    # rows 0..5 hold one gate of each kind in the order of ROW_GENERATORS; per kind: wires whose change must break the row
    row_of = {k: i for i, k in enumerate(kinds)}
    cases = [("U32ArithmeticGate", w) for w in (0, 2, 3, 4, 5, 6 + 3, 18 + 0, 18 + 31, 18 + 32 + 17)] + \
            [("U32AddManyGate", w) for w in (0, 3, 4, 5, 6 + 2, 30 + 0, 30 + 17, 30 + 18 * 4 + 16)] + \
            [("U32SubtractionGate", w) for w in (0, 1, 2, 3, 4, 5 * 6 + 7, 30 + 16 * 5 + 15)] + \
            [("U32RangeCheckGate", w) for w in (0, 6, 7, 7 + 16 * 6 + 15)] + \
            [("ComparisonGate", w) for w in (0, 1, 2, 3, 4 + 5, 4 + 16 + 9, 4 + 48 + 2, 4 + 64 + 15, 4 + 80, 4 + 82)] + \
            [("CosetInterpolationGate", w) for w in (0, 1, 32, 33, 35, 36, 37 + 3, 37 + 8, 46)]
    for kind, wire in cases:
        w2 = wires.copy()
        w2[wire, row_of[kind]] = np.uint64((int(w2[wire, row_of[kind]]) + 1) % m.GOLDILOCKS_P)
        bad, first = oc.check_witness(w2, pis)
        assert bad > 0 and first[0] == row_of[kind], (kind, wire)
    # wires a gate does not use are free
    for kind, wire in (("U32RangeCheckGate", 7 + 16 * 7), ("ComparisonGate", 4 + 83), ("CosetInterpolationGate", 47), ("U32SubtractionGate", 30 + 16 * 6)):
        w2 = wires.copy()
        w2[wire, row_of[kind]] ^= np.uint64(5)
        assert oc.check_witness(w2, pis)[0] == 0, (kind, wire)
    proof = oc.prove(wires, pis)
    assert oc.verify(proof, pis) == 0
    digest, cap = oc.digest()
    vd = m.CircuitData.verifier_only(circ, digest, cap)
    vd.verify(proof, pis)
    bad = proof.copy()
    bad[3 * 64 + 9] ^= np.uint64(1)  # an opening
    with pytest.raises(m.ProofRejected):
        vd.verify(bad, pis)
    oc.close()
    vd.close()


def test_proof_byte_serialisation_round_trip(oracle):
    """ProofWithPublicInputs <-> bytes (plonky2 util/serialization.rs layout, PARITY UNPINNED: the reference holds no serialised
    proof).  Round trip, exact size, u8 sibling counts where the layout says, and every malformed buffer is refused."""
    import struct
    import eth_lc_plonky2_amd as m
    params = m.standard_params(6, 4)
    circ, wires, pis = m.circuit.synthetic_circuit(params, seed=12)
    oc = oracle_lib.OracleCircuit(oracle, circ)
    proof = oc.prove(wires, pis)
    vd = m.CircuitData.verifier_only(circ, *oc.digest())
    for flags in (0, m.binding.SER_PUBLIC_INPUT_COUNT):
        data = m.proof_to_bytes(params, proof, pis, flags)
        nsib_lists = params.num_query_rounds * (4 + params.num_fri_layers)
        assert len(data) == 8 * proof.size + nsib_lists + 8 * len(pis) + (8 if flags else 0)
        back, back_pis = m.proof_from_bytes(params, data, len(pis), flags)
        assert (back == proof).all() and (back_pis == pis).all()
        vd.verify(back, back_pis)
    # layout spot checks: the three caps come first as plain little-endian words; the first MerkleProof of the first query round
    # starts with its sibling count (lg(8n) - cap_height) as one byte, right after the constants/sigmas leaf
    capw = 4 << params.cap_height
    assert list(struct.unpack("<%dQ" % (3 * capw), data[:24 * capw])) == [int(x) for x in proof[:3 * capw]]
    NC, NR, W, CH, Q = params.num_constants, params.num_routed_wires, params.num_wires, params.num_challenges, params.quotient_degree_factor
    npp = -(-NR // Q) - 1
    openings = 2 * (NC + NR + W + CH + CH + CH * npp + CH * Q)
    first_query = 8 * (3 * capw + openings + params.num_fri_layers * capw)
    assert data[first_query + 8 * (NC + NR)] == params.degree_bits + params.rate_bits - params.cap_height
    # the public inputs are the tail, preceded by their count
    assert struct.unpack("<Q", data[-8 * len(pis) - 8:-8 * len(pis)])[0] == len(pis)
    # malformed buffers
    for bad in (data[:-1], data + b"\\0", data[:100]):
        with pytest.raises(m.Lcp2Error):
            m.proof_from_bytes(params, bad, len(pis))
    tampered = bytearray(data)
    tampered[first_query + 8 * (NC + NR)] ^= 1          # a sibling count
    with pytest.raises(m.Lcp2Error):
        m.proof_from_bytes(params, bytes(tampered), len(pis))
    tampered = bytearray(data)
    tampered[0:8] = struct.pack("<Q", 2 ** 64 - 1)       # a non-canonical field element
    with pytest.raises(m.Lcp2Error):
        m.proof_from_bytes(params, bytes(tampered), len(pis))
    with pytest.raises(m.Lcp2Error):
        m.proof_from_bytes(params, data, len(pis) + 1)
    # VerifierOnlyCircuitData: cap then digest
    vb = vd.verifier_data_bytes()
    digest, cap = oc.digest()
    assert vb == cap.tobytes() + digest.tobytes()
    oc.close()
    vd.close()
