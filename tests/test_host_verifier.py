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
        for w0, w1 in [((saved[0] & ~0xF) | 9, saved[1]),                    # unknown opcode
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
