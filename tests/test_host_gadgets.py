"""Circuit-level tests of the C++ host layer (eth-lc-plonky2_amd/host): the reference's own #[test]s restated on the
mirrored CircuitBuilder / gadget API (tests/cpp/test_gadgets.cpp).  CPU mode: witness generation, every gate
constraint checked row-wise, oracle prove + verify, product host verifier.  GPU mode: data.prove on the MI355X."""
import pytest

import cpp_build

CPU_TESTS = [
    "test_merkle_root_2_leaves", "test_merkle_root_4_leaves", "test_merkle_root_8_leaves", "test_merkle_root_16_leaves",
    "test_merkle_root_wrong_root_panics", "test_signing_root", "test_beacon_block_header", "test_verify_finality_branch",
    "test_contract_state", "test_verify_sync_committee_branch", "test_verify_sync_committee_branch_panics",
    "test_read_u32_be_public_input", "test_ssz_sync_committee", "test_light_client_update",
    "test_light_client_update_bad_state_root_panics", "test_light_client_update_low_participation_panics",
    "test_light_client_update_with_recursive_proof", "test_light_client_update_recursive_proof_of_other_bits_panics",
    "test_find_sync_committee_current_period", "test_find_sync_committee_next_period",
    "test_find_sync_committee_stale_period_panics", "test_find_sync_committee_previous_period_panics",
    "test_slot_connect_rejects_wide_encoding_panics", "test_update_validity", "test_update_validity_equal_slots_and_343",
    "test_update_validity_finalized_before_current_panics", "test_update_validity_threshold_not_exceeded_panics",
    # the recursive verifier gadget (host/recursion.cpp): an inner proof checked natively and in-circuit; each tampered word
    # (opening, cap, leaf, FRI evaluation, final polynomial, PoW witness, Merkle sibling, public input, digest) must fail
    # BigUint gadgets (host/biguint.cpp): the reference's plonky2_crypto biguint surface and its two BigUint sub-circuits
    "test_biguint_arithmetic", "test_biguint_arithmetic_all_ones", "test_biguint_division_by_zero_panics", "test_biguint_hash256_connect",
    "test_find_sync_committee_big_current_period", "test_find_sync_committee_big_next_period", "test_find_sync_committee_big_stale_period_panics",
    "test_update_validity_big", "test_update_validity_big_finalized_before_current_panics", "test_update_validity_big_threshold_not_exceeded_panics",
    "test_poseidon_gate_outputs_match_rows", "test_builder_primitives", "test_builder_inverse_of_zero_panics",
    "test_recursive_verifier", "test_recursive_verifier_constant_verifier_data_sha_inner",
    "test_recursive_verifier_tampered_opening_panics", "test_recursive_verifier_tampered_cap_panics",
    "test_recursive_verifier_tampered_leaf_panics", "test_recursive_verifier_tampered_fri_layer_panics",
    "test_recursive_verifier_tampered_final_poly_panics", "test_recursive_verifier_tampered_pow_panics",
    "test_recursive_verifier_tampered_sibling_panics", "test_recursive_verifier_wrong_public_input_panics",
    "test_recursive_verifier_wrong_digest_panics",
]


# the reference's scale made of real gadgets (2.24 M gates, 2^22 rows): GPU prove, device witness = host witness, oracle verifier
GPU_ONLY_TESTS = ["test_real_gadget_circuit_2p22"]


def test_test_list_is_complete():
    out = cpp_build.run("cpu", "list").stdout.split()
    assert out == CPU_TESTS
    assert cpp_build.run("cpu", "list-gpu-only").stdout.split() == GPU_ONLY_TESTS


@pytest.mark.parametrize("name", CPU_TESTS)
def test_gadget_cpu(name):
    r = cpp_build.run("cpu", name)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"test {name} ... ok" in r.stdout


@pytest.mark.gpu
def test_gadgets_gpu_all():
    # BASELINE configs[0] (ContractState), configs[1] (SyncCommitteeSSZ, 2^19 rows) and configs[2] (light-client update
    # 633 -> 634, BLS verifier stubbed) end to end on the GPU
    r = cpp_build.run("gpu", "all", timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    for name in CPU_TESTS + GPU_ONLY_TESTS:
        assert f"test {name} ... ok" in r.stdout
    # every proving test ends with the oracle's verifier (built from digest + cap alone) accepting the GPU proof and rejecting two
    # one-word changes; the 2^19-row light-client proofs and the 2^22-row real-gadget proof are among them
    assert r.stdout.count("oracle verifier (digest + cap only) accepted the GPU proof") >= 20
    assert "rejected two one-word changes (degree_bits 22)" in r.stdout
    assert "rejected two one-word changes (degree_bits 19)" in r.stdout
