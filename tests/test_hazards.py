"""Static wait-state check of the device code in liblcp2.so (tools/check_hazards.py).

Parity tests cannot see an under-padded carry chain (commit dc307ac: a pair with one wait state too few had passed
all of them), so the build disassembles the library and checks every VALU-writes-VCC/SGPR -> VALU-reads pair.  These
tests pin the checker itself: it must flag an under-padded pair, accept hipcc's own padding, and actually cover the
hand-written multiply of csrc/gl64.hpp inside the Poseidon kernels.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_hazards as ch  # noqa: E402


def asm(*insns):
    lines = ["0000000000001000 <k_test>:"]
    for i, t in enumerate(insns):
        lines.append("\t%-58s // %012X: 00000000" % (t, 0x1000 + 4 * i))
    return lines


def test_flags_one_wait_state_between_carry_out_and_select():
    v, s = ch.scan(asm("v_subbrev_co_u32_e32 v1, vcc, 0, v5, vcc", "s_nop 0", "v_cndmask_b32_e32 v0, 0, v9, vcc"))
    assert len(v) == 1 and v[0][3] == 1 and s["pairs"] == 1


def test_flags_back_to_back_carry_chain():
    v, _ = ch.scan(asm("v_add_co_u32_e32 v1, vcc, v2, v3", "v_addc_co_u32_e32 v4, vcc, v5, v6, vcc"))
    assert len(v) == 1 and v[0][3] == 0


def test_accepts_two_wait_states_in_any_form():
    for mid in (["s_nop 1"], ["s_nop 0", "v_mov_b32_e32 v7, v8"], ["v_mov_b32_e32 v7, v8", "v_mov_b32_e32 v9, v8"]):
        v, s = ch.scan(asm("v_cmp_lt_u64_e32 vcc, s[24:25], v[20:21]", *mid, "v_cndmask_b32_e32 v29, v21, v45, vcc"))
        assert not v and s["pairs"] == 1 and s["min_wait_states"] == 2


def test_sgpr_pair_carry_out_and_mad_carry():
    v, _ = ch.scan(asm("v_mad_u64_u32 v[2:3], s[4:5], v6, v7, v[2:3]", "s_nop 0", "v_cndmask_b32_e64 v1, 0, 1, s[4:5]"))
    assert len(v) == 1
    v, _ = ch.scan(asm("v_mad_u64_u32 v[2:3], vcc, v6, v7, v[2:3]", "s_nop 1", "v_cndmask_b32_e64 v1, 0, 1, vcc"))
    assert not v


def test_scalar_rewrite_ends_the_window_and_functions_are_independent():
    v, _ = ch.scan(asm("v_cmp_eq_u32_e32 vcc, v1, v2", "s_mov_b64 vcc, s[2:3]", "v_cndmask_b32_e32 v0, v1, v2, vcc"))
    assert not v
    lines = asm("v_cmp_eq_u32_e32 vcc, v1, v2") + asm("v_cndmask_b32_e32 v0, v1, v2, vcc")
    v, _ = ch.scan(lines)
    assert not v


def test_built_library_is_clean_and_the_check_is_not_vacuous():
    import eth_lc_plonky2_amd as m
    lib = m.build_native()
    v, s = ch.check_library(lib)
    assert not v, v[:5]
    # the Poseidon kernels alone hold thousands of hand-written carry chains
    assert s["pairs"] > 5000 and s["min_wait_states"] == 2 and s["pairs_at_min"] > 1000
    # and the hand-written multiply is really in there (17-instruction form: v_mad_u64_u32 with a vcc carry-out)
    text = "".join(ch.device_disassembly(lib))
    assert text.count("v_mad_u64_u32") > 1000 and "k_hash_leaves" in text
