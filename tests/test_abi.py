"""The C-ABI library builds, loads and exports every symbol include/lcp2.h declares (no GPU needed),
and refuses to compute without a device instead of falling back to the CPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "lcp2.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lcp2_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import eth_lc_plonky2_amd as m
    lib = m.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"liblcp2.so does not export {n}"
    assert lib.lcp2_abi_version() == 2


def test_standard_params_match_standard_recursion_config():
    import eth_lc_plonky2_amd as m
    p = m.standard_params(22, 5)
    assert (p.num_wires, p.num_routed_wires, p.rate_bits, p.cap_height, p.num_challenges) == (135, 80, 3, 4, 2)
    assert (p.quotient_degree_factor, p.proof_of_work_bits, p.num_query_rounds) == (8, 16, 28)
    assert p.num_fri_layers == 5 and list(p.fri_arity_bits)[:5] == [4] * 5  # 22 -> 18 -> 14 -> 10 -> 6 -> 2
    p = m.standard_params(12, 5)
    assert p.num_fri_layers == 2  # 12 -> 8 -> 4
    p = m.standard_params(5, 5)
    assert p.num_fri_layers == 0


def test_no_cpu_fallback_without_device():
    import eth_lc_plonky2_amd as m
    lib = m.load_library()
    if lib.lcp2_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(m.Lcp2Error) as e:
        m.Context(0)
    assert e.value.status == -2
    assert lib.lcp2_status_str(-5) == b"witness does not satisfy the circuit"
