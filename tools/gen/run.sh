#!/bin/bash
# regenerates eth-lc-plonky2_amd/csrc/generated_gates*.hpp from the host layer's gate programs (host/gates.cpp) and the dump of the
# Python gate libraries (tools/gen/reference_gate_programs.txt, itself rewritten first from u32_gates.py / recursion_gates.py)
set -e
cd "$(dirname "$0")/../.."
python3 tools/gen/dump_reference_programs.py
g++ -O1 -std=c++17 -o /tmp/gen_native_gates tools/gen/gen_native_gates.cpp eth-lc-plonky2_amd/host/gates.cpp eth-lc-plonky2_amd/host/poseidon_host.cpp
/tmp/gen_native_gates tools/gen/reference_gate_programs.txt eth-lc-plonky2_amd/csrc
wc -l eth-lc-plonky2_amd/csrc/generated_gates*.hpp
