#!/bin/bash
# regenerates eth-lc-plonky2_amd/csrc/generated_gates.hpp from the host layer's gate programs
set -e
cd "$(dirname "$0")/../.."
g++ -O1 -std=c++17 -o /tmp/gen_native_gates tools/gen/gen_native_gates.cpp eth-lc-plonky2_amd/host/gates.cpp eth-lc-plonky2_amd/host/poseidon_host.cpp
/tmp/gen_native_gates > eth-lc-plonky2_amd/csrc/generated_gates.hpp
wc -l eth-lc-plonky2_amd/csrc/generated_gates.hpp
